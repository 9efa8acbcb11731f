"""ora_grid.py -- ORACLE (test infrastructure only: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
may import it; the product never does).

CPU restatement of the reference's grid-line search, SURVEY 8(f) rank 3:

    StonesFinder.find_intersections   /root/reference/src/camkifu/stone/stonesfinder.py:516-552
    StonesFinder.get_intersections    stonesfinder.py:554-576
    update_grid                        stonesfinder.py:888-947
    PosGrid.learn                      stonesfinder.py:1015-1043
    imgutil.Segment / within_margin    /root/reference/src/camkifu/core/imgutil.py:344-357, 464-530

"parity unpinned": the arithmetic of cv2.HoughLinesP (OpenCV 3.1.0, absent here) is restated from the published
progressive probabilistic Hough transform (Matas, Galambos, Kittler 2000) as the library implements it; every
library-side assumption is listed so that a session with cv2 can confirm or correct it:

  * the random order comes from cv::RNG seeded with (uint64)-1: multiply-with-carry, state = (uint32)state * 4164903690
    + (state >> 32), uniform(0, count) = (uint32)state % count; the chosen point is replaced by the last of the list;
  * trig table (float)(cos((double)n * theta)), theta = (float)(pi / 180), 180 angles, numrho = 2 (w + h) + 1,
    r = cvRound(x * cos + y * sin) in float arithmetic (round half to even) + (numrho - 1) / 2;
  * a point votes in all 180 rows; the first angle whose counter is the largest one >= threshold gives the line;
  * the walk in 16.16 fixed point from the point in both directions over the remaining-points mask, stopping at the
    border or after maxLineGap + 1 empty pixels; a line is kept when its extent in x or in y reaches minLineLength; its
    pixels leave the mask, and if the line is kept their votes are taken back;
  * cv2.threshold(gray, 1, 1, THRESH_OTSU) returns the Otsu level (oracle.otsu_level); cv2.Canny on one channel with
    (level / 2, level) floored.
"""
import math
import sys

import numpy as np

from . import oracle as O

GSIZE = 19


class CvRNG:
    """cv::RNG"""

    def __init__(self, state=0xFFFFFFFFFFFFFFFF):
        self.state = state

    def next(self):
        self.state = ((self.state & 0xFFFFFFFF) * 4164903690 + (self.state >> 32)) & 0xFFFFFFFFFFFFFFFF
        return self.state & 0xFFFFFFFF

    def uniform(self, a, b):
        return a if a == b else self.next() % (b - a) + a


def _trigtab():
    theta = float(np.float32(math.pi / 180))
    n = np.arange(180, dtype=np.float64)
    return np.cos(n * theta).astype(np.float32), np.sin(n * theta).astype(np.float32)


_COS, _SIN = _trigtab()


def _cv_round32(x):
    return int(np.rint(np.float32(x)))


def hough_lines_p(image, threshold, min_len, max_gap=0):
    """cv2.HoughLinesP(image, 1, pi / 180, threshold, minLineLength=min_len, maxLineGap=max_gap) -> list of
    (x0, y0, x1, y1) in the order the library appends them"""
    image = np.asarray(image)
    height, width = image.shape
    numrho = (width + height) * 2 + 1
    accum = np.zeros((180, numrho), np.int32)
    mask = (image != 0).astype(np.uint8)
    ys, xs = np.nonzero(mask)                                  # raster order
    nzloc = list(zip(xs.tolist(), ys.tolist()))
    rng = CvRNG()
    rows = np.arange(180)
    half = (numrho - 1) // 2
    lines = []
    SHIFT = 16
    for count in range(len(nzloc), 0, -1):
        idx = rng.uniform(0, count)
        j, i = nzloc[idx]
        nzloc[idx] = nzloc[count - 1]
        if not mask[i, j]:
            continue
        r = np.rint(np.float32(j) * _COS + np.float32(i) * _SIN).astype(np.int64) + half
        accum[rows, r] += 1
        vals = accum[rows, r]
        max_n = int(np.argmax(vals))                           # first of the largest
        if vals[max_n] < threshold:
            continue
        a, b = np.float32(-_SIN[max_n]), np.float32(_COS[max_n])
        x0, y0 = j, i
        if abs(a) > abs(b):
            xflag = True
            dx0 = 1 if a > 0 else -1
            dy0 = _cv_round32(np.float32(b * np.float32(1 << SHIFT)) / np.float32(abs(a)))
            y0 = (y0 << SHIFT) + (1 << (SHIFT - 1))
        else:
            xflag = False
            dy0 = 1 if b > 0 else -1
            dx0 = _cv_round32(np.float32(a * np.float32(1 << SHIFT)) / np.float32(abs(b)))
            x0 = (x0 << SHIFT) + (1 << (SHIFT - 1))
        ends = [None, None]
        for k in range(2):
            gap, x, y = 0, x0, y0
            dx, dy = (dx0, dy0) if k == 0 else (-dx0, -dy0)
            while True:
                j1, i1 = (x, y >> SHIFT) if xflag else (x >> SHIFT, y)
                if j1 < 0 or j1 >= width or i1 < 0 or i1 >= height:
                    break
                if mask[i1, j1]:
                    gap = 0
                    ends[k] = (j1, i1)
                else:
                    gap += 1
                    if gap > max_gap:
                        break
                x += dx
                y += dy
        good = abs(ends[1][0] - ends[0][0]) >= min_len or abs(ends[1][1] - ends[0][1]) >= min_len
        for k in range(2):
            x, y = x0, y0
            dx, dy = (dx0, dy0) if k == 0 else (-dx0, -dy0)
            while True:
                j1, i1 = (x, y >> SHIFT) if xflag else (x >> SHIFT, y)
                if mask[i1, j1]:
                    if good:
                        rr = np.rint(np.float32(j1) * _COS + np.float32(i1) * _SIN).astype(np.int64) + half
                        accum[rows, rr] -= 1
                    mask[i1, j1] = 0
                if (j1, i1) == ends[k]:
                    break
                x += dx
                y += dy
        if good:
            lines.append((ends[0][0], ends[0][1], ends[1][0], ends[1][1]))
    return lines


# ---- imgutil.Segment, the part update_grid uses ------------------------------------------------------------------
def seg_theta(c):
    return math.acos((c[2] - c[0]) / math.sqrt((c[0] - c[2]) ** 2 + (c[1] - c[3]) ** 2))


def seg_intersection(s, o):
    x = (o[0] - s[0], o[1] - s[1])
    d1 = (s[2] - s[0], s[3] - s[1])
    d2 = (o[2] - o[0], o[3] - o[1])
    cross = float(d1[0] * d2[1] - d1[1] * d2[0])
    if abs(cross) < sys.float_info.epsilon:
        return None
    t1 = (x[0] * d2[1] - x[1] * d2[0]) / cross
    return int(s[0] + t1 * d1[0]), int(s[1] + t1 * d1[1])


def within_margin(p, box, margin):
    return box[0] + margin < p[0] < box[2] - margin and box[1] + margin < p[1] < box[3] - margin


def update_grid(lines, box, slot):
    """stonesfinder.py:888-947; slot: the int16 pair of this intersection, modified in place.
    Lines that are level or upright (|cos| or |sin| of their angle above 0.995) and pass through the middle of the zone
    (margin = a seventh of its smaller side) count; any such line marks the zone empty (negated position); two to four
    of them move the position to the mean of their pairwise crossings that fall inside the margin."""
    x_lo, y_lo, x_hi, y_hi = box
    margin = min(x_hi - x_lo, y_hi - y_lo) / 7
    kept = []
    for c in lines:
        theta = seg_theta(c)
        if 0.995 < abs(math.cos(theta)):
            probe = ((x_lo + x_hi) / 2, (c[0] + c[2]) / 2 + y_lo)
        elif 0.995 < abs(math.sin(theta)):
            probe = ((c[1] + c[3]) / 2 + x_lo, (y_hi + y_lo) / 2)
        else:
            continue
        if within_margin(probe, box, margin):
            kept.append(c)
    if not kept:
        return
    slot *= -1
    if not 1 < len(kept) < 5:
        return
    sx = sy = hits = 0
    for a, first in enumerate(kept):
        for b, second in enumerate(kept):
            if a == b:
                continue
            cross = seg_intersection(first, second)
            if cross is None:
                continue
            q = (cross[1] + x_lo, cross[0] + y_lo)               # the crossing is in (column, row) order
            if within_margin(q, box, margin):
                sx, sy, hits = sx + q[0], sy + q[1], hits + 1
    if hits:
        slot[0] = np.int16(int(-sx / hits))
        slot[1] = np.int16(int(-sy / hits))


def grid_canny(img):
    """the edge map find_intersections works on: grey, Otsu level, Canny(gray, level / 2, level)"""
    gray = O.bgr2gray(img)
    level = O.otsu_level(gray)
    return O.canny(gray, int(np.floor(level / 2)), int(np.floor(level)))


def find_intersections(img, mtx, rects, want_lines=False):
    """stonesfinder.py:516-552: img (s, s, 3) goban image, mtx (19, 19, 2) int16 PosGrid.mtx, rects (19, 19, 4) getrect
    table -> the grid with the positions where a line was found negated (and moved where a cross was found)"""
    canny = grid_canny(img)
    grid = np.array(mtx, np.int16, copy=True)
    found = {}
    for r in range(GSIZE):
        for c in range(GSIZE):
            x0, y0, x1, y1 = (int(v) for v in rects[r][c])
            zone = canny[x0:x1, y0:y1]
            min_side = min(zone.shape[0], zone.shape[1])
            lines = hough_lines_p(zone, int(min_side * 3 / 4), int(min_side * 2 / 3), 0)
            if lines:
                found[(r, c)] = lines
                update_grid(lines, (x0, y0, x1, y1), grid[r][c])
    return (grid, found, canny) if want_lines else grid


class GridState:
    """PosGrid's learning state (stonesfinder.py:964-969): mtx int16 (19, 19, 2), adjust_vect float32 (2,), adjust_contribs"""

    def __init__(self, mtx):
        self.mtx = np.array(mtx, np.int16, copy=True)
        self.adjust_vect = np.zeros(2, np.float32)
        self.adjust_contribs = 0

    def learn(self, grid, rate=0.2):
        """stonesfinder.py:1015-1043: the mean displacement of the intersections that moved, blended into adjust_vect;
        applied to the whole grid (truncated to int16) once more than 20 intersections have contributed"""
        if not 0 < rate <= 1:
            raise AssertionError("rate")
        shift = np.asarray(grid, np.int16) - self.mtx
        if shift.min() < -200:
            raise ValueError("Provided grid seems too far from original, at least for one point.")
        movers = int((shift != 0).any(-1).sum())
        if not movers:
            return
        mean = shift.sum(axis=(0, 1), dtype=np.float32)
        mean /= movers
        self.adjust_vect *= (1.0 - rate)
        self.adjust_vect += mean * rate
        self.adjust_contribs += movers
        if self.adjust_contribs > 20:
            self.mtx += self.adjust_vect.astype(np.int16)
            self.adjust_vect[:] = 0
            self.adjust_contribs = 0
