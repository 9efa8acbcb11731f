"""ora_stones.py -- ORACLE (test infrastructure only: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
may import it; the product never does).

CPU restatement of the reference's contour-based stones finder, SURVEY 8(f) rank 3:

    SfContours.find_stones          /root/reference/src/camkifu/stone/sf_contours.py:48-111
    SfContours._norm_channels       sf_contours.py:113-126
    SfContours.find_color           sf_contours.py:128-184
    SfContours._filter_contours     sf_contours.py:186-205
    SfContours.analyse_fg           sf_contours.py:207-249
    SfContours.extract_contours_fg  sf_contours.py:251-300
    SfContours._find_centers        sf_contours.py:302-330

The arithmetic lives in OpenCV 3.1.0 (absent here, as everywhere in this tree): "parity unpinned".  What is assumed
about the library, call by call, so that a session with cv2 can confirm or correct it:

  * cv2.morphologyEx(fg, MORPH_OPEN, (5, 5), iterations=3): the binding turns the TUPLE (5, 5) into a 2x1 matrix of
    doubles (two rows, one column, both non-zero) -- not a 5x5 element.  morphOp folds the three iterations of an
    all-ones element into one 4x1 element anchored at its last row, so the opening is a 4-row erosion followed by a
    4-row dilation, both over rows y-3 .. y, rows outside the image ignored (the default border value).
  * cv2.Canny on the one-channel result: K2's algorithm (oracle.canny) with one channel.
  * cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE): oracle.find_external_suzuki, handed back in reverse
    discovery order; `cont.shape[0]` is the length of the compressed vertex list.
  * cv2.minAreaRect: ora_min_area_rect_box (float rotating calipers on the clockwise hull; angle from the first
    vector).  cv2.convexHull: the strictly convex vertex set.  cv2.boundingRect: min / max + 1.
  * cv2.drawContours(thickness=-1) = CollectPolyEdges + FillEdgeCollection of drawing.cpp: every polygon side drawn
    with the 8-connected LineIterator (left to right), then the scanline fill in 16.16 fixed point (edge slope
    = (dx << 16) / dy truncated, span = ceil(left) .. floor(right), the last scanline left to the outline).
  * cv2.drawContours(thickness=1) of a compressed contour = the pixels the border follower visited.
  * cv2.distanceTransform(DIST_L2, DIST_MASK_5): the library's own two-pass 5x5 chamfer in 16.16 fixed point with
    (1, 1.4, 2.1969) -- the non-IPP path; cv2.minMaxLoc: first maximum in raster order.
"""
import math

import numpy as np

from . import oracle as O

GSIZE = 19
E, B, W = 0, 1, 2


# ---------------------------------------------------------------------------------------------- library pieces
def morph_open_rows(fg):
    """the opening the reference actually asks for (see the header): min over rows y-3..y, then max over rows y-3..y"""
    def window(a, op, fill):
        out = a.copy()
        for k in (1, 2, 3):
            sh = np.full_like(a, fill)
            sh[k:] = a[:-k]
            out = op(out, sh)
        return out
    fg = np.ascontiguousarray(fg, np.uint8)
    return window(window(fg, np.minimum, 255), np.maximum, 0)


def contours_cv(edges):
    """cv2.findContours(edges, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)[1] as a list of dicts (vert, pix, start)"""
    return list(reversed(O.find_external_suzuki(edges)))


def min_area_rect_box(pts):
    pts = np.ascontiguousarray(pts, np.int32).reshape(-1, 2)
    out = np.zeros(3, np.float32)
    O.lib().ora_min_area_rect_box(O._vp(pts), len(pts), O._vp(out))
    return float(out[0]), float(out[1]), float(out[2])


def convex_hull(pts):
    pts = np.ascontiguousarray(pts, np.int32).reshape(-1, 2)
    out = np.zeros((len(pts) + 2, 2), np.int32)
    O.lib().ora_convex_hull.restype = int
    n = O.lib().ora_convex_hull(O._vp(pts), len(pts), O._vp(out))
    return out[:n].copy()


def bounding_rect(pts):
    pts = np.asarray(pts).reshape(-1, 2)
    x0, y0 = int(pts[:, 0].min()), int(pts[:, 1].min())
    return x0, y0, int(pts[:, 0].max()) - x0 + 1, int(pts[:, 1].max()) - y0 + 1


def line_pixels(p0, p1):
    """cv::LineIterator(img, p0, p1, 8, left_to_right=true) for points inside the image: the (x, y) it visits"""
    (x0, y0), (x1, y1) = p0, p1
    dx, dy = x1 - x0, y1 - y0
    if dx < 0:
        dx, dy, x0, y0 = -dx, -dy, x1, y1
    sy = -1 if dy < 0 else 1
    dy = abs(dy)
    steep = dy > dx
    if steep:
        dx, dy = dy, dx
    err = dx - (dy + dy)
    plus, minus = dx + dx, -(dy + dy)
    out = []
    x, y = x0, y0
    for _ in range(dx + 1):
        out.append((x, y))
        neg = err < 0
        err += minus + (plus if neg else 0)
        if steep:
            y += sy
            x += 1 if neg else 0
        else:
            x += 1
            y += sy if neg else 0
    return out


def fill_polygon(img, pts, value, offset=(0, 0)):
    """cv2.drawContours(img, [pts], 0, value, thickness=-1, offset=offset) for one polygon of integer vertices lying
    inside the image: CollectPolyEdges (outline + edge table) then FillEdgeCollection (drawing.cpp)"""
    h, w = img.shape[:2]
    XY = 16
    v = [(int(x) + offset[0], int(y) + offset[1]) for x, y in np.asarray(pts).reshape(-1, 2)]
    if not v:
        return
    v = v + [v[0]]                                   # cvDrawContours closes the polyline itself
    edges = []
    p0 = v[-1]
    for p1 in v:
        for x, y in line_pixels(p0, p1):
            assert 0 <= x < w and 0 <= y < h
            img[y, x] = value
        if p0[1] != p1[1]:
            fx0, fx1 = p0[0] << XY, p1[0] << XY
            num, den = fx1 - fx0, p1[1] - p0[1]
            q = abs(num) // abs(den)                             # C integer division truncates toward zero
            dxfix = q if (num > 0) == (den > 0) else -q
            if p0[1] < p1[1]:
                edges.append(dict(y0=p0[1], y1=p1[1], x=fx0, dx=dxfix))
            else:
                edges.append(dict(y0=p1[1], y1=p0[1], x=fx1, dx=dxfix))
        p0 = p1
    total = len(edges)
    if total < 2:
        return
    y_max = max(e["y1"] for e in edges)
    edges.sort(key=lambda e: (e["y0"], e["x"], e["dx"]))
    edges.append(dict(y0=1 << 40, y1=0, x=0, dx=0))
    i = 0
    e = edges[0]
    active = []
    y_max = min(y_max, h)
    for y in range(e["y0"], y_max):
        draw = 0
        pos = 0
        prev = None
        while True:
            last = active[pos] if pos < len(active) else None
            if last is None and e["y0"] != y:
                break
            if last is not None and last["y1"] == y:
                active.pop(pos)
                continue
            if last is not None and (e["y0"] > y or last["x"] < e["x"]):
                cur = last
                pos += 1
            elif i < total:
                active.insert(pos, e)
                cur = e
                pos += 1
                i += 1
                e = edges[i]
            else:
                break
            if draw:
                if y >= 0:
                    xa, xb = prev["x"], cur["x"]
                    if xa > xb:
                        xa, xb = xb, xa
                    xa = (xa + (1 << XY) - 1) >> XY
                    xb = xb >> XY
                    if xa < w and xb >= 0:
                        xa, xb = max(xa, 0), min(xb, w - 1)
                        if xa <= xb:
                            img[y, xa:xb + 1] = value
                prev["x"] += prev["dx"]
                cur["x"] += cur["dx"]
            prev = cur
            draw ^= 1
        active.sort(key=lambda q: q["x"])              # the library's bubble sort is stable


def distance_transform_5x5(img):
    """cv2.distanceTransform(img, DIST_L2, DIST_MASK_5) -> float32 (distransform.cpp, distanceTransform_5x5)"""
    img = np.asarray(img)
    h, w = img.shape
    HV = int(round(float(np.float32(1.0)) * 65536))
    DIAG = int(round(float(np.float32(1.4)) * 65536))
    LONG = int(round(float(np.float32(2.1969)) * 65536))
    INIT = (2 ** 31 - 1) >> 2
    t = np.full((h + 4, w + 4), INIT, np.int64)
    fwd = ((-2, -1, LONG), (-2, 1, LONG), (-1, -2, LONG), (-1, -1, DIAG), (-1, 0, HV), (-1, 1, DIAG), (-1, 2, LONG), (0, -1, HV))
    for i in range(h):
        for j in range(w):
            if not img[i, j]:
                t[i + 2, j + 2] = 0
            else:
                t[i + 2, j + 2] = min(t[i + 2 + di, j + 2 + dj] + c for di, dj, c in fwd)
    out = np.empty((h, w), np.float32)
    for i in range(h - 1, -1, -1):
        for j in range(w - 1, -1, -1):
            t0 = t[i + 2, j + 2]
            if t0 > HV:
                t0 = min(t0, min(t[i + 2 - di, j + 2 - dj] + c for di, dj, c in fwd))
                t[i + 2, j + 2] = t0
            out[i, j] = np.float32(t0) * np.float32(1.0 / 65536)
    return out


# ------------------------------------------------------------------------------------------ the finder's geometry
def getrect(r, c, size=380):
    """StonesFinder.getrect with cursor=1 on the default grid (stonesfinder.py:412-450): (x0, y0, x1, y1), x = rows"""
    return O.sf_getrect(r, c, 1.0, size)


def stone_radius(size=380):
    return size / GSIZE / 2                           # stonesfinder.py:578-584


def find_centers(dist, radius):
    """sf_contours.py:302-330 -> list of (x, y): the box is cut into cells of about one stone; a cell answers when the
    farthest point from the contour lies within a third of the cell's smaller side of the cell centre"""
    n_rows, n_cols = dist.shape
    cells_down = int(round(n_rows / 2 / radius))          # Python's round: ties to even
    cells_across = int(round(n_cols / 2 / radius))
    cell_h = int(n_rows / cells_down)                     # ZeroDivisionError for a box thinner than a radius, as the reference
    cell_w = int(n_cols / cells_across)
    reach = min(cell_h, cell_w) / 3
    out = []
    for a in range(cells_down):
        top = a * cell_h
        for b in range(cells_across):
            left = b * cell_w
            cell = dist[top:top + cell_h + 1, left:left + cell_w + 1]
            my, mx = divmod(int(np.argmax(cell)), cell.shape[1])          # first maximum, raster order
            if reach < math.sqrt((mx - cell_w / 2) ** 2 + (my - cell_h / 2) ** 2):
                continue
            out.append((left + mx, top + my))
    return out


def extract_contours_fg(sub_fg, radius):
    """sf_contours.py:251-300 -> the contours (dicts of oracle.find_external_suzuki) that could be a stone"""
    smoothed = morph_open_rows(sub_fg)
    canny = O.canny(smoothed, 25, 75)
    kept = []
    for cont in contours_cv(canny):
        if len(cont["vert"]) < 10:
            continue
        bw, bh, angle_deg = min_area_rect_box(cont["vert"])
        if min(bw, bh) < 3 / 2 * radius:
            continue
        if 5 * radius < max(bw, bh):
            continue
        angle = math.radians(angle_deg)
        if 2.5 * radius < max(bw, bh) and max(abs(math.cos(angle)), abs(math.sin(angle))) < 0.97:
            continue
        hull = convex_hull(cont["vert"])
        y0, x0, dy, dx = bounding_rect(hull)
        ghost = np.zeros((dx, dy), np.uint8)
        fill_polygon(ghost, hull, 1, offset=(-y0, -x0))
        ratio = int(np.sum(sub_fg[x0:x0 + dx, y0:y0 + dy].astype(np.int64) * (ghost == 1))) / dx / dy / 255
        if ratio < 0.3:
            continue
        kept.append(cont)
    return kept


def analyse_fg(fg, x0, y0, x1, y1, radius):
    """sf_contours.py:207-249"""
    sub_fg = np.ascontiguousarray(fg[x0:x1, y0:y1])
    kept = []
    ghost = np.zeros(sub_fg.shape, np.uint8)
    for cont in extract_contours_fg(sub_fg, radius):
        ghost[cont["pix"][:, 1], cont["pix"][:, 0]] = 255
        ry0, rx0, dy, dx = bounding_rect(cont["vert"])
        negative = 255 - ghost[rx0:rx0 + dx, ry0:ry0 + dy]
        if find_centers(distance_transform_5x5(negative), radius):
            kept.append(cont)
    return kept


def filter_contours(contours, radius):
    """sf_contours.py:186-205"""
    for cont in contours:
        if len(cont["vert"]) < 10:
            continue
        bw, bh, _ = min_area_rect_box(cont["vert"])
        if 10 * radius < max(bw, bh):
            continue
        yield cont


_AROUND = [(i, j) for i in (-1, 0, 1) for j in (-1, 0, 1) if (i, j) != (0, 0)]      # raster order, as the nested loops


def find_color(r, c, zones, stones):
    """sf_contours.py:128-184; `stones` (uint8 view, E/B/W = 0/1/2) is data and result slot.  The eight neighbours are
    visited in raster order; each may cast a vote; the third vote (or one look-alike empty neighbour) ends the search,
    and the zone takes a colour only if all votes agree."""
    votes, cast = set(), 0
    here = zones[r, c, 1:4].astype(np.int64)
    for i, j in _AROUND:
        rr, cc = r + i, c + j
        if not (0 <= rr < zones.shape[0] and 0 <= cc < zones.shape[1]):
            continue
        other = zones[rr, cc, 1:4].astype(np.int64)
        delta = here - other
        gap = int(np.abs(delta).sum())                        # |diff|; the sign of the plain sum says darker or brighter
        darker = int(delta.sum()) < 0
        if not zones[rr, cc, 0]:                              # a neighbour seen as bare board
            if gap > 100:
                votes.add(B if darker else W)
                cast += 1
            elif gap < 70:
                votes.add(E)
                cast = 3
        elif i < 1 and j < 1:                                 # a neighbour under a hull: only those already decided
            theirs = int(stones[rr, cc])
            if theirs not in (B, W):
                continue
            floor_ = min(int(here.sum()), int(other.sum()))
            if gap < floor_ * 0.1:
                votes.add(theirs)
                cast += 1
            elif floor_ < gap:
                votes.add(B if theirs == W else W)
                cast += 1
        if cast == 3:
            if len(votes) == 1:
                stones[r, c] = votes.pop()
            return


def find_stones(img, fg, rs=0, re=GSIZE, cs=0, ce=GSIZE, want_all=False):
    """sf_contours.py:48-111: img (380, 380, 3) uint8 goban image, fg (380, 380) uint8 foreground mask ->
    stones uint8 (19, 19) of 0 E / 1 B / 2 W (+ zones int16 (re-rs, ce-cs, 4), mask uint8 (h, w) with want_all)"""
    size = img.shape[0]
    radius = stone_radius(size)
    x0, y0, _, _ = getrect(rs, cs, size)
    _, _, x1, y1 = getrect(re - 1, ce - 1, size)
    contours_fg = analyse_fg(fg, x0, y0, x1, y1, radius)
    subimg = np.ascontiguousarray(img[x0:x1, y0:y1])
    canny = O.goban_canny(subimg)
    contours_img = list(filter_contours(contours_cv(canny), radius))
    mask = np.zeros(subimg.shape[:2], np.uint8)
    for cont in contours_fg + contours_img:
        fill_polygon(mask, convex_hull(cont["vert"]), 1)
    m3 = mask[:, :, None]
    visible_sub = subimg * m3
    masked_sub = subimg * (1 - m3)
    zones = np.zeros((re - rs, ce - cs, 4), np.int16)
    for zr, zc in np.ndindex(re - rs, ce - cs):
        a0, b0, a1, b1 = getrect(zr + rs, zc + cs, size)
        win = (slice(a0 - x0, a1 - x0), slice(b0 - y0, b1 - y0))
        whole = (a1 - a0) * (b1 - b0)
        seen = int(mask[win].sum())
        under_hull = 0.4 * whole < seen                       # masked if more than 60 % of the zone is outside the hulls
        src, norm = (visible_sub[win], seen) if under_hull else (masked_sub[win], whole - seen)
        zones[zr, zc, 0] = 1 if under_hull else 0
        zones[zr, zc, 1:] = [int(int(src[:, :, k].sum()) / norm) for k in range(3)]      # float mean, truncated into int16
    stones = np.zeros((GSIZE, GSIZE), np.uint8)
    view = stones[rs:re, cs:ce]
    for zr, zc in np.ndindex(re - rs, ce - cs):
        if zones[zr, zc, 0]:
            find_color(zr, zc, zones, view)
    if want_all:
        return stones, zones, mask, dict(fg=contours_fg, img=contours_img, canny=canny)
    return stones
