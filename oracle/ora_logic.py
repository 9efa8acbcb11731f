"""TEST INFRASTRUCTURE ONLY -- scalar restatement, in plain Python floats and loops, of the two ORDERED
parts of the reference's finders.  The product implements them once in C++ (camkifu_amd/csrc/ck_fold.cpp:
ck_boardfold_*, ck_policy_*); tests drive both with the same inputs and require identical outputs.
Nothing under camkifu_amd/ imports this module.

Each function names the reference lines it follows (paths under src/camkifu/).  Pinned by the reference's
own doctests where they exist (cyclic_permute x3, get_ordered_hull x3, norm: core/imgutil.py:244-249,
279-284, 337-338 -> tests/golden/reference_known_answers.json); everything else here is "parity unpinned"
(the reference holds no vector for it) and follows the source text operation by operation.
"""
import math
import sys

GSIZE = 19
E, B, W = 0, 1, 2


# ------------------------------------------------------------------------------------------------ geometry
def seg_from_hough(rho, theta, h, w):
    """core/imgutil.py:216-233 -> (x0, y0, x1, y1) ints"""
    rho, theta = float(rho), float(theta)
    a, b = math.cos(theta), math.sin(theta)
    x0, y0 = a * rho, b * rho
    ext = max(h, w)
    return (int(x0 + ext * (-b)), int(y0 + ext * a), int(x0 - ext * (-b)), int(y0 - ext * a))


def seg_len(s):
    """core/imgutil.py:497-502"""
    return math.sqrt((s[0] - s[2]) ** 2 + (s[1] - s[3]) ** 2)


def seg_theta(s):
    """core/imgutil.py:477-480"""
    return math.acos((s[2] - s[0]) / seg_len(s))


def line_angle(s, o):
    """core/imgutil.py:504-513"""
    ax, ay = (s[2] - s[0]) / seg_len(s), (s[3] - s[1]) / seg_len(s)
    bx, by = (o[2] - o[0]) / seg_len(o), (o[3] - o[1]) / seg_len(o)
    t = math.acos(round(ax * bx + ay * by, 10))
    return t if t <= math.pi / 2 else math.pi - t


def intersection(s, o):
    """core/imgutil.py:515-530"""
    q = (o[0] - s[0], o[1] - s[1])
    d1 = (s[2] - s[0], s[3] - s[1])
    d2 = (o[2] - o[0], o[3] - o[1])
    cross = float(d1[0] * d2[1] - d1[1] * d2[0])
    if abs(cross) < sys.float_info.epsilon:
        return None
    t1 = (q[0] * d2[1] - q[1] * d2[0]) / cross
    return int(s[0] + t1 * d1[0]), int(s[1] + t1 * d1[1])


def norm(p, q):
    """core/imgutil.py:331-340"""
    return math.sqrt((p[0] - q[0]) ** 2 + (p[1] - q[1]) ** 2)


def cyclic_permute(points):
    """core/imgutil.py:236-264"""
    best, idx = sys.maxsize, 0
    for i, p in enumerate(points):
        d = p[0] ** 2 + p[1] ** 2
        if d < best:
            best, idx = d, i
    return [tuple(points[(idx + k) % len(points)]) for k in range(len(points))]


def convex_hull(points):
    """cv2.convexHull(points) with its default orientation (clockwise on screen), collinear points dropped
    (library behaviour restated; monotone chain on the integer points)"""
    pts = sorted(set((int(p[0]), int(p[1])) for p in points))
    if len(pts) <= 2:
        return pts

    def cr(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])
    lo, up = [], []
    for p in pts:
        while len(lo) >= 2 and cr(lo[-2], lo[-1], p) <= 0:
            lo.pop()
        lo.append(p)
    for p in reversed(pts):
        while len(up) >= 2 and cr(up[-2], up[-1], p) <= 0:
            up.pop()
        up.append(p)
    return lo[:-1] + up[:-1]


def ordered_hull(points):
    """core/imgutil.py:267-288"""
    return cyclic_permute(convex_hull(points))


# ------------------------------------------------------------------------------------------------ board
class BoardLogic:
    """bf_auto.py:76-102 (after the image chain), 143-217; imgutil.py:38-68"""

    def __init__(self):
        self.lines, self.groups = [], []

    def group_intersections(self, h, w):
        ref = min(h, w)
        ordered = sorted(self.lines, key=seg_theta)
        for s1 in ordered:
            for s2 in reversed(ordered):
                if math.pi / 3 < line_angle(s1, s2):
                    p0 = intersection(s1, s2)
                    margin = -ref / 15
                    if 0 + margin < p0[0] < w - margin and 0 + margin < p0[1] < h - margin:
                        placed = False
                        for g in self.groups:
                            for p1 in g:
                                if (p0[0] - p1[0]) ** 2 + (p0[0] - p1[0]) ** 2 < (ref / 80) ** 2:
                                    g.append(p0)
                                    placed = True
                                    break
                            if placed:
                                break
                        if not placed:
                            self.groups.append([p0])
                else:
                    break

    @staticmethod
    def connect_clusters(groups, dist):
        gone = []
        for g0 in groups:
            into = None
            for p0 in g0:
                for g1 in groups:
                    if g0 is not g1 and not any(g1 is d for d in gone):
                        for p1 in g1:
                            if (p0[0] - p1[0]) ** 2 + (p0[0] - p1[0]) ** 2 < dist:
                                into = g1
                                break
                    if into:
                        break
                if into:
                    break
            if into:
                into.extend(g0)
                gone.append(g0)
        for d in gone:
            for k, g in enumerate(groups):
                if g is d:
                    del groups[k]
                    break

    def step(self, h, w, status, lines, counter, cur_hull):
        """-> (found, update, centers, (clusters, intersections) | None)"""
        if status != 0:
            return False, False, [], None
        self.lines.extend(seg_from_hough(r, t, h, w) for r, t in lines)
        if counter % 4:
            return False, False, [], None
        ref = min(h, w)
        self.group_intersections(h, w)
        while 4 < len(self.groups):
            before = len(self.groups)
            self.connect_clusters(self.groups, (ref / 50) ** 2)
            if len(self.groups) == before:
                break
        found, update, centers = False, False, []
        if len(self.groups) == 4:
            for g in self.groups:
                x = y = 0
                for p in g:
                    x += p[0]
                    y += p[1]
                centers.append((int(x / len(g)), int(y / len(g))))
            centers = ordered_hull(centers)
            found = True
            for i in range(len(centers)):
                if norm(centers[i - 1], centers[i]) < ref / 3:
                    found = False
                    break
            update = cur_hull is None
            if found and not update:
                for i in range(4):
                    if 5 < norm(centers[i], cur_hull[i]):
                        update = True
                        break
        stats = (len(self.groups), sum(len(g) for g in self.groups))
        self.lines, self.groups = [], []
        return found, update, centers, stats


# ------------------------------------------------------------------------------------------------ stones
def region_rows(i):
    """stone/nn_manager.py:92-126 with split 10, step 2: the last region is pulled back to rows 17, 18"""
    return (2 * i, 2 * i + 2) if GSIZE - 2 * i >= 2 else (GSIZE - 2, GSIZE)


def decode(label):
    """stone/nn_manager.py:246-254: four base-3 digits, least significant first"""
    out, k = [0] * 4, int(label)
    for i in reversed(range(4)):
        out[i] = int(k / 3 ** i)
        k %= 3 ** i
    return out


def cell_rect(r, c):
    """stone/stonesfinder.py:412-450 at the default grid: 20 px cells, the last row / column ends at 379"""
    return 20 * r, 20 * c, (379 if r == GSIZE - 1 else 20 * r + 20), (379 if c == GSIZE - 1 else 20 * c + 20)


class _Heat:
    """stone/sf_neural.py:198-244"""

    def __init__(self, color, conf, stamp):
        self.goal = self.energy = 3
        self.color, self.conf, self.stamp = color, conf, stamp
        self.checks = self.passed = 0

    def check(self, color, conf):
        self.checks += 1
        self.energy -= 1
        add = 0
        if color == self.color:
            self.passed += 1
            add = conf
        self.conf = (self.conf * self.checks + add) / (self.checks + 1)

    def valid(self):
        ok = 2 * self.goal / 3 <= self.passed + self.energy
        if not ok:
            self.energy = 0
            self.conf = 0.0
        return ok


class StonePolicy:
    """stone/sf_neural.py:37-195 driven by plain arrays.  `frame(...)` returns the requests the reference would
    send to its sink, in order: ('bulk', [(color, r, c), ...]) / ('suggest', (color, r, c)); `board_of()` is called
    for the goban as it is at that moment (after earlier requests of the frame were applied by the caller through
    the `apply` callback)."""

    def __init__(self, bg_init_frames=50):
        self.bg = bg_init_frames
        self.sampled = False
        self.targets = [[0] * GSIZE for _ in range(GSIZE)]
        self.heat = [[None] * GSIZE for _ in range(GSIZE)]

    @staticmethod
    def agitated(fgcount, r, c, ratio):
        a0, b0, a1, b1 = cell_rect(r, c)
        return (a1 - a0) * (b1 - b0) * ratio < fgcount[r][c]

    def frame(self, f, rl, rc, fgcount, board_of, apply):
        if f == 0 or f < self.bg:
            return
        if not self.sampled:
            grid = [[(0, 0.0)] * GSIZE for _ in range(GSIZE)]
            for i in range(10):
                for j in range(10):
                    (rs, re), (cs, ce) = region_rows(i), region_rows(j)
                    st = decode(rl[i][j])
                    for k in range(4):
                        grid[rs + k // 2][cs + k % 2] = (st[k], rc[i][j])
            moves = []
            for r in range(GSIZE):
                for c in range(GSIZE):
                    col, cf = grid[r][c]
                    if col != E and cf > 0.6:
                        moves.append((col, r, c))
                        self.heat[r][c] = _Heat(col, cf, f)
            if moves:
                apply(("bulk", moves))
            self.sampled = True
            return
        # mark_targets (72-83)
        for r in range(GSIZE):
            for c in range(GSIZE):
                if self.heat[r][c] is None and self.agitated(fgcount, r, c, 0.7):
                    self.targets[r][c] = (self.targets[r][c] + 5) % 256          # numpy uint8 wraps
        for r in range(GSIZE):
            for c in range(GSIZE):
                if self.targets[r][c] > 0:
                    self.targets[r][c] -= 1
        # select_targets (129-154)
        chosen = []
        for i in range(10):
            for j in range(10):
                (rs, re), (cs, ce) = region_rows(i), region_rows(j)
                cells = [(a, b) for a in range(rs, re) for b in range(cs, ce)]
                if not any(self.targets[a][b] > 15 for a, b in cells):
                    continue
                if any(self.agitated(fgcount, a, b, 0.5) for a, b in cells):
                    continue
                chosen.append((i, j))
                for a, b in cells:
                    self.targets[a][b] = 0
        # predict_moves (101-127); the reference collects into a set (iteration order = hash order, randomised per
        # process): here first-seen order, duplicates dropped
        moves = []
        if chosen:
            board = board_of()
            for i, j in chosen:
                if rc[i][j] < 0.6:
                    continue
                (rs, re), (cs, ce) = region_rows(i), region_rows(j)
                st = decode(rl[i][j])
                for k in range(4):
                    if st[k] == E:
                        continue
                    r, c = rs + k // 2, cs + k % 2
                    if board[r][c] == E:
                        m = (st[k], r, c, rc[i][j])
                        if m not in moves:
                            moves.append(m)
        # process_targets (85-99) with get_color_ratio (186-195)
        if moves:
            cnt = {B: 0, W: 0}
            for m in moves:
                cnt[m[0]] += 1
            if 0 in cnt.values():
                cnt[B] += 1
                cnt[W] += 1
            if abs(math.log(cnt[B] / cnt[W], 3)) < 1:
                for col, r, c, cf in moves:
                    self.heat[r][c] = _Heat(col, cf, f)
                if len(moves) == 1:
                    apply(("suggest", moves[0][:3]))
                else:
                    apply(("bulk", [m[:3] for m in moves]))
        # lookback (156-176)
        board = board_of()
        dels = []
        for r in range(GSIZE):
            for c in range(GSIZE):
                hp = self.heat[r][c]
                if hp is None or not 0 < hp.energy:
                    continue
                if hp.color != board[r][c]:
                    self.heat[r][c] = None
                    continue
                if 10 < f - hp.stamp:
                    hp.stamp = f
                    i, j = r // 2, c // 2                                # nn_cache.py:16-23
                    hp.check(decode(rl[i][j])[2 * (r % 2) + c % 2], rc[i][j])
                    if not hp.valid():
                        dels.append((E, r, c))
        if dels:
            apply(("bulk", dels))
        for r in range(GSIZE):
            for c in range(GSIZE):
                hp = self.heat[r][c]
                if hp is not None and hp.energy < -5:                    # _cleanup_heatmap (182-184)
                    self.heat[r][c] = None
        for r in range(GSIZE):
            for c in range(GSIZE):
                hp = self.heat[r][c]
                if hp is not None and hp.energy <= 0:                    # str(hp) on the debug canvas ages it (238-244)
                    hp.energy -= 1
