"""ORACLE (test infrastructure only): plain-loop restatement of the reference's consistency checks, statement for
statement (/root/reference/src/camkifu/stone/stonesfinder.py:597-783, core/imgutil.py:360-379).  cv2.distanceTransform
(DIST_C, 3x3 mask) is restated as the two-pass chamfer it is; nothing from the reference is imported."""
import numpy as np

gsize, E, B, W = 19, 'E', 'B', 'W'


def check_against(stones, reference, rs=0, re=gsize, cs=0, ce=gsize):          # :597-632
    refs = matches = 0
    for r in range(rs, re):
        for c in range(cs, ce):
            if reference[r, c] in (B, W):
                refs += 1
                if stones[r, c] == reference[r, c]:
                    matches += 1
    if 4 < refs:
        return 1 if 0.81 < matches / refs else -1
    return 0


def check_lines(stones, grid, rs=0, re=gsize, cs=0, ce=gsize):                  # :634-673 (grid = get_intersections(img))
    lines = matches = 0
    for r in range(rs, re):
        for c in range(cs, ce):
            if int(grid[r, c][0]) + int(grid[r, c][1]) < 0:
                lines += 1
                if stones[r, c] == E:
                    matches += 1
    if 4 < lines:
        return 1 if 0.9 < matches / lines else -1
    return 0


def distance_c3(mask):
    """cv2.distanceTransform(mask, cv2.DIST_C, 3): two-pass 3x3 chamfer with all weights 1 over an image padded by a
    border of 'infinite' distance -- exact chessboard distance to the nearest zero pixel of the image"""
    h, w = mask.shape
    big = 10 ** 6
    d = np.full((h + 2, w + 2), big, np.int64)
    for y in range(h):
        for x in range(w):
            if mask[y, x] == 0:
                d[y + 1, x + 1] = 0
    for y in range(1, h + 1):
        for x in range(1, w + 1):
            if d[y, x]:
                d[y, x] = min(d[y, x], d[y - 1, x - 1] + 1, d[y - 1, x] + 1, d[y - 1, x + 1] + 1, d[y, x - 1] + 1)
    for y in range(h, 0, -1):
        for x in range(w, 0, -1):
            if d[y, x]:
                d[y, x] = min(d[y, x], d[y + 1, x + 1] + 1, d[y + 1, x] + 1, d[y + 1, x - 1] + 1, d[y, x + 1] + 1)
    return d[1:-1, 1:-1]


def check_thickness(stones, rs=0, re=gsize, cs=0, ce=gsize):                     # :675-700
    for color in (B, W):
        avatar = np.array([[1 if stones[r, c] == color else 0 for c in range(cs, ce)] for r in range(rs, re)], np.uint8)
        if avatar.size and 2 < distance_c3(avatar.reshape((re - rs, ce - cs))).max():
            return -1
    return 0


def check_flow(stones, is_empty, rs=0, re=gsize, cs=0, ce=gsize):               # :702-736 (is_empty(r, c) -> bool)
    moves = []
    for r in range(rs, re):
        for c in range(cs, ce):
            if is_empty(r, c) and stones[r, c] != E:
                moves.append(stones[r, c])
    diff = 0
    for mv in moves:
        diff += 1 if mv == B else -1
    return 0 if abs(diff) <= 1 else -1


def around(x, y, margin, xmin=None, xmax=None, ymin=None, ymax=None):            # imgutil.py:360-379
    for i in range(-margin, margin + 1):
        if (xmin is None or xmin <= x + i) and (xmax is None or x + i < xmax):
            for j in range(-margin, margin + 1):
                if i == j == 0:
                    continue
                if (ymin is None or ymin <= y + j) and (ymax is None or y + j < ymax):
                    yield x + i, y + j


def first_line_lonelies(stones, reference, rs=0, re=gsize, cs=0, ce=gsize):    # :738-783
    pos = set()
    for r in (rs, re):
        if r in (0, gsize - 1):
            for c in range(cs, ce):
                pos.add((r, c))
    for c in (cs, ce):
        if c in (0, gsize - 1):
            for r in range(rs, re):
                pos.add((r, c))
    lonelies = []
    for (r, c) in pos:
        if stones[r, c] in (B, W):
            alone = True
            for x, y in around(r, c, 2, xmin=0, xmax=gsize, ymin=0, ymax=gsize):
                if reference[x, y] in (B, W) or stones[x, y] in (B, W):
                    alone = False
                    break
            if alone:
                lonelies.append((r, c))
    return lonelies
