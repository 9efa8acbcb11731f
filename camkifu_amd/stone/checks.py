"""Consistency checks a stones finder can run on a candidate 19x19 result before submitting it
(reference: stone/stonesfinder.py:597-783; used by SfMeta, stone/sf_meta.py:57-58, 256, 432).

Each check answers -1 (refused), 0 (undetermined) or 1 (passed), like the reference's methods, and takes the same
(rs, re, cs, ce) sub-region arguments.  They are written over whole arrays: `stones` / `reference` are (19, 19)
arrays of the colour symbols 'E', 'B', 'W' (object or unicode dtype, as the finders hand them around).

    check_against      :597-632   the candidate agrees with > 81 % of the stones already on the goban (needs > 4 of them)
    check_lines        :634-673   > 90 % of the zones in which a grid line was found are empty in the candidate (needs > 4)
    check_thickness    :675-700   no stone sits more than 2 cells deep (chessboard metric) inside a blob of its colour
    check_flow         :702-736   the newly added stones are colour-balanced to within one
    first_line_lonelies:738-783   first-line stones with no neighbour within two lines (a list of (r, c))

check_thickness replaces cv2.distanceTransform(mask, DIST_C, 3) > 2: with the 3x3 chessboard mask that transform is
the exact chessboard distance to the nearest zero INSIDE the array (the library pads with a border it never treats
as zero), so "some distance exceeds 2" is "some 5x5 neighbourhood, clipped to the array, holds no zero"."""
import numpy as np

from ..golib_shim import gsize, E, B, W


def _sub(a, rs, re, cs, ce):
    return np.asarray(a, dtype=object)[rs:re, cs:ce]


def check_against(stones, reference, rs=0, re=gsize, cs=0, ce=gsize):
    st, ref = _sub(stones, rs, re, cs, ce), _sub(reference, rs, re, cs, ce)
    there = (ref == B) | (ref == W)
    refs = int(there.sum())
    if 4 < refs:
        matches = int((there & (st == ref)).sum())
        return 1 if 0.81 < matches / refs else -1
    return 0


def check_lines(stones, grid, rs=0, re=gsize, cs=0, ce=gsize):
    """grid: (19, 19, 2) as StonesFinder.get_intersections returns it (positions negated where a line was found)"""
    st = _sub(stones, rs, re, cs, ce)
    found = np.asarray(grid)[rs:re, cs:ce].astype(np.int64).sum(axis=2) < 0
    lines = int(found.sum())
    if 4 < lines:
        matches = int((found & (st == E)).sum())
        return 1 if 0.9 < matches / lines else -1
    return 0


def check_thickness(stones, rs=0, re=gsize, cs=0, ce=gsize):
    st = _sub(stones, rs, re, cs, ce)
    h, w = st.shape
    if h == 0 or w == 0:
        return 0
    for color in (B, W):
        solid = np.ones((h + 4, w + 4), bool)                  # beyond the array: never a zero
        solid[2:2 + h, 2:2 + w] = st == color
        deep = np.ones((h, w), bool)
        for dy in range(5):
            for dx in range(5):
                deep &= solid[dy:dy + h, dx:dx + w]
        if deep.any():                                          # a stone 2 cells deep inside its own colour: not Go
            return -1
    return 0


def check_flow(stones, empty, rs=0, re=gsize, cs=0, ce=gsize):
    """empty: (19, 19) bool, StonesFinder.is_empty of every intersection"""
    st = _sub(stones, rs, re, cs, ce)
    new = np.asarray(empty, bool)[rs:re, cs:ce] & (st != E)
    diff = int((new & (st == B)).sum()) - int((new & (st != B)).sum())
    return 0 if abs(diff) <= 1 else -1


def first_line_lonelies(stones, reference, rs=0, re=gsize, cs=0, ce=gsize):
    """The reference looks at row rs and row re (and column cs, column ce) when that INDEX is 0 or 18 -- re / ce are the
    exclusive ends, so the far side is only ever examined for a region that stops one line short of it.  Kept as is.
    -> list of (r, c), rows first then columns, each in ascending order (the reference iterates a set: order unspecified)"""
    st, ref = np.asarray(stones, dtype=object), np.asarray(reference, dtype=object)
    pos = []
    for r in (rs, re):
        if r in (0, gsize - 1):
            pos.extend((r, c) for c in range(cs, ce))
    for c in (cs, ce):
        if c in (0, gsize - 1):
            pos.extend((r, c) for r in range(rs, re))
    busy = (ref == B) | (ref == W) | (st == B) | (st == W)
    out, seen = [], set()
    for r, c in pos:
        if (r, c) in seen:
            continue
        seen.add((r, c))
        if st[r, c] in (B, W):
            around = busy[max(0, r - 2):min(gsize, r + 3), max(0, c - 2):min(gsize, c + 3)]
            if int(around.sum()) - 1 == 0:                      # nothing but the stone itself
                out.append((r, c))
    return out
