"""ORACLE (test infrastructure only): the consistency checks of a stones finder, stated from what each one MEANS and
answered by brute force over point sets -- a second, independent reading of
/root/reference/src/camkifu/stone/stonesfinder.py:597-783, not a walk through its control flow, and sharing no
formulation with the product's whole-array code (camkifu_amd/stone/checks.py).

    check_against        of the points where the goban already holds a stone, which fraction does the candidate agree on?
    check_lines          of the zones in which a grid line was seen, which fraction does the candidate leave empty?
    check_thickness      is some stone more than two king's moves away from every point of another symbol?
    check_flow           do the newly added stones of the two colours balance to within one?
    first_line_lonelies  first-line stones with no stone of either array within two king's moves

Distances are chessboard distances between points, looked up pairwise (no distance transform, no neighbourhood
generator).  Nothing from the reference is imported."""
from fractions import Fraction

gsize, E, B, W = 19, 'E', 'B', 'W'


def _points(rs, re, cs, ce):
    return [(r, c) for r in range(rs, re) for c in range(cs, ce)]


def _king(p, q):
    """moves a chess king needs from p to q"""
    return max(abs(p[0] - q[0]), abs(p[1] - q[1]))


def _verdict(good, total, needed, bar):
    """-1 / 0 / 1: undetermined unless MORE than `needed` cases, passed when the share of good ones EXCEEDS `bar`
    (exact rational comparison against the decimal the reference writes: 0.81, 0.9)"""
    if total <= needed:
        return 0
    return 1 if Fraction(good, total) > Fraction(bar) else -1


def check_against(stones, reference, rs=0, re=gsize, cs=0, ce=gsize):
    occupied = {p for p in _points(rs, re, cs, ce) if reference[p] != E}
    agreed = {p for p in occupied if stones[p] == reference[p]}
    return _verdict(len(agreed), len(occupied), 4, "0.81")


def check_lines(stones, grid, rs=0, re=gsize, cs=0, ce=gsize):
    """grid = get_intersections(img): the two coordinates of a zone are negated where a line was found"""
    with_line = {p for p in _points(rs, re, cs, ce) if int(grid[p][0]) + int(grid[p][1]) < 0}
    left_empty = {p for p in with_line if stones[p] == E}
    return _verdict(len(left_empty), len(with_line), 4, "0.9")


def depth_inside_colour(stones, color, rs, re, cs, ce):
    """for every point of the region holding `color`: king's moves to the nearest point OF THE REGION that holds
    something else (None when the region holds nothing else: the array's edge is not 'something else')"""
    pts = _points(rs, re, cs, ce)
    others = [q for q in pts if stones[q] != color]
    return {p: (min(_king(p, q) for q in others) if others else None) for p in pts if stones[p] == color}


def check_thickness(stones, rs=0, re=gsize, cs=0, ce=gsize):
    for color in (B, W):
        depths = depth_inside_colour(stones, color, rs, re, cs, ce).values()
        if any(d is None or d > 2 for d in depths):
            return -1
    return 0


def check_flow(stones, is_empty, rs=0, re=gsize, cs=0, ce=gsize):
    """is_empty(r, c) -> bool: the goban's view.  Only stones put on empty points are new."""
    new = [stones[p] for p in _points(rs, re, cs, ce) if stones[p] != E and is_empty(*p)]
    return 0 if abs(new.count(B) - new.count(W)) <= 1 else -1


def first_line_lonelies(stones, reference, rs=0, re=gsize, cs=0, ce=gsize):
    """The sides of the region that lie on the goban's first line -- where the reference tests the row INDICES rs and
    re, and the column indices cs and ce, against 0 and 18 (re / ce being the exclusive ends: the far side only counts
    for a region that stops at index 18).  On those sides, every stone of the candidate that has no stone, in the
    candidate or on the goban, within two king's moves."""
    board = _points(0, gsize, 0, gsize)
    anything = [q for q in board if stones[q] != E or reference[q] != E]
    side = {(r, c) for r in (rs, re) if r in (0, gsize - 1) for c in range(cs, ce)}
    side |= {(r, c) for c in (cs, ce) if c in (0, gsize - 1) for r in range(rs, re)}
    return [p for p in sorted(side)
            if stones[p] != E and not any(q != p and _king(p, q) <= 2 for q in anything)]
