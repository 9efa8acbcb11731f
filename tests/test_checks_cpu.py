"""StonesFinder.check_against / check_lines / check_thickness / check_flow / first_line_lonelies
(reference stone/stonesfinder.py:597-783): the whole-array forms in camkifu_amd/stone/checks.py against
oracle/ora_checks.py -- each check stated from its meaning and answered by brute force over point sets (pairwise
king's-move distances, exact rational thresholds), a reading independent of the reference's control flow -- on random
gobans and sub-regions, plus hand-checked cases at the decision boundaries."""
import numpy as np
import pytest

from camkifu_amd.stone import checks
from oracle import ora_checks as ora

E, B, W = 'E', 'B', 'W'


def _goban(rng, density, blobs=0):
    g = np.full((19, 19), E, dtype=object)
    m = rng.random((19, 19)) < density
    g[m] = rng.choice([B, W], size=int(m.sum()))
    for _ in range(blobs):
        r, c, k = rng.integers(0, 15), rng.integers(0, 15), rng.integers(3, 7)
        g[r:r + k, c:c + k] = rng.choice([B, W])
    return g


def _region(rng):
    rs, cs = int(rng.integers(0, 12)), int(rng.integers(0, 12))
    return rs, int(rng.integers(rs + 1, 20)), cs, int(rng.integers(cs + 1, 20))


@pytest.mark.parametrize("seed", range(6))
def test_checks_equal_the_restatement_on_random_gobans(seed):
    rng = np.random.default_rng(100 + seed)
    for trial in range(60):
        ref = _goban(rng, rng.choice([0.0, 0.02, 0.2, 0.5]))
        st = ref.copy()
        flip = rng.random((19, 19)) < rng.choice([0.0, 0.05, 0.3])
        st[flip] = rng.choice([E, B, W], size=int(flip.sum()))
        if trial % 3 == 0:
            st = _goban(rng, 0.1, blobs=int(rng.integers(0, 3)))
        region = (0, 19, 0, 19) if trial % 4 == 0 else _region(rng)
        rs, re, cs, ce = region
        assert checks.check_against(st, ref, *region) == ora.check_against(st, ref, *region)
        assert checks.check_thickness(st, *region) == ora.check_thickness(st, *region)
        empty = ref == E
        assert checks.check_flow(st, empty, *region) == ora.check_flow(st, lambda r, c: bool(empty[r, c]), *region)
        grid = np.stack(np.meshgrid(10 + 20 * np.arange(19), 10 + 20 * np.arange(19), indexing="ij"), -1).astype(np.int16)
        neg = rng.random((19, 19)) < rng.choice([0.0, 0.03, 0.4])
        grid[neg] *= -1
        assert checks.check_lines(st, grid, *region) == ora.check_lines(st, grid, *region)
        if re <= 18 and ce <= 18:                           # re / ce are used as INDICES by the reference here
            assert sorted(checks.first_line_lonelies(st, ref, *region)) == sorted(ora.first_line_lonelies(st, ref, *region))


def test_the_oracle_depth_is_the_chessboard_distance_transform():
    """the oracle's pairwise king's-move depth against scipy's chamfer distance transform (chessboard metric) of the
    colour mask -- an implementation neither the oracle nor the product shares; scipy treats the array edge as the
    library does (never a zero)"""
    from scipy import ndimage
    rng = np.random.default_rng(5)
    for _ in range(20):
        g = np.full((19, 19), E, dtype=object)
        g[rng.random((19, 19)) < 0.8] = B
        if (g == B).all():
            g[4, 4] = E
        rs, re, cs, ce = _region(rng)
        sub = (g[rs:re, cs:ce] == B)
        depth = ora.depth_inside_colour(g, B, rs, re, cs, ce)
        if sub.all():
            assert all(d is None for d in depth.values())
            continue
        cdt = ndimage.distance_transform_cdt(sub, metric="chessboard")
        assert len(depth) == int(sub.sum())
        for (r, c), d in depth.items():
            assert d == cdt[r - rs, c - cs]


def test_boundaries_by_hand():
    g = np.full((19, 19), E, dtype=object)
    # check_against: needs MORE than 4 reference stones, and MORE than 81 % of them matched
    ref = g.copy()
    ref[0, :4] = B
    assert checks.check_against(ref, ref) == 0                      # 4 stones: undetermined
    ref[0, 4] = W
    assert checks.check_against(ref, ref) == 1                      # 5 of 5
    st = ref.copy()
    st[0, 0] = E
    assert checks.check_against(st, ref) == -1                      # 4 of 5 = 0.8: refused
    ref[1, :6] = W                                                  # 11 stones
    st = ref.copy()
    st[1, 0] = B
    st[1, 1] = E
    assert checks.check_against(st, ref) == 1                       # 9 of 11 = 0.818 > 0.81
    # check_thickness: a 5x5 block is 2 deep at its centre only (distance 3 > 2); a 4x4 block is not
    st = g.copy()
    st[5:9, 5:9] = B
    assert checks.check_thickness(st) == 0
    st[5:10, 5:10] = B
    assert checks.check_thickness(st) == -1
    st = g.copy()
    st[0:3, 0:3] = W                                                # in a corner the array edge is not a zero: 3x3 is enough
    assert checks.check_thickness(st) == -1 and ora.check_thickness(st) == -1
    st[2, 2] = E
    assert checks.check_thickness(st) == 0
    # check_flow: new stones may be unbalanced by one at most; stones on occupied points do not count
    empty = np.ones((19, 19), bool)
    st = g.copy()
    st[3, 3], st[3, 4] = B, B
    assert checks.check_flow(st, empty) == -1
    st[3, 5] = W
    assert checks.check_flow(st, empty) == 0
    empty[3, 5] = False
    assert checks.check_flow(st, empty) == -1
    # check_lines: more than 4 zones with a line, more than 90 % of them empty
    grid = np.full((19, 19, 2), 10, np.int16)
    grid[0, :10] = -10
    st = g.copy()
    assert checks.check_lines(st, grid) == 1
    st[0, 0] = B
    assert checks.check_lines(st, grid) == -1                       # 9 of 10 = 0.9: not MORE than 0.9
    grid[0, 4:10] = 10
    assert checks.check_lines(st, grid) == 0                        # 4 zones: undetermined
    # first_line_lonelies: a first-line stone alone within two lines; the far side only for a region ending at index 18
    st = g.copy()
    st[0, 9] = B
    assert checks.first_line_lonelies(st, g) == [(0, 9)]
    ref = g.copy()
    ref[2, 11] = W
    assert checks.first_line_lonelies(st, ref) == []
    st = g.copy()
    st[18, 5] = W
    assert checks.first_line_lonelies(st, g, 0, 18, 0, 18) == [(18, 5)]


def test_finder_methods_delegate(monkeypatch):
    from camkifu_amd.stone import stonesfinder as sfm
    names = [n for n in ("check_against", "check_lines", "check_thickness", "check_flow", "first_line_lonelies")]
    assert all(callable(getattr(sfm.StonesFinder, n)) for n in names)
