"""Host-side mirror of the reference's geometry / codec helpers against the reference's own
doctests and known answers (tests/golden/reference_known_answers.json)."""
import json
import math
import os

import numpy as np

from camkifu_amd import capi
from camkifu_amd.stone.nn_manager import NNManager
from oracle import ora_logic as ol
from camkifu_amd.stone.stonesfinder import PosGrid, StonesFinder

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_known_answers.json")))


def _t(lst):
    return [tuple(p) for p in lst]


def test_cyclic_permute_doctests():
    for case in GOLD["cyclic_permute"]:
        assert ol.cyclic_permute(_t(case["in"])) == _t(case["out"])


def test_get_ordered_hull_doctests():
    """the reference's doctests, through the product (C++ ck_ordered_hull) and through the oracle"""
    for case in GOLD["get_ordered_hull"]:
        assert capi.ordered_hull(_t(case["in"])) == _t(case["out"])
        assert ol.ordered_hull(_t(case["in"])) == _t(case["out"])
    # a point inside the quadrilateral, duplicates and collinear points do not survive
    pts = [(5, 367), (126, 96), (514, 92), (638, 364), (300, 200), (126, 96), (320, 94)]
    assert len(capi.ordered_hull(pts)) == 4 and capi.ordered_hull(pts) == ol.ordered_hull(pts)
    assert len(capi.ordered_hull([(0, 0), (5, 5), (10, 10), (3, 3)])) == 2
    rng = np.random.default_rng(4)
    for _ in range(200):
        pts = [tuple(int(v) for v in p) for p in rng.integers(0, 40, (int(rng.integers(1, 12)), 2))]
        assert capi.ordered_hull(pts) == ol.ordered_hull(pts), pts


def test_norm_doctest():
    g = GOLD["norm"]
    assert "{:.6f}".format(ol.norm(g["p1"], g["p2"])) == g["fmt6"]


def test_segment_helpers():
    """the oracle's restatement of the segment helpers (the C++ fold is compared with it in test_fold_cpu.py)"""
    s = ol.seg_from_hough(100.0, 0.0, 480, 640)                      # vertical line x = 100
    assert s == (100, 640, 100, -640)
    assert abs(ol.seg_theta(s) - math.pi / 2) < 1e-12
    h = ol.seg_from_hough(50.0, math.pi / 2, 480, 640)               # horizontal line y = 50
    # int() truncation of 50 -/+ 640*cos(pi/2) (= 50 -/+ 4e-14): the reference quirk parity keeps
    assert h[1] == 50 and h[3] == 49
    assert abs(ol.line_angle(s, h) - math.pi / 2) < 2e-3
    assert ol.intersection(s, (0, 50, 640, 50)) == (100, 50)
    assert ol.intersection(s, (200, 0, 200, 10)) is None
    # truncation toward zero, not rounding
    d = ol.seg_from_hough(10.7, 0.3, 100, 100)
    x0, y0 = math.cos(0.3) * 10.7, math.sin(0.3) * 10.7
    assert d[0] == int(x0 - 100 * math.sin(0.3)) and d[1] == int(y0 + 100 * math.cos(0.3))


def test_connect_clusters_uses_x_only():
    groups = [[(10, 0)], [(11, 500)], [(300, 0)]]
    ol.BoardLogic.connect_clusters(groups, 25)
    assert len(groups) == 2            # (10,0) and (11,500) merge although 500 px apart in y


def test_goban_corners_bookkeeping():
    from camkifu_amd.board.boardfinder import GobanCorners
    gc = GobanCorners()
    gc.frame = np.zeros((480, 640, 3), np.uint8)
    for p in [(126, 96), (514, 92), (520, 100), (638, 364), (5, 367)]:
        gc.submit(p)                   # (520, 100) is closer than 480 / 5 to (514, 92): rejected while filling up
    assert gc.is_ready() and gc.hull == [(126, 96), (514, 92), (638, 364), (5, 367)]
    gc.submit((630, 370))              # complete: the nearest corner is replaced
    assert gc.hull == [(126, 96), (514, 92), (630, 370), (5, 367)]
    gc.clear()
    assert not gc.is_ready() and gc.hull is None
    assert GobanCorners([(0, 0), (5, 5), (10, 10), (3, 3)]).hull is None          # four points, no quadrilateral


def test_nnmanager_codec_and_geometry():
    m = NNManager()
    for k, v in GOLD["compute_stones"].items():
        assert list(m.compute_stones(int(k))) == v
    ci = m.class_indices()
    for k, v in GOLD["class_indices"].items():
        d, c = map(int, k.split(","))
        assert list(ci[d, c]) == v
    assert (m.split, m.step, m.nb_classes, m.r_width, m.c_width) == (10, 2, 81, 40, 40)
    origins = [m._get_rect_nn(*m._subregion(i, i))[0] for i in range(10)]
    assert origins == GOLD["patch_origins"]
    stones = np.full((19, 19), 'E', dtype=object)
    stones[17, 18] = 'B'
    assert m.compute_label(17, 19, 17, 19, stones) == 3
    assert m.get_region_indices(18, 18) == (9, 9)


def test_posgrid_and_getrect():
    g = PosGrid(380)
    for i in range(19):
        for j in range(19):
            assert tuple(g.mtx[i, j]) == (10 + 20 * i, 10 + 20 * j)

    class Bare(StonesFinder):
        def __init__(self):
            self._posgrid = PosGrid(380)
    sf = Bare()
    for k, v in GOLD["sf_getrect"].items():
        r, c = map(int, k.split(","))
        assert list(sf.getrect(r, c)) == v
