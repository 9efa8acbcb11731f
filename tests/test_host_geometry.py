"""Host-side mirror of the reference's geometry / codec helpers against the reference's own
doctests and known answers (tests/golden/reference_known_answers.json)."""
import json
import math
import os

import numpy as np

from camkifu_amd.core import imgutil
from camkifu_amd.stone.nn_manager import NNManager
from camkifu_amd.stone.stonesfinder import PosGrid, StonesFinder

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "reference_known_answers.json")))


def _t(lst):
    return [tuple(p) for p in lst]


def test_cyclic_permute_doctests():
    for case in GOLD["cyclic_permute"]:
        assert imgutil.cyclic_permute(_t(case["in"])) == _t(case["out"])


def test_get_ordered_hull_doctests():
    for case in GOLD["get_ordered_hull"]:
        assert imgutil.get_ordered_hull(_t(case["in"])) == _t(case["out"])
    # a point inside the quadrilateral, duplicates and collinear points do not survive
    pts = [(5, 367), (126, 96), (514, 92), (638, 364), (300, 200), (126, 96), (320, 94)]
    assert len(imgutil.get_ordered_hull(pts)) == 4
    assert len(imgutil.get_ordered_hull([(0, 0), (5, 5), (10, 10), (3, 3)])) == 2


def test_norm_doctest():
    g = GOLD["norm"]
    assert "{:.6f}".format(imgutil.norm(g["p1"], g["p2"])) == g["fmt6"]


def test_segment_helpers():
    s = imgutil.segment_from_hough((100.0, 0.0), (480, 640))         # vertical line x = 100
    assert s.coords == (100, 640, 100, -640)
    assert abs(s.theta - math.pi / 2) < 1e-12
    h = imgutil.segment_from_hough((50.0, math.pi / 2), (480, 640))  # horizontal line y = 50
    # int() truncation of 50 -/+ 640*cos(pi/2) (= 50 -/+ 4e-14): the reference quirk parity keeps
    assert h.coords[1] == 50 and h.coords[3] == 49
    assert abs(s.line_angle(h) - math.pi / 2) < 2e-3
    assert s.intersection(imgutil.Segment((0, 50, 640, 50))) == (100, 50)
    assert s.intersection(imgutil.Segment((200, 0, 200, 10))) is None
    # truncation toward zero, not rounding
    d = imgutil.segment_from_hough((10.7, 0.3), (100, 100))
    x0, y0 = math.cos(0.3) * 10.7, math.sin(0.3) * 10.7
    assert d.coords[0] == int(x0 - 100 * math.sin(0.3)) and d.coords[1] == int(y0 + 100 * math.cos(0.3))


def test_connect_clusters_uses_x_only():
    groups = [[(10, 0)], [(11, 500)], [(300, 0)]]
    imgutil.connect_clusters(groups, 25)
    assert len(groups) == 2            # (10,0) and (11,500) merge although 500 px apart in y
    assert imgutil.within_margin((5, 5), (0, 0, 10, 10), 1)
    assert not imgutil.within_margin((0, 5), (0, 0, 10, 10), 0)


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


def test_group_intersections_matches_the_plain_double_loop():
    """the bisect membership test in BoardFinderAuto.group_intersections is a pure speed-up: same
    groups, same order as the reference's any(...) formulation (bf_auto.py:143-172)"""
    import math
    import random
    from camkifu_amd.board.bf_auto import BoardFinderAuto
    from camkifu_amd.core import imgutil

    class _VM:
        imqueue = None

    def plain(lines, shape):
        length_ref = min(shape[0], shape[1])
        margin, thresh = -length_ref / 15, (length_ref / 80) ** 2
        groups = []
        ordered = sorted(lines, key=lambda s: s.theta)
        for s1 in ordered:
            for s2 in reversed(ordered):
                if not (math.pi / 3 < s1.line_angle(s2)):
                    break
                p0 = s1.intersection(s2)
                if not imgutil.within_margin(p0, (0, 0, shape[1], shape[0]), margin):
                    continue
                for g in groups:
                    if any((p0[0] - p1[0]) ** 2 + (p0[0] - p1[0]) ** 2 < thresh for p1 in g):
                        g.append(p0)
                        break
                else:
                    groups.append([p0])
        return groups
    rng = random.Random(5)
    for trial in range(200):
        shape = (480, 640, 3)
        lines = []
        for _ in range(rng.randint(2, 24)):
            theta = rng.choice([0.03, 0.05, 1.55, 1.6, 1.58, 3.1, 0.8]) + rng.uniform(-0.02, 0.02)
            lines.append(imgutil.segment_from_hough((rng.uniform(-300, 600), theta), shape[:2]))
        bf = BoardFinderAuto(_VM(), ctx=False)
        bf.lines_accu = list(lines)
        bf.group_intersections(shape)
        assert bf.groups_accu == plain(lines, shape), trial
