"""Host-side policy of the stones finder, against the reference's rules (stone/sf_neural.py:26-244,
stone/stonesfinder.py:178-349): acceptance threshold, colour-ratio veto, single move -> suggest /
several -> bulk_update, target marking and selection from the foreground mask, the three-check
lookback with HeatPoint's arithmetic, and the deletion watch.  No GPU: the classifier's answers are
scripted through a fake NNCache."""
import math

import numpy as np
import pytest

from camkifu_amd.controller import ControllerHeadless
from camkifu_amd.core.exceptions import CorrectionWarning, DeletedError
from camkifu_amd.golib_shim import B, E, W, Move, NP_TYPE
from camkifu_amd.stone import sf_neural
from camkifu_amd.stone.sf_neural import HeatPoint, SfNeural


class _Ctx:
    """the two calls the constructor / background model make"""
    def mog2_create(self, h, w):
        return 0

    def cnn_set_weights(self, w):
        pass


class _VM:
    imqueue = None
    board_finder = None
    current_video = None

    def __init__(self):
        self.controller = ControllerHeadless()


class _Cache:
    """scripted classifier: board = (19,19) array of E/B/W, conf = scalar or (10,10) per region"""
    def __init__(self, manager, board, conf=0.9):
        self.m, self.board = manager, board
        self.conf = np.full((10, 10), conf, float) if np.isscalar(conf) else conf

    def predict_4_stones(self, i, j):
        rs, re, cs, ce = self.m._subregion(i, j)
        return self.board[rs:re, cs:ce].copy(), float(self.conf[i, j])

    def predict_stone(self, r, c):
        i, j = self.m.get_region_indices(r, c)
        return self.board[r, c], float(self.conf[i, j])

    def predict_all_stones(self):
        out = np.ndarray((19, 19, 2), dtype=object)
        for i in range(10):
            for j in range(10):
                rs, re, cs, ce = self.m._subregion(i, j)
                sq, cf = self.predict_4_stones(i, j)
                out[rs:re, cs:ce, 0] = sq
                out[rs:re, cs:ce, 1] = cf
        return out


@pytest.fixture
def sf():
    f = SfNeural(_VM(), ctx=_Ctx())
    f.goban_img = np.zeros((380, 380, 3), np.uint8)
    f._fg = np.zeros((380, 380), np.uint8)
    return f


def board_of(stones):
    b = np.full((19, 19), E, dtype=object)
    for color, r, c in stones:
        b[r, c] = color
    return b


def test_predict_all_accepts_only_confident_stones(sf):
    conf = np.full((10, 10), 0.9)
    conf[0, 0] = 0.6                                     # not > 0.6: region (0,0) is discarded
    sf.cache = _Cache(sf.manager, board_of([(B, 0, 0), (W, 5, 5), (B, 18, 18)]), conf)
    sf.predict_all()
    got = sf.get_stones()
    assert got[5, 5] == W and got[18, 18] == B and got[0, 0] == E
    assert sf.heatmap[5, 5].color == W and sf.heatmap[5, 5].energy == 3 and sf.heatmap[0, 0] is None
    assert [repr(m) for m in sf.vmanager.controller.kifu.moves] == ["W[F14]", "B[T1]"]


def test_color_ratio_veto_and_single_move_suggest(sf):
    assert SfNeural.get_color_ratio([(B, 0, 0, 1.0)]) == pytest.approx(abs(math.log(2 / 1, 3)))     # zero count: +1 to both
    assert SfNeural.get_color_ratio([(B, 0, 0, 1), (W, 1, 1, 1)]) == 0
    assert SfNeural.get_color_ratio([(B, i, 0, 1) for i in range(3)] + [(W, 9, 9, 1)]) == pytest.approx(1.0)
    # three blacks for one white: |log3(3)| = 1 is NOT < 1 -> the whole batch is dropped
    sf.cache = _Cache(sf.manager, board_of([(B, 0, 0), (B, 0, 1), (B, 1, 0), (W, 1, 1)]))
    sf.targets[0:2, 0:2] = 20
    sf.process_targets()
    assert (sf.get_stones() == E).all() and (sf.targets[0:2, 0:2] == 0).all()
    # a single stone goes through suggest(): append + auto_save, and gets a heat point
    sf.cache = _Cache(sf.manager, board_of([(W, 4, 6)]))
    sf.targets[4, 6] = 16
    sf.process_targets()
    assert sf.get_stones()[4, 6] == W and sf.heatmap[4, 6].color == W
    # low confidence regions are ignored; a colour change of an existing stone is only reported
    sf.cache = _Cache(sf.manager, board_of([(B, 4, 6), (B, 10, 10)]), 0.5)
    sf.targets[4, 6] = sf.targets[10, 10] = 16
    sf.process_targets()
    assert sf.get_stones()[4, 6] == W and sf.get_stones()[10, 10] == E


def test_targets_follow_the_foreground(sf):
    x0, y0, x1, y1 = sf.getrect(7, 3)
    area = (x1 - x0) * (y1 - y0)
    zone = np.zeros(area, np.uint8)
    zone[: int(area * 0.7) + 1] = 255                                       # just over 70 % of the zone
    sf._fg[x0:x1, y0:y1] = zone.reshape(x1 - x0, y1 - y0)
    assert sf.is_agitated(7, 3, sf._fg) and not sf.is_agitated(7, 4, sf._fg)
    for k in range(4):
        sf.mark_targets()                                                   # +5 then -1 per frame
    assert sf.targets[7, 3] == 16 and sf.targets.sum() == 16
    assert sf.select_targets() == []                                        # hot, but still agitated (ratio 0.5)
    sf._fg[:] = 0
    assert sf.select_targets() == [(3, 1)] and sf.targets.sum() == 0       # calm now: region (7//2, 3//2), reset
    # a watched location (heat point) is not marked
    sf.heatmap[7, 3] = HeatPoint(B, 0.9, 0)
    sf._fg[x0:x1, y0:y1] = 255
    sf.mark_targets()
    assert sf.targets[7, 3] == 0


def test_heatpoint_arithmetic():
    hp = HeatPoint(B, 0.9, stamp=100)
    hp.check(B, 0.8)                                    # pass: conf = (0.9*1 + 0.8) / 2
    assert hp.nb_passed == 1 and hp.energy == 2 and hp.confidence == pytest.approx(0.85)
    hp.check(W, 0.99)                                   # fail: conf = (0.85*2 + 0) / 3
    assert hp.confidence == pytest.approx(1.7 / 3) and hp.is_valid()        # 2 <= 1 + 1
    hp.check(W, 0.99)
    assert not hp.is_valid() and hp.energy == 0 and hp.confidence == 0.0    # 2 <= 1 + 0 fails: cancelled
    assert [repr(hp) for _ in range(5)] == ["0"] * 5 and not hp.is_cold()   # repr ages an exhausted point
    assert repr(hp) == "" and hp.is_cold()


def test_lookback_cancels_a_stone_that_stops_being_seen(sf):
    sf.cache = _Cache(sf.manager, board_of([(B, 3, 3)]))
    sf.total_f_processed = 60
    sf.predict_all()
    assert sf.get_stones()[3, 3] == B
    sf.cache = _Cache(sf.manager, board_of([]))          # the classifier now sees nothing there
    seen = []
    for f in range(61, 120):
        sf.total_f_processed = f
        sf.lookback()
        seen.append(sf.get_stones()[3, 3])
    # checks happen when more than 10 frames passed since the last stamp: frames 71 and 82;
    # after two failures 2 <= 0 + 1 is false -> deleted at the second check
    assert seen[71 - 61 - 1] == B and seen[82 - 61 - 1] == B and seen[82 - 61] == E
    assert sf.heatmap[3, 3] is None or sf.heatmap[3, 3].energy <= 0
    # a stone changed by somebody else just drops its heat point
    sf.heatmap[9, 9] = HeatPoint(W, 0.9, 0)
    sf.lookback()
    assert sf.heatmap[9, 9] is None


def test_deletion_watch(sf):
    """stonesfinder.py:178-245: a user deletion locks the intersection; once 50 calm frames have been
    averaged, a suggestion is refused unless the zone changed by 40 grey levels per pixel on average"""
    sf.nb_del_samples = 4
    sf.goban_img[:] = 100
    sf.corrected(Move(NP_TYPE, (B, 6, 2)), None)         # user deleted the stone at row 6, col 2
    sf._learn()
    assert sf.deleted == {(6, 2): 3}
    with pytest.raises(DeletedError, match="too recently"):
        sf._check_dels(6, 2)
    x0, y0, x1, y1 = sf.getrect(6, 2)
    sf._fg[x0:x1, y0:y1] = 255                           # every pixel foreground: the frame is not sampled
    sf._learn()
    assert sf.deleted[(6, 2)] == 3
    sf._fg[x0, y0] = 0                                   # reference quirk: one calm pixel is enough
    for _ in range(3):
        sf._learn()
    assert sf.deleted[(6, 2)] == 0 and np.allclose(sf.saved_bg[x0:x1, y0:y1], 100.0)
    with pytest.raises(DeletedError, match="not changed enough"):
        sf.suggest(B, 6, 2, doprint=False)
    sf.goban_img[x0:x1, y0:y1] = 113                     # 3 channels x 13 = 39 per pixel: still locked
    with pytest.raises(DeletedError):
        sf._check_dels(6, 2)
    sf.goban_img[x0:x1, y0:y1] = 114                     # 42 per pixel: unlocked
    sf.suggest(B, 6, 2, doprint=False)
    assert (6, 2) not in sf.deleted and sf.get_stones()[6, 2] == B
    # bulk_update sends what it can and reports the locked locations
    sf.corrected(Move(NP_TYPE, (W, 1, 1)), Move(NP_TYPE, (W, 2, 2)))        # a moved stone locks its origin
    sf._learn()
    with pytest.raises(DeletedError) as ei:
        sf.bulk_update([(W, 1, 1), (W, 12, 12)])
    assert len(ei.value.locations) == 1 and sf.get_stones()[12, 12] == W and sf.get_stones()[1, 1] == E
    # a correction that adds a stone is not learnt from: reported
    sf.corrected(None, Move(NP_TYPE, (B, 5, 5)))
    with pytest.raises(CorrectionWarning):
        sf._learn()
