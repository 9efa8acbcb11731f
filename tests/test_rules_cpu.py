"""Go rules behind the headless SGF sink: captures, suicide, simple ko, and
the ordered fold replaying a game with captures into the move record."""
import numpy as np
import pytest

from camkifu_amd import synth
from camkifu_amd.controller import ControllerHeadless
from camkifu_amd.golib_shim import B, E, W, Move, NP_TYPE, Rule, StateError, Kifu
from camkifu_amd.kifu_checker import KifuChecker
from camkifu_amd.pipeline import StonesFold


def mv(color, r, c):
    return Move(NP_TYPE, (color, r, c))


def test_capture_suicide_and_ko():
    rule = Rule()
    # corner: white at (0,0) loses its two liberties
    assert rule.put(mv(W, 0, 0)) == [] and rule.put(mv(B, 0, 1)) == []
    assert rule.put(mv(B, 1, 0)) == [(W, 0, 0)] and rule.stones[0][0] == E and rule.deads[W] == 1
    with pytest.raises(StateError):
        rule.put(mv(W, 0, 0))                       # suicide: no liberty, captures nothing
    with pytest.raises(StateError):
        rule.put(mv(B, 0, 1))                       # occupied
    assert rule.stones[0][0] == E                   # a refused move leaves no trace
    # a two-stone group dies as one
    rule = Rule()
    for color, r, c in [(W, 5, 5), (W, 5, 6), (B, 4, 5), (B, 4, 6), (B, 6, 5), (B, 6, 6), (B, 5, 4)]:
        assert rule.put(mv(color, r, c)) == []
    assert sorted(rule.put(mv(B, 5, 7))) == [(W, 5, 5), (W, 6, 5)]      # (color, x, y): x = column
    # capture takes precedence over suicide, and the single-stone retake is a ko
    rule = Rule()
    for color, r, c in [(B, 3, 2), (B, 2, 3), (B, 4, 3), (W, 3, 5), (W, 2, 4), (W, 4, 4), (W, 3, 3)]:
        rule.put(mv(color, r, c))
    assert rule.put(mv(B, 3, 4)) == [(W, 3, 3)]     # lands without a liberty of its own, takes one stone
    with pytest.raises(StateError, match="ko"):
        rule.put(mv(W, 3, 3))
    rule.put(mv(W, 10, 10))                         # a move elsewhere lifts the ban
    assert rule.put(mv(W, 3, 3)) == [(B, 4, 3)]


def test_controller_with_rules_keeps_record_and_goban_apart():
    ctrl = ControllerHeadless(rules=True)
    for m in [mv(W, 0, 0), mv(B, 0, 1), mv(B, 1, 0)]:
        ctrl.pipe("append", m)
    assert ctrl.last_captured == [(W, 0, 0)] and ctrl.is_empty_blocking(0, 0)
    assert [repr(m) for m in ctrl.kifu.moves] == ["W[A19]", "B[B19]", "B[A18]"]        # the record keeps the prisoner's move
    assert ctrl.get_stones()[0, 0] == E and ctrl.get_stones()[0, 1] == B
    with pytest.raises(StateError):
        ctrl.pipe("append", mv(W, 0, 0))
    plain = ControllerHeadless()
    for m in [mv(W, 0, 0), mv(B, 0, 1), mv(B, 1, 0)]:
        plain.pipe("append", m)
    assert plain.get_stones()[0, 0] == W            # default: a plain mirror of the finders' reports


def _regions_of(grid):
    """(19, 19) uint8 grid -> the (10, 10) labels a perfect classifier would answer"""
    from camkifu_amd.stone import nn_manager as nm
    lab = np.zeros((10, 10), np.uint8)
    for i, rs in enumerate(nm.REGION_START):
        for j, cs in enumerate(nm.REGION_START):
            lab[i, j] = sum(int(v) * 3 ** k for k, v in enumerate(grid[rs:rs + 2, cs:cs + 2].reshape(4)))
    return lab


def _film(moves, positions, hand=6, calm=3, bg=3):
    """what the camera and the background model would report for a game: `bg` frames of the empty board, then per
    move `hand` frames during which a hand covers the point played and its neighbours (the board still shows the
    previous position, prisoners being taken off by the same hand) and `calm` frames showing the new position"""
    rl, fg = [], []
    shown = np.zeros((19, 19), np.uint8)
    for _ in range(bg + 1):
        rl.append(_regions_of(shown))
        fg.append(np.zeros((19, 19), np.int32))
    for (color, r, c), pos in zip(moves, positions):
        busy = np.zeros((19, 19), np.int32)
        busy[max(0, r - 1):r + 2, max(0, c - 1):c + 2] = 400
        for _ in range(hand):
            rl.append(_regions_of(shown))
            fg.append(busy)
        shown = pos
        for _ in range(calm):
            rl.append(_regions_of(shown))
            fg.append(np.zeros((19, 19), np.int32))
    return np.stack(rl), np.stack(fg)


@pytest.mark.parametrize("chunk", [None, 37])
def test_fold_replays_a_game_with_captures(chunk, tmp_path):
    """a random legal game (several captures) filmed move by move and folded by the stones policy: every stone is
    found once its zone has been agitated and has calmed down, prisoners leave the goban through the rule engine,
    and the record equals the game -- whether the film is folded in one run or in batches"""
    rng = np.random.default_rng(20161001)
    moves, positions, captured = synth.random_game(120, rng)
    assert sum(len(c) for c in captured) >= 5, "the fixture game must contain captures"
    rl, fg = _film(moves, positions)
    rc = np.full(rl.shape, 0.95)
    ctrl = ControllerHeadless(rules=True)
    fold = StonesFold(ctrl, bg_init_frames=3)
    step = chunk or len(rl)
    requests = []
    for k in range(0, len(rl), step):
        requests.extend(fold.run(rl[k:k + step], rc[k:k + step], fg[k:k + step]))
    assert [(m.color, m.y, m.x) for m in ctrl.kifu.moves] == moves
    assert all(kind == 1 for per_frame in requests for kind, _ in per_frame)          # one stone at a time: suggest()
    got = ctrl.get_stones()
    want = np.array([[E, B, W][v] for v in positions[-1].reshape(-1)], dtype=object).reshape(19, 19)
    assert (got == want).all()
    # scoring against the reference SGF, as DetectionTest does
    ref = Kifu()
    for color, r, c in moves:
        ref.append(mv(color, r, c))
    path = str(tmp_path / "ref.sgf")
    ref.save(path)
    assert KifuChecker(path).check(ctrl.kifu).ratio() == 1.0
    # without the rule engine the goban keeps the prisoners: it no longer matches the camera
    plain = ControllerHeadless()
    StonesFold(plain, bg_init_frames=3).run(rl, rc, fg)
    assert (plain.get_stones() != want).any()
