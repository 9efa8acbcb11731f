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


@pytest.mark.parametrize("linger", [0, 3])
def test_fold_replays_a_game_with_captures(linger, tmp_path):
    """positions of a random legal game (several captures) shown frame by frame; with linger > 0 the
    prisoners stay on the board for a few frames after the capture, as in real footage"""
    rng = np.random.default_rng(20161001)
    moves, positions, captured = synth.random_game(120, rng, cool=linger)
    assert sum(len(c) for c in captured) >= 5, "the fixture game must contain captures"
    ctrl = ControllerHeadless(rules=True)
    fold = StonesFold(ctrl)
    conf = np.ones((19, 19))
    for k, pos in enumerate(positions):
        shown = pos.copy()
        for j in range(max(0, k - linger + 1), k + 1):                 # prisoners of the last `linger` moves
            for col, r, c in (captured[j] if linger else ()):
                if shown[r, c] == 0:
                    shown[r, c] = 1 if col == B else 2
        for _ in range(2):                                             # every position is seen twice
            fold.step(shown, conf)
    for _ in range(2):
        fold.step(positions[-1], conf)
    assert [(m.color, m.y, m.x) for m in ctrl.kifu.moves] == moves
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
    fold2 = StonesFold(plain)
    for pos in positions:
        fold2.step(pos, conf)
    assert (plain.get_stones() != want).any()
