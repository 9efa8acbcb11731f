"""BASELINE config 1: DetectionTest-style plumbing on a 640x480 synthetic clip, no GPU.
The finders' host logic runs exactly as in production, but every C-ABI call is answered by the
test-only OracleCtx (tests/stub_ctx.py)."""
import numpy as np
import pytest

from camkifu_amd import cvconf, synth
from camkifu_amd.controller import ControllerHeadless
from camkifu_amd.core.vmanager import VManagerBase, VManagerSeq
from camkifu_amd.golib_shim import E

from .stub_ctx import OracleCtx


@pytest.fixture(scope="module")
def clip():
    frames, corners, grids, moves = synth.video(70, 480, 640, seed=synth.SEED, new_stone_every=1000)
    # one fixed position for the whole clip (no new stones): every frame shows grids[0]
    stones = synth.random_stones(np.random.default_rng(5), density=0.25)
    frames = np.stack([synth.render(480, 640, stones, corners, seed=100 + f).numpy() for f in range(70)])
    return frames, corners, stones


def test_reflect_registry():
    bf = VManagerBase._reflect(None, cvconf.bfinders)
    sf = VManagerBase._reflect("SfNeural", cvconf.sfinders)
    assert bf.__name__ == "BoardFinderAuto" and sf.__name__ == "SfNeural"
    assert VManagerBase._reflect("None", cvconf.sfinders) is None
    assert VManagerBase._reflect("DoesNotExist", cvconf.sfinders).__name__ == "SfNeural"   # falls back to default


def test_sequential_detection_plumbing(clip, monkeypatch):
    frames, corners, stones = clip
    ctx = OracleCtx()
    from camkifu_amd import capi
    monkeypatch.setattr(capi, "Context", lambda device=0: ctx)
    # K7 without the HIP library: the oracle's solver (host float64, same system of equations)
    from oracle import oracle as ora
    monkeypatch.setattr(capi, "get_perspective_transform", ora.get_perspective_transform)

    controller = ControllerHeadless(video=frames)
    vm = VManagerSeq(controller)
    assert vm.bf_class.__name__ == "BoardFinderAuto" and vm.sf_class.__name__ == "SfNeural"
    vm.run()
    assert getattr(vm, "error", None) is None
    bf, sf = vm.board_finder, vm.stones_finder
    # board: found within the first accumulation rounds, corners within a few pixels of the truth
    assert bf.mtx is not None and bf.corners.hull is not None
    hull = np.array(bf.corners.hull, np.float64)
    assert np.abs(hull - corners).max() < 12, (hull, corners)
    assert bf.total_f_processed <= 9
    # stones: 50 background frames, then the one-off full-board assessment
    assert sf.total_f_processed > sf.bg_init_frames and sf.has_sampled
    # what reached the controller is exactly the accepted part of predict_all_stones on that frame
    f_assess = bf.total_f_processed + sf.bg_init_frames
    goban = ctx.warp_perspective(frames[f_assess], bf.mtx)
    labels, conf = ctx.cnn_predict(goban, want_y=False)
    expected = np.full((19, 19), E, dtype=object)
    sym = "EBW"
    for r in range(19):
        for c in range(19):
            if labels[0, r, c] and conf[0, r, c] > 0.6:
                expected[r, c] = sym[labels[0, r, c]]
    got = controller.get_stones()
    heat_cancelled = [(r, c) for r in range(19) for c in range(19) if got[r, c] != expected[r, c]]
    # lookback may cancel a stone later on; nothing else may differ
    assert all(got[r, c] == E for r, c in heat_cancelled)
    assert (expected != E).sum() == len(controller.kifu.moves) + len(heat_cancelled)


def test_board_fold_matches_finder(clip):
    """BoardFold replays BoardFinderAuto's temporal logic on records: same corners as the
    per-frame finder fed the same per-frame results"""
    frames, corners, _ = clip
    from camkifu_amd.pipeline import BoardFold
    ctx = OracleCtx()
    recs = ctx.board_detect(frames[:8])
    fold = BoardFold(480, 640)
    for r in recs:
        fold.step(r)
    assert fold.finder.corners.hull is not None
    assert np.abs(np.array(fold.finder.corners.hull, np.float64) - corners).max() < 12


def test_kifu_checker_scoring(tmp_path):
    from camkifu_amd.golib_shim import Kifu, Move, NP_TYPE
    from camkifu_amd.kifu_checker import KifuChecker, report
    ref = Kifu()
    for k, (col, r, c) in enumerate([('B', 3, 3), ('W', 15, 15), ('B', 3, 15), ('W', 15, 3)]):
        ref.append(Move(NP_TYPE, (col, r, c)))
    path = str(tmp_path / "ref.sgf")
    ref.save(path)
    again = Kifu(sgffile=path)
    assert [repr(m) for m in again.moves] == [repr(m) for m in ref.moves]       # SGF round trip
    found = Kifu()
    for m in ref.moves[:3]:
        found.append(Move(NP_TYPE, (m.color, m.y, m.x)))
    mt = KifuChecker(path).check(found)
    assert abs(mt.ratio() - 2 * 3 / 7) < 1e-12
    assert report("x", mt, 1.0).startswith("[x: 85.7% in")
    with pytest.raises(AssertionError):
        bad = Kifu()
        bad.append(Move(NP_TYPE, ('W', 3, 3)))
        KifuChecker(path, failfast=True).check(bad)
