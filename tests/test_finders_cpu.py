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


def test_finders_take_the_host_applications_base_classes(tmp_path):
    """camkifu_amd/host.py: when CamKifu is importable the drop-ins inherit ITS BoardFinder / StonesFinder (what
    "drop into vmanager.py unchanged" means); here a stand-in package plays the host application"""
    import os
    import subprocess
    import sys
    import textwrap
    pkg = tmp_path / "camkifu"
    for sub in ("", "board", "stone"):
        (pkg / sub).mkdir(exist_ok=True)
        (pkg / sub / "__init__.py").write_text("")
    (pkg / "board" / "boardfinder.py").write_text(textwrap.dedent("""
        import collections
        class Corners:
            hull = None
            frame = None
            def __init__(self): self.pts = []
            def clear(self): self.pts, self.hull = [], None
            def submit(self, p):
                self.pts.append(tuple(p))
                if len(self.pts) == 4: self.hull = list(self.pts)
        class BoardFinder:
            HOST = True
            def __init__(self, vmanager):
                self.vmanager, self.total_f_processed, self.corners = vmanager, 0, Corners()
                self.metadata, self.mtx, self.shown = collections.defaultdict(list), None, 0
            def _show(self, *a, **k): self.shown += 1
            def _doframe(self, frame): self.corners.frame = frame; self.hit = self._detect(frame)
    """))
    (pkg / "stone" / "stonesfinder.py").write_text(textwrap.dedent("""
        class StonesFinder:
            HOST = True
            def __init__(self, vmanager, learn_bg=True):
                self.vmanager, self.total_f_processed, self.bg_init_frames = vmanager, 0, 3
                self._fg = None
            def get_foreground(self): return self._fg
            def getrect(self, r, c, cursor=1.0): return 20 * r, 20 * c, min(20 * r + 20, 379), min(20 * c + 20, 379)
    """))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import numpy as np
        from camkifu_amd import host
        from camkifu_amd.board.bf_auto import BoardFinderAuto
        from camkifu_amd.stone.sf_neural import SfNeural
        import camkifu.board.boardfinder as hb, camkifu.stone.stonesfinder as hs
        assert host.in_host() and issubclass(BoardFinderAuto, hb.BoardFinder) and issubclass(SfNeural, hs.StonesFinder)
        bf = BoardFinderAuto(object())
        frame = np.zeros((480, 640, 3), np.uint8)
        lines = np.array([[100, 0.05], [540, 0.08], [60, 1.55], [420, 1.62]], np.float32)
        hit = bf._detect(frame, record=dict(status=0, n_lines=4, lines=lines))
        assert hit and bf.corners.hull is not None and len(bf.corners.hull) == 4 and bf.shown == 1
        class Ctx:
            def cnn_set_weights(self, w): pass
        sf = SfNeural(object(), ctx=Ctx())
        assert sf.HOST and sf.policy is not None and sf.bg_init_frames == 3
        # SfContours on the host's base: the zone table comes from the host's getrect, one library call per find_stones
        from camkifu_amd.stone.sf_contours import SfContours
        from camkifu_amd.golib_shim import B, W, E
        assert issubclass(SfContours, hs.StonesFinder)
        class Rec:
            def contour_stones(self, img, fg, rects, rs, re, cs, ce):
                self.seen = (img.shape, fg.shape, np.asarray(rects).shape, tuple(np.asarray(rects)[18, 18]), rs, re, cs, ce)
                out = np.zeros((19, 19), np.uint8); out[3, 4] = 1; out[5, 6] = 2
                return out
        rec = Rec()
        sc = SfContours(object(), ctx=rec)
        sc._fg = np.zeros((380, 380), np.uint8)
        got = sc.find_stones(np.zeros((380, 380, 3), np.uint8), rs=2, re=9)
        assert rec.seen == ((380, 380, 3), (380, 380), (19, 19, 4), (360, 360, 379, 379), 2, 9, 0, 19)
        assert got[3, 4] == B and got[5, 6] == W and got[0, 0] == E and got.dtype == object
        sc.total_f_processed = 5
        sc._find(np.zeros((380, 380, 3), np.uint8))
        assert sc.last_stones is not None and sc.last_stones[5, 6] == W
        print("host bases ok")
    """)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([str(tmp_path), root]))
    env.pop("CAMKIFU_AMD_STANDALONE", None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0 and "host bases ok" in out.stdout, out.stderr[-2000:]
