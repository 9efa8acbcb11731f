"""Frame sources and capture readers: .y4m container, the file_fps frame
skipping of CaptureReaderBase.skip (reference core/vmanager.py:511-525) and the lock-step
CaptureReader (core/vmanager.py:528-635).  CPU only: the I420->BGR conversion is answered by the
oracle here; tests/test_gpu_parity.py checks the HIP kernel against the same oracle."""
import threading

import numpy as np
import pytest

from camkifu_amd import cvconf, synth
from camkifu_amd.core import capture as cap


def _clip(tmp_path, n=40, h=16, w=24, fps=(30, 1)):
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, (n, h * w * 3 // 2), dtype=np.uint8)
    frames[:, 0] = np.arange(n)                     # frame id in the first luma byte
    path = str(tmp_path / "clip.y4m")
    cap.write_y4m(path, frames, h, w, fps)
    return path, frames


def test_y4m_round_trip_and_properties(tmp_path):
    path, frames = _clip(tmp_path, fps=(30000, 1001))
    c = cap.Y4MCapture(path)
    assert c.isOpened() and len(c) == 40
    assert c.get(cap.CAP_PROP_FRAME_WIDTH) == 24 and c.get(cap.CAP_PROP_FRAME_HEIGHT) == 16
    assert abs(c.get(cap.CAP_PROP_FPS) - 29.97) < 0.01 and c.get(cap.CAP_PROP_FRAME_COUNT) == 40
    for i in range(40):
        assert np.array_equal(c.read_raw(), frames[i])
    assert c.read_raw() is None
    c.set(cap.CAP_PROP_POS_FRAMES, 7.9)             # fractional index truncates
    assert c.read_raw()[0] == 7 and c.get(cap.CAP_PROP_POS_FRAMES) == 8
    assert np.array_equal(c.read_raw_batch([3, 30, 5]), frames[[3, 30, 5]])
    assert c.get(cap.CAP_PROP_POS_FRAMES) == 8      # batch reads leave the position alone


def test_y4m_frame_params_truncation_and_errors(tmp_path):
    h, w = 4, 6
    body = bytes(range(h * w * 3 // 2))
    good = tmp_path / "p.y4m"
    good.write_bytes(b"YUV4MPEG2 W6 H4 F25:1 C420mpeg2\nFRAME Ip\n" + body + b"FRAME\n" + body + b"FRAME\n" + body[:5])
    c = cap.Y4MCapture(str(good))
    assert c.isOpened() and len(c) == 2 and c.get(cap.CAP_PROP_FPS) == 25      # truncated third frame ignored
    assert bytes(c.read_raw()) == body
    for name, head in (("a", b"RIFF....\n"), ("b", b"YUV4MPEG2 W6 H4 C444\n"), ("c", b"YUV4MPEG2 W5 H4 C420\n"),
                       ("d", b"YUV4MPEG2 H4\n")):
        bad = tmp_path / (name + ".y4m")
        bad.write_bytes(head + b"FRAME\n" + body)
        cb = cap.Y4MCapture(str(bad))
        assert not cb.isOpened() and isinstance(cb.error, cap.Y4MError)
    assert not cap.Y4MCapture(str(tmp_path / "missing.y4m")).isOpened()


class _VM:
    """the two attributes a capture reader looks at"""
    def __init__(self, video):
        self.controller = type("C", (), {"video": video})()
        self.processes = []
        self.progress = []

    def vid_progress(self, p):
        self.progress.append(p)


def _ident(raw, h, w):
    return np.full((h, w, 3), raw[0], np.uint8)      # "decode" to the frame id


def test_file_reader_skips_to_file_fps(tmp_path):
    path, _ = _clip(tmp_path, n=40)
    rd = cap.CaptureReaderBase(cap.Y4MCapture(path, convert=_ident), _VM(path))
    seen = []
    while True:
        ok, img = rd.read(None)
        if not ok:
            break
        seen.append(int(img[0, 0, 0]))
    # skip() runs before EVERY read: +30/5 frames, then the read itself moves one further
    assert seen == [6, 13, 20, 27, 34] == cap.file_frame_indices(40, 30.0, cvconf.file_fps)
    assert cap.file_frame_indices(10, 30.0, 60) == [1, 3, 5, 7, 9]      # max(1, fps/rate)
    # the same capture under a non-file `video` (e.g. a camera index): no skipping
    rd2 = cap.CaptureReaderBase(cap.Y4MCapture(path, convert=_ident), _VM(0))
    assert [int(rd2.read(None)[1][0, 0, 0]) for _ in range(3)] == [0, 1, 2]
    assert rd2.get(cap.CAP_PROP_FRAME_COUNT) == 40                      # everything else is delegated


class _Proc:
    def __init__(self):
        self.got = []
        self.active = True

    def ready_to_read(self):
        return self.active


def test_lock_step_reader_serves_every_processor_the_same_frames(tmp_path):
    path, _ = _clip(tmp_path, n=60)
    vm = _VM(path)
    rd = cap.CaptureReader(cap.Y4MCapture(path, convert=_ident), vm)
    rd.sleep_time = 0.001
    procs = [_Proc(), _Proc()]
    vm.processes = [type("VT", (), {"processor": p, "ready_to_read": p.ready_to_read})() for p in procs]

    def consume(p, delay):
        import time
        while True:
            ok, img = rd.read(p)
            if not ok:
                p.active = False
                return
            p.got.append(int(img[0, 0, 0]))
            img[:] = 255                                # consumers may scribble on their copy
            time.sleep(delay)
    ts = [threading.Thread(target=consume, args=(p, d)) for p, d in zip(procs, (0.0, 0.003))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=30)
    assert not any(t.is_alive() for t in ts)
    # first frame is read unskipped by init_buffer; afterwards consume() skips like the base reader
    expected = [0] + cap.file_frame_indices(60, 30.0, cvconf.file_fps, start=1)
    assert procs[0].got == expected and procs[1].got == expected
    assert vm.progress and vm.progress[-1] <= 100
    # unsync releases waiting readers with the `unsynced` marker
    rd2 = cap.CaptureReader(cap.Y4MCapture(path, convert=_ident), vm)
    rd2.unsync_threads(True)
    assert rd2.read(procs[0]) == (False, cvconf.unsynced)


def test_i420_oracle_against_float_bt601(ora):
    rng = np.random.default_rng(11)
    h, w = 32, 48
    buf = rng.integers(0, 256, h * w * 3 // 2, dtype=np.uint8)
    got = ora.i420_to_bgr(buf, h, w).astype(np.float64)
    y = np.maximum(buf[:h * w].reshape(h, w).astype(np.float64) - 16, 0)
    u = np.repeat(np.repeat(buf[h * w:h * w * 5 // 4].reshape(h // 2, w // 2), 2, 0), 2, 1).astype(np.float64) - 128
    v = np.repeat(np.repeat(buf[h * w * 5 // 4:].reshape(h // 2, w // 2), 2, 0), 2, 1).astype(np.float64) - 128
    ref = np.stack([1.164 * y + 2.018 * u, 1.164 * y - 0.391 * u - 0.813 * v, 1.164 * y + 1.596 * v], -1)
    assert np.abs(got - np.clip(np.round(ref), 0, 255)).max() <= 1
    # grey ramp: studio range 16..235 maps onto 0..255, chroma neutral
    ramp = np.concatenate([np.repeat(np.array([16, 126, 235, 255], np.uint8), 4), np.full(8, 128, np.uint8)])
    assert ora.i420_to_bgr(ramp, 4, 4)[:, 0, 0].tolist() == [0, 128, 255, 255]
    # and a rendered scene survives BGR -> I420 -> BGR within chroma-subsampling error
    fr = synth.scene(96, 128, seed=2)["frame"].numpy()
    back = ora.i420_to_bgr(synth.bgr_to_i420(fr), 96, 128).astype(np.int32)
    assert np.median(np.abs(back - fr)) <= 2


def test_threaded_vmanager_lock_step_over_y4m(tmp_path, ora, monkeypatch):
    """VManager spawns one daemon thread per finder; over a file both see the same frames"""
    from camkifu_amd.controller import ControllerHeadless
    from camkifu_amd.core import vmanager as vmod
    from camkifu_amd.core.video import VidProcessor
    frames = [synth.bgr_to_i420(np.full((16, 24, 3), 3 * i, np.uint8)) for i in range(50)]
    path = str(tmp_path / "v.y4m")
    cap.write_y4m(path, frames, 16, 24)
    monkeypatch.setattr(vmod, "open_capture", lambda video: cap.Y4MCapture(video, convert=ora.i420_to_bgr))
    seen = {}

    class Rec(VidProcessor):
        def __init__(self, vm):
            super().__init__(vm)
            seen[type(self).__name__] = self.got = []

        def _doframe(self, frame):
            self.got.append(int(frame[0, 0, 1]))

    class RecB(Rec):
        pass

    class RecS(Rec):
        pass
    vm = vmod.VManager(ControllerHeadless(video=path))
    vm.bf_class, vm.sf_class = RecB, RecS
    vm.start()
    import time
    t0 = time.time()
    while time.time() - t0 < 30 and not (vm.hasrun and not vm.is_processing() and len(seen) == 2 and seen["RecB"]):
        time.sleep(0.02)
    vm.interrupt()
    vm.join(timeout=5)
    a, b = seen["RecB"], seen["RecS"]
    # every frame at most once and in order for each finder; on the stretch both were reading they saw the SAME frames.
    # The finder spawned first may read a frame or two alone before the other one is registered, and a finder leaves its
    # loop as soon as the capture position reaches bounds[1], so either of them may miss the very last frame.
    assert len(a) >= 6 and len(b) >= 6 and a == sorted(set(a)) and b == sorted(set(b))
    lo, hi = max(a[0], b[0]), min(a[-1], b[-1])
    assert [v for v in a if lo <= v <= hi] == [v for v in b if lo <= v <= hi]
    assert sum(v < lo for v in a + b) <= 3 and sum(v > hi for v in a + b) <= 1
