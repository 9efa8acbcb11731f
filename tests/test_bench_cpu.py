"""bench.py's CPU-side helpers: the cpu_baseline leg must time the SAME arithmetic the oracle defines (BASELINE.md 3:
filters by the C restatement with OpenMP across frames, classifier by torch on the CPU), and count the threads it
really has."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_torch_cpu_classifier_equals_the_oracle_network(ora):
    import bench
    from camkifu_amd import synth
    from camkifu_amd.stone.nn_manager import NNManager
    W = NNManager.init_net()
    sc = synth.scene(480, 640, seed=5, density=0.3)
    dst = np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32)
    goban = ora.warp_perspective(sc["frame"].numpy(), ora.get_perspective_transform(sc["corners"], dst))
    p = bench.patches_of(goban[None])
    assert tuple(p.shape) == (100, 3, 40, 40)
    # patch (i, j) starts at rows / columns 0, 40, ..., 320, 340 (nn_manager.py:92-126)
    assert np.array_equal(p[99].permute(1, 2, 0).numpy().astype(np.uint8), goban[340:380, 340:380])
    assert np.array_equal(p[10].permute(1, 2, 0).numpy().astype(np.uint8), goban[40:80, 0:40])
    y = bench.torch_cpu_net(W)(p).numpy()
    y_ref = ora.cnn_predict_regions(W, goban)
    assert np.abs(y - y_ref).max() <= 1e-4 and np.array_equal(y.argmax(1), y_ref.argmax(1))


def test_across_frames_baseline_equals_the_per_call_oracle(ora):
    from camkifu_amd import synth
    from camkifu_amd.stone.nn_manager import NNManager
    W = NNManager.init_net()
    frames, corners, *_ = synth.film(3, 240, 320, seed=9, quiet=1, move_every=2, hand_frames=1)
    frames = frames.numpy()
    M = ora.get_perspective_transform(corners, np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32))
    nl, labels, used = ora.baseline_frames(W, frames, M, 3)
    nl2, gobans, _ = ora.baseline_frames(None, frames, M, 2)
    assert used >= 1 and np.array_equal(nl, nl2)
    for f in range(3):
        ref = ora.board_lines(ora.canny(ora.median(frames[f], 15), 25, 75))
        g = ora.warp_perspective(frames[f], M)
        assert nl[f] == ref["status"] and np.array_equal(gobans[f], g)
        assert np.array_equal(labels[f], ora.decode_all(ora.cnn_predict_regions(W, g))[0])


def test_host_threads_is_what_the_process_may_use():
    import bench
    n = bench.host_threads()
    assert 1 <= n <= (os.cpu_count() or 1)
    if hasattr(os, "sched_getaffinity"):
        assert n <= len(os.sched_getaffinity(0))
