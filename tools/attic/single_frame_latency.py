"""BASELINE config 2: ONE 1920x1080 frame through the board path and the stones path, as a per-frame finder calls them
(one frame per call, results back on the host): milliseconds per call, frame in host memory (what a capture thread
hands over; pageable numpy) and frame already in HBM.  CK_PROFILE_HOST=1 adds the board call's host laps.
usage: python tools/single_frame_latency.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from camkifu_amd import capi, pipeline, synth
from camkifu_amd.stone.nn_manager import NNManager

dev = torch.device("cuda:0")
frames, corners = synth.film(64, 1080, 1920, seed=synth.SEED, device=dev, quiet=52, move_every=32, hand_frames=12, select=[60, 61, 62, 63])[:2]
torch.cuda.synchronize()
M = capi.get_perspective_transform(corners, np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32))
ctx_b, ctx_s = capi.Context(0), capi.Context(0)
ctx_s.cnn_set_weights(NNManager.init_net())
handle = ctx_s.mog2_create(380, 380)
host = [f.cpu().numpy() for f in frames]


def timeit(fn, reps=30):
    for _ in range(3):
        fn(0)
    ts = []
    for k in range(reps):
        t0 = time.perf_counter()
        fn(k)
        ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]


for where, src in (("host", host), ("HBM", [f[None] for f in frames])):
    b = timeit(lambda k: ctx_b.board_detect(src[k % 4] if where == "HBM" else src[k % 4][None], -1, pipeline.LMAX, True))
    s = timeit(lambda k: ctx_s.stones_run(src[k % 4] if where == "HBM" else src[k % 4][None], M, mog2=handle, learning_rates=[0.005],
                                          want_grid=True))
    print("frame in %-4s  board_detect %.3f ms (min %.3f max %.3f)   stones_run %.3f ms (min %.3f max %.3f)" % ((where,) + b + s))
