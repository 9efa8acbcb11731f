#!/usr/bin/env python3
"""PCIe-inclusive rate: frames start in (pageable) host memory, results end on the host.
Never the headline `value` (bench.py keeps inputs resident in HBM); noted in DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from camkifu_amd import capi, synth
from camkifu_amd.stone.nn_manager import NNManager
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ctx = capi.Context(0)
ctx.cnn_set_weights(NNManager.init_net())
sc = synth.scene(1080, 1920, seed=1)
frames = np.ascontiguousarray(np.broadcast_to(sc["frame"].numpy(), (n, 1080, 1920, 3)))
M = capi.get_perspective_transform(sc["corners"], np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32))
for _ in range(2):
    ctx.board_detect(frames, raw=True); ctx.stones_detect(frames, M)
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    ctx.board_detect(frames, raw=True); ctx.stones_detect(frames, M)
dt = time.perf_counter() - t0
print("PCIe-inclusive (pageable host frames in, host results out, serial, frames uploaded twice): %.1f frames/s"
      % (reps * n / dt))
