"""Board path alone, one context, serial: for `rocprofv3 --kernel-trace --stats` (per-kernel times without overlap).
usage: python tools/board_serial.py [frames] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from camkifu_amd import capi, pipeline, synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H, W = 1080, 1920
dev = torch.device("cuda:0")
rng = np.random.default_rng(synth.SEED)
corners = synth.random_corners(H, W, rng)
frames = torch.empty((F, H, W, 3), dtype=torch.uint8, device=dev)
for i in range(F):
    if i % 8 == 0:
        frames[i] = synth.render(H, W, synth.random_stones(np.random.default_rng(i), 0.3), corners, seed=i, device=dev)
    else:
        frames[i] = frames[i - 1]
torch.cuda.synchronize()
ctx = capi.Context(0)
for _ in range(R):
    ctx.board_detect(frames, -1, pipeline.LMAX, True)
ctx.close()
print("done", F, R)
