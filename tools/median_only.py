"""K1 alone on a batch of synthetic 1080p frames (for rocprofv3 --pmc runs and quick timing).
usage: python tools/median_only.py [frames] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from camkifu_amd import capi, synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 32
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H, W = 1080, 1920
dev = torch.device("cuda:0")
rng = np.random.default_rng(synth.SEED)
corners = synth.random_corners(H, W, rng)
frames = torch.empty((F, H, W, 3), dtype=torch.uint8, device=dev)
for i in range(F):
    frames[i] = synth.render(H, W, synth.random_stones(np.random.default_rng(i), 0.3), corners, seed=i, device=dev)
torch.cuda.synchronize()
ctx = capi.Context(0)
ctx.median15(frames)
ctx.timing_enable(True)
ctx.timing_reset()
for _ in range(R):
    ctx.median15(frames)
ms, cnt = ctx.timing_get("median")
print("median15: %.2f us per 1080p frame (%d frames x %d calls)" % (1e3 * ms / (cnt * F), F, cnt))
ctx.close()
