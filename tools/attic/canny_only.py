"""K2 alone on the median image of the bench's film, for per-phase timing of the NMS kernel (tools/nms_phase_times.sh).
usage: python tools/canny_only.py [frames] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from camkifu_amd import capi, synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H, W = 1080, 1920
dev = torch.device("cuda:0")
frames = synth.film(F, H, W, seed=synth.SEED, device=dev, quiet=52, move_every=32, hand_frames=12)[0]
torch.cuda.synchronize()
ctx = capi.Context(0)
med = ctx.median15(frames)
for _ in range(R):
    ctx.canny(med)
ctx.close()
print("done", F, R)
