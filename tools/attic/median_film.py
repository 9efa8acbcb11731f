"""K1 alone on frames of the bench's film (timing of kernel variants): python tools/median_film.py [frames] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from camkifu_amd import capi, synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
frames = synth.film(256, 1080, 1920, seed=synth.SEED, device=dev, quiet=52, move_every=32, hand_frames=12, select=list(range(0, 256, 256 // F)))[0]
torch.cuda.synchronize()
ctx = capi.Context(0)
ctx.median15(frames)
ctx.timing_enable(True)
ctx.timing_reset()
for _ in range(R):
    ctx.median15(frames)
ms, cnt = ctx.timing_get("median")
print("%s median15: %.2f us per 1080p frame (%d frames x %d calls)" % (os.environ.get("CK_HIP_LIB", "default"), 1e3 * ms / (cnt * len(frames)), len(frames), cnt))
ctx.close()
