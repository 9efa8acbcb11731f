"""The board path alone at 3840 x 2160 (BASELINE config 4's frame size): stage times per frame, HIP events on the context's stream.
usage: python tools/board_only_4k.py [frames = 64] [reps = 3]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from camkifu_amd import capi, pipeline, synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H, W = 2160, 3840
frames = synth.film(F, H, W, seed=synth.SEED, device=torch.device("cuda:0"), quiet=52, move_every=32, hand_frames=12)[0]
torch.cuda.synchronize()
ctx = capi.Context(0)
ctx.board_detect(frames, -1, pipeline.LMAX, True)
ctx.timing_enable(True)
ctx.timing_reset()
for _ in range(R):
    ctx.board_detect(frames, -1, pipeline.LMAX, True)
out = []
for name in ("median", "canny_nms", "canny_hyst", "ccl", "contour_gather", "ghost", "hough_vote"):
    ms, cnt = ctx.timing_get(name)
    if cnt:
        out.append("%s %.2f" % (name, 1e3 * ms / (R * F)))
print("4K board path, us per frame at %d frames per call: %s" % (F, "  ".join(out)))
ctx.close()
