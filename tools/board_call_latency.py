import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from camkifu_amd import capi, synth, pipeline
dev = torch.device("cuda:0")
frames = synth.film(64, 1080, 1920, seed=synth.SEED, device=dev, quiet=52, move_every=32, hand_frames=12)[0]
torch.cuda.synchronize()
ctx = capi.Context(0, priority=1)
for n in (1, 4, 8, 16):
    ctx.board_detect(frames[:n], -1, pipeline.LMAX, True)
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.board_detect(frames[:n], -1, pipeline.LMAX, True)
    print("board_detect on %2d frames: %.3f ms per call" % (n, 1e2 * (time.perf_counter() - t0)))
os.environ["CK_PROFILE_HOST"] = "1"
