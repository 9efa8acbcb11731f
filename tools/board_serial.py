"""Both paths alone, one context each, strictly serial, on the first frames of the bench's film: for
`rocprofv3 --kernel-trace --stats` (per-kernel times without any overlap).
usage: python tools/board_serial.py [frames] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from camkifu_amd import capi, pipeline, synth
from camkifu_amd.stone.nn_manager import NNManager

F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
R = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H, W = 1080, 1920
dev = torch.device("cuda:0")
frames, corners, truth, moves, hands = synth.film(F, H, W, seed=synth.SEED, device=dev, quiet=52, move_every=32, hand_frames=12)
torch.cuda.synchronize()
ctx = capi.Context(0)
ctx.cnn_set_weights(NNManager.init_net())
M = capi.get_perspective_transform(corners, np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32))
handle = ctx.mog2_create(380, 380)
for _ in range(R):
    ctx.board_detect(frames, -1, pipeline.LMAX, True)
    ctx.stones_run(frames, M, mog2=handle, learning_rates=np.full(F, 0.005))
ctx.close()
print("done", F, R)
