"""cProfile of the ordered host fold over one 256-frame batch of records (bench.py's finish_host)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from camkifu_amd import capi, pipeline, synth
from camkifu_amd.controller import ControllerHeadless
from camkifu_amd.stone.nn_manager import NNManager

F, H, W = 256, 1080, 1920
dev = torch.device("cuda:0")
ctx, ctx_b = capi.Context(0), capi.Context(0)
rng = np.random.default_rng(synth.SEED)
corners = synth.random_corners(H, W, rng)
stones0 = synth.random_stones(rng, density=0.3)
st, color, positions = stones0.copy(), 1, [stones0.copy()]
for k in range((F - 1) // 5):
    while True:
        r, c = rng.integers(1, 18, 2)
        if st[r, c] == 0:
            break
    st[r, c] = color
    color = 3 - color
    positions.append(st.copy())
frames = torch.empty((F, H, W, 3), dtype=torch.uint8, device=dev)
for i in range(F):
    if i % 5 == 0:
        frames[i] = synth.render(H, W, positions[i // 5], corners, seed=synth.SEED + i, device=dev)
    else:
        frames[i] = frames[i - 1]
ctx.cnn_set_weights({k: torch.from_numpy(v).to(dev) for k, v in NNManager.init_net().items()})
M = capi.get_perspective_transform(corners, np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32))
pipe = pipeline.FastFilePipeline(H, W, ControllerHeadless(), ctx=ctx, ctx_board=ctx_b, device=dev)
board = ctx_b.board_detect(frames, -1, pipeline.LMAX, True)
labels, conf = ctx.stones_detect(frames, M)
rec = pipeline.pack_records_raw(board[0], board[1], labels.cpu().numpy(), conf.cpu().numpy())
for rep in range(2):
    pipe.stones = pipeline.StonesFold(ControllerHeadless())
    t0 = time.perf_counter()
    pipe.fold(rec)
    print("fold %d: %.2f ms" % (rep, 1e3 * (time.perf_counter() - t0)))
pipe.stones = pipeline.StonesFold(ControllerHeadless())
pr = cProfile.Profile()
pr.enable()
pipe.fold(rec)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
np.save("gpurun_out/records_256.npy", rec)
