"""Throughput of the board path alone (K1..K6 + host pruning) with 1..3 lanes of contexts: how much of the
GPU-serial time do the host round trips of k_board_lines cost?"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from camkifu_amd import capi, pipeline, synth

F, H, W = 256, 1080, 1920
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
for lanes in (1, 2, 3, 4):
    ctxs = [capi.Context(0) for _ in range(lanes)]
    pools = [ThreadPoolExecutor(1) for _ in range(lanes)]
    cuts = [round(i * F / lanes) for i in range(lanes + 1)]
    sl = [frames[cuts[i]:cuts[i + 1]] for i in range(lanes)]

    def step():
        return [p.submit(c.board_detect, s, -1, pipeline.LMAX, True) for p, c, s in zip(pools, ctxs, sl)]
    for f in step():
        f.result()
    t0 = time.perf_counter()
    K = 6
    infl = [step(), step()]
    for i in range(K):
        for f in infl.pop(0):
            f.result()
        if i + 2 < K:
            infl.append(step())
    dt = time.perf_counter() - t0
    ctxs[0].timing_enable(True); ctxs[0].timing_reset()
    ctxs[0].board_detect(frames, -1, pipeline.LMAX, True)
    tot = sum(ctxs[0].timing_get(n)[0] for n in ("median", "canny_nms", "canny_hyst", "ccl", "contour_gather", "ghost", "hough_vote", "hough_peaks"))
    print("lanes %d: %.2f ms per 256 frames (GPU kernels serial: %.2f ms) -> %.0f fps" % (lanes, 1e3 * dt / K, tot, K * F / dt))
    for c in ctxs:
        c.close()
