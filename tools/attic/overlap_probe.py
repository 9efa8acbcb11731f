"""Do the VALU-bound board kernels and the MFMA-bound classifier co-execute on the same CUs?
Times board_edges (median + Canny) alone, cnn_predict alone, and both at once on two contexts
(two host threads, two HIP streams).  usage: python tools/overlap_probe.py [frames]"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from camkifu_amd import capi, synth          # noqa: E402


def main():
    F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    base = synth.scene(1080, 1920, seed=3)["frame"].to(dev)
    frames = (base[None].to(torch.int16) + torch.randint(-2, 3, (F,) + tuple(base.shape), generator=g, device=dev,
                                                         dtype=torch.int16)).clamp_(0, 255).to(torch.uint8)
    gobans = torch.randint(0, 256, (F, 380, 380, 3), generator=g, device=dev, dtype=torch.uint8)
    cb, cs = capi.Context(0), capi.Context(0)
    cs.cnn_set_weights({k: torch.from_numpy(v).to(dev) for k, v in synth.cnn_weights().items()})
    pb, ps = ThreadPoolExecutor(1), ThreadPoolExecutor(1)

    def board():
        cb.board_edges(frames)

    def stones():
        cs.cnn_predict(gobans, want_y=False)

    def timed(fns, reps=3):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            futs = [p.submit(f) for p, f in fns]
            for f in futs:
                f.result()
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3

    board(); stones()
    tb = timed([(pb, board)])
    ts = timed([(ps, stones)])
    tboth = timed([(pb, board), (ps, stones)])
    print("frames %d: board_edges alone %.2f ms, cnn alone %.2f ms, sum %.2f ms, together %.2f ms (max would be %.2f)"
          % (F, tb, ts, tb + ts, tboth, max(tb, ts)))


if __name__ == "__main__":
    main()
