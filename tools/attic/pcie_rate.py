#!/usr/bin/env python3
"""PCIe-inclusive rates: frames start in host memory, results end on the host.  Never the headline
`value` (bench.py keeps inputs resident in HBM); noted in DESIGN.md.
  a) BGR frames in pageable host memory, each finder call uploads them itself (what a drop-in finder
     fed numpy frames does): 2 x 3 B/px over PCIe
  b) BGR frames in pinned memory, uploaded once per batch, both paths read the HBM copy: 3 B/px
  c) I420 frames (as a .y4m reader holds them) in pinned memory, uploaded once, converted to BGR on the
     GPU (ck_i420_to_bgr): 1.5 B/px
Also prints the achieved HBM rate of the conversion kernel (HBM-bound: 1.5 B/px in, 3 B/px out)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from camkifu_amd import capi, synth
from camkifu_amd.stone.nn_manager import NNManager

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, W = 1080, 1920
dev = torch.device("cuda:0")
ctx = capi.Context(0)
ctx.cnn_set_weights(NNManager.init_net())
sc = synth.scene(H, W, seed=1)
frames = np.ascontiguousarray(np.broadcast_to(sc["frame"].numpy(), (n, H, W, 3)))
M = capi.get_perspective_transform(sc["corners"], np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32))
pin_bgr = torch.from_numpy(frames).pin_memory()
pin_yuv = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(synth.bgr_to_i420(sc["frame"].numpy()), (n, H * W * 3 // 2)))).pin_memory()


def run_a():
    ctx.board_detect(frames, raw=True)
    ctx.stones_detect(frames, M)


def run_b():
    d = pin_bgr.to(dev, non_blocking=True)
    torch.cuda.synchronize()
    ctx.board_detect(d, raw=True)
    lab, conf = ctx.stones_detect(d, M)
    lab.cpu(); conf.cpu()


def run_c():
    d = ctx.i420_to_bgr(pin_yuv.numpy(), H, W, to_device=dev)
    ctx.board_detect(d, raw=True)
    lab, conf = ctx.stones_detect(d, M)
    lab.cpu(); conf.cpu()


for name, fn in (("a) pageable BGR, uploaded by each call", run_a), ("b) pinned BGR, one upload", run_b),
                 ("c) pinned I420, one upload + GPU conversion", run_c)):
    for _ in range(2):
        fn()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("PCIe-inclusive %-48s %8.1f frames/s (serial: copy, then board path, then stones path)" % (name, reps * n / dt))

yuv_dev = pin_yuv.to(dev)
ctx.timing_enable(True)
ctx.timing_reset()
for _ in range(5):
    ctx.i420_to_bgr(yuv_dev, H, W)
ms, cnt = ctx.timing_get("i420_to_bgr")
per = ms / cnt * 1e-3
print("i420_to_bgr kernel: %.1f us per %d-frame launch, %.0f GB/s algorithmic (4.5 B/px) = %.1f %% of 8 TB/s"
      % (per * 1e6, n, n * H * W * 4.5 / per / 1e9, n * H * W * 4.5 / per / 8e12 * 100))
