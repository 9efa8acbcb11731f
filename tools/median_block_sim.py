"""What would finer threshold masks buy the median kernel?  MFMA counts per 48x48 tile on the CPU against the oracle's
medians (test infrastructure; nothing here runs in the product), by content class:
  A  the kernel as it is: a threshold is evaluated per 16-row block (16 x 48 medians) that holds its prefix: 10 MFMAs
  B  masks per 16 x 16 block: pass 1 only for the input column tiles the blocks in need read (2 .. 4), pass 2 two per block
both with the full radix descent (no scan), and as lower bounds the linear scans: span + 2 thresholds per block.
usage: python tools/median_block_sim.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as TF

from camkifu_amd import synth
from oracle import oracle as ora

ora.build()
H, W = 1080, 1920
g = torch.Generator()
g.manual_seed(7)
table = synth.natural_texture(H, W, seed=synth.SEED + 3)
classes = {
    "bench_scene": synth.film(200, H, W, seed=synth.SEED, quiet=52, move_every=32, hand_frames=12, select=[60])[0][0].numpy(),
    "natural_texture": synth.film(200, H, W, seed=synth.SEED, quiet=52, move_every=32, hand_frames=12, select=[60], background=table)[0][0].numpy(),
    "smooth_texture": (TF.interpolate(torch.rand((1, 3, H // 12 + 2, W // 12 + 2), generator=g), size=(H, W), mode="bilinear") * 255)[0].permute(1, 2, 0).to(torch.uint8).numpy(),
    "uniform_noise": torch.randint(0, 256, (H, W, 3), generator=g, dtype=torch.uint8).numpy(),
}


def prefixes(vals, b):
    return set(np.unique(vals >> (b + 1)).tolist())


for name, fr in classes.items():
    med = ora.median(np.ascontiguousarray(fr), 15).astype(np.int64)
    nty, ntx = H // 48, W // 48
    A = B = scanA = scanB = 0
    ntile = 0
    for ty in range(0, nty, 2):                       # every other tile row: enough for a mean
        for tx in range(ntx):
            for c in range(3):
                t = med[48 * ty:48 * ty + 48, 48 * tx:48 * tx + 48, c]
                ntile += 1
                for rb in range(3):
                    row = t[16 * rb:16 * rb + 16]
                    blocks = [row[:, 16 * u:16 * u + 16] for u in range(3)]
                    scanA += 10 * (int(row.max() - row.min()) + 2)
                    # a linear scan with per-block masks: threshold v is needed by block u iff min_u - 1 <= v <= max_u
                    lo, hi = int(row.min()) - 1, int(row.max())
                    for v in range(lo, hi + 1):
                        S = [u for u in range(3) if blocks[u].min() - 1 <= v <= blocks[u].max()]
                        scanB += len({x for u in S for x in (u, u + 1)}) + 2 * len(S)
                    for b in range(8):
                        pa = prefixes(row, b)
                        A += 10 * len(pa)
                        pu = [prefixes(blk, b) for blk in blocks]
                        for q in pa:
                            S = [u for u in range(3) if q in pu[u]]
                            B += len({x for u in S for x in (u, u + 1)}) + 2 * len(S)
    print("%-16s MFMAs per tile:  A radix %7.1f   B radix %7.1f (%+.0f%%)   A scan bound %7.1f   B scan bound %7.1f (%+.0f%% of A radix)"
          % (name, A / ntile, B / ntile, 100.0 * (B / A - 1), scanA / ntile, scanB / ntile, 100.0 * (scanB / A - 1)), flush=True)
