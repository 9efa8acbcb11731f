"""How many (threshold, row block) box counts does the median kernel need on a bench frame, by threshold strategy?
Simulated on the CPU against the oracle's medians (test infrastructure; nothing here runs in the product): the radix
descent of round 3 (distinct prefixes per level and 16-row block) against round 4's classification by the range of 256
sampled input pixels + linear scan around their mean, for several range limits and scan caps.
usage: python tools/median_threshold_sim.py      (numbers quoted in DESIGN.md 4 / profiles/r04_median_scan.txt)"""
import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from camkifu_amd import synth
from oracle import oracle as ora
ora.build()
H, W = 1080, 1920
for sel in ([60], [150]):
    frames = synth.film(200, H, W, seed=synth.SEED, device="cpu", quiet=52, move_every=32, hand_frames=12, select=sel)[0]
    fr = frames[0].numpy()
    med = ora.median(fr, 15)
    nty, ntx = H // 48, W // 48
    def radix_cost(vals, bits=8):
        return sum(len(np.unique(vals >> (b + 1))) for b in range(bits))
    pad = np.pad(fr, ((7, 64), (7, 64), (0, 0)), mode="edge")
    rows = []
    for ty in range(nty):
        for tx in range(ntx):
            for c in range(3):
                t = med[48 * ty:48 * ty + 48, 48 * tx:48 * tx + 48, c].astype(np.int64).reshape(3, 16 * 48)
                win = pad[48 * ty:48 * ty + 64, 48 * tx:48 * tx + 64, c].astype(np.int64)      # the 64x64 input region (origin -7)
                # samples: lane (n, g): rows 16g + {1, 6, 9, 14}, cols 16 i + n for i = 0..3
                smp = np.stack([win[np.arange(4)[:, None] * 16 + r, np.arange(16)[None, :] + 16 * i] for i, r in enumerate((1, 6, 9, 14))]).ravel()
                rows.append((t, smp))
    now = np.array([sum(radix_cost(rb) for rb in t) for t, _ in rows])
    for R0 in (16, 20, 24, 28, 32, 40):
        for CAP in (8, 12):
            tot = []; nscan = nfall = 0
            for (t, smp), base in zip(rows, now):
                rng = smp.max() - smp.min()
                if rng > R0:
                    tot.append(base); continue
                g = int(round(smp.mean()))
                nscan += 1
                c = 0; fail = False
                for rb in t:
                    up = max(rb.max() - g, 0) + 1
                    dn = (g - rb.min() + 1) if rb.min() <= g else 0
                    if up > CAP or dn > CAP: fail = True
                    c += min(up, CAP) + min(dn, CAP)
                if fail:
                    nfall += 1; c += base
                tot.append(c)
            tot = np.array(tot)
            print(sel, "range<=%d cap %d" % (R0, CAP), "scan frac %.3f fallback %.4f" % (nscan / len(rows), nfall / len(rows)), "units/tile %.2f (now %.2f) -> %.1f%%" % (tot.mean(), now.mean(), 100 * (tot.mean() / now.mean() - 1)))
