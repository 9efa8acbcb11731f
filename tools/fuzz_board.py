"""Randomised parity sweep of the board path against the oracle (sizes, textures, thresholds); not part of the test
suite because of its run time.  usage: python tools/fuzz_board.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import ndimage

from camkifu_amd import capi
from oracle import oracle as ora

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ora.build()
ck = capi.Context(0)
bad = 0
for case in range(ncases):
    h, w = int(rng.integers(3, 260)), int(rng.integers(3, 330))
    n = int(rng.choice([1, 2, 3, 8]))
    kind = case % 4
    frames = []
    for _ in range(n):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if kind == 1:
            img = ndimage.uniform_filter(img.astype(np.float32), (9, 9, 1)).astype(np.uint8)
        elif kind == 2:                                   # blocks and strokes: long straight edges
            img[:] = rng.integers(60, 120, 3, dtype=np.uint8)
            for _ in range(int(rng.integers(1, 8))):
                y0, x0 = int(rng.integers(0, h)), int(rng.integers(0, w))
                y1, x1 = int(rng.integers(y0, h + 1)), int(rng.integers(x0, w + 1))
                img[y0:y1, x0:x1] = rng.integers(0, 256, 3, dtype=np.uint8)
        elif kind == 3:
            img = (img // 64) * 64
        frames.append(img)
    frames = np.stack(frames)
    med = ck.median15(frames)
    edges = ck.board_edges(frames)
    out = ck.board_detect(frames)
    for k in range(n):
        m = ora.median(frames[k], 15)
        e = ora.canny(m, 25, 75)
        o = ora.board_lines(e)
        parts = (np.array_equal(med[k], m), np.array_equal(edges[k], e), out[k]["n_lines"] == max(o["status"], 0),   # negative oracle status = no contour / gate not passed: no lines
                 np.array_equal(out[k]["lines"], o["lines"][:len(out[k]["lines"])]))      # the C-ABI caps the line list
        if not all(parts):
            bad += 1
            if bad <= 12:
                print("MISMATCH case %d frame %d/%d size %dx%d kind %d: median %s edges %s status %s (%s vs %s) lines %s"
                      % (case, k, n, h, w, kind, parts[0], parts[1], parts[2], out[k]["n_lines"], o["status"], parts[3]))
print("%d cases, %d mismatches" % (ncases, bad))
sys.exit(1 if bad else 0)
