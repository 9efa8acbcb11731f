"""Random weight sets and gobans through the default (split-precision, fused) classifier against the oracle.
usage: python tools/fuzz_cnn.py [weight sets] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from camkifu_amd import capi, synth
from oracle import oracle as ora

nsets = int(sys.argv[1]) if len(sys.argv) > 1 else 4
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ora.build()
ck = capi.Context(0)
rng = np.random.default_rng(seed)
worst = 0.0
flips = 0
for s in range(nsets):
    W = synth.cnn_weights(seed=seed * 1000 + s)
    scale = float(rng.choice([0.5, 1.0, 2.0]))
    for k in W:
        if not k.endswith("b"):
            W[k] = (W[k] * np.float32(scale ** 0.25)).astype(np.float32)
    ck.cnn_set_weights(W)
    gob = rng.integers(0, 256, (3, 380, 380, 3), dtype=np.uint8)
    gob[1] = (gob[1] // 32) * 32
    gob[2, 100:300, 50:350] = rng.integers(0, 256, 3, dtype=np.uint8)
    y, labels, conf = ck.cnn_predict(gob)
    for k in range(len(gob)):
        y2 = ora.cnn_predict_regions(W, gob[k])
        d = float(np.abs(y[k] - y2).max())
        worst = max(worst, d)
        l2, c2 = ora.decode_all(y2)
        flips += int((labels[k] != l2).sum())
    print("set %d (scale %.2f): worst |softmax - oracle| so far %.3g, label differences %d" % (s, scale, worst, flips))
print("worst %.3g, label differences %d" % (worst, flips))
sys.exit(1 if worst > 1e-4 else 0)
