"""The bf16 classifier's labels against the oracle's on the shipped model, over many rendered gobans (densities, noise levels,
camera positions).  usage: python tools/fuzz_bf16.py [images = 40] [seed = 1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from camkifu_amd import capi, synth
from camkifu_amd.stone.nn_manager import NNManager
from oracle import oracle as ora

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ora.build()
ck = capi.Context(0)
W = NNManager.init_net()
ck.cnn_set_weights(W)
ck.cnn_set_mode(capi.CK_CNN_BF16)
dst = np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32)
rng = np.random.default_rng(seed)
flips = cells = low_margin = 0
worst = 0.0
for k in range(n):
    sc = synth.scene(480, 640, seed=seed * 1000 + k, density=float(rng.uniform(0.0, 0.7)), noise=float(rng.choice([0.0, 3.0, 6.0])))
    gob = ora.warp_perspective(sc["frame"].numpy(), ora.get_perspective_transform(sc["corners"], dst))
    y, labels, conf = ck.cnn_predict(gob)
    y2 = ora.cnn_predict_regions(W, gob)
    l2, c2 = ora.decode_all(y2)
    d = labels[0] != l2
    flips += int(d.sum())
    cells += d.size
    top2 = np.sort(y2, axis=1)[:, -2:]
    low_margin += int(((top2[:, 1] - top2[:, 0]) < 0.05).sum())
    worst = max(worst, float(np.abs(y[0] - y2).max()))
    if (k + 1) % 10 == 0:
        print("%d gobans: %d label differences in %d cells, worst |softmax - oracle| %.3g, regions with a top-2 margin < 0.05: %d"
              % (k + 1, flips, cells, worst, low_margin), flush=True)
sys.exit(1 if flips else 0)
