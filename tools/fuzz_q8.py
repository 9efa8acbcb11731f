"""CK_CNN_F16Q8 against the oracle over many rendered gobans (densities, noise levels, camera positions) and noise images, on
the shipped model and on random weights: pooled maps relative to their scale (bar 1e-4), softmax, labels.
usage: python tools/fuzz_q8.py [images = 40] [seed = 1]"""
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
dst = np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32)
bad = 0
for name, W in (("trained", NNManager.init_net()), ("random", synth.cnn_weights())):
    ck.cnn_set_weights(W)
    ck.cnn_set_mode(capi.CK_CNN_F16Q8)
    rng = np.random.default_rng(seed)
    flips = cells = 0
    w2 = w4 = wy = 0.0
    for k in range(n):
        if k % 8 == 7:
            gob = rng.integers(0, 256, (380, 380, 3), dtype=np.uint8)
        else:
            sc = synth.scene(480, 640, seed=seed * 1000 + k, density=float(rng.uniform(0.0, 0.7)), noise=float(rng.choice([0.0, 3.0, 6.0])))
            gob = ora.warp_perspective(sc["frame"].numpy(), ora.get_perspective_transform(sc["corners"], dst))
        p2, p4 = ck.cnn_maps(gob)
        y, labels, conf = ck.cnn_predict(gob)
        y2, o2, o4 = ora.cnn_region_maps(W, gob)
        l2, _ = ora.decode_all(y2)
        top2 = np.sort(y2, axis=1)[:, -2:]
        clear = np.repeat((top2[:, 1] - top2[:, 0]) > 1e-3, 1)
        rl, _ = ck.cnn_regions(gob)
        d = (np.asarray(rl).reshape(-1) != y2.argmax(1)) & clear
        flips += int(d.sum())
        cells += int(clear.sum())
        w2 = max(w2, float(np.abs(p2[0].reshape(o2.shape) - o2).max() / np.abs(o2).max()))
        w4 = max(w4, float(np.abs(p4[0].reshape(o4.shape) - o4).max() / np.abs(o4).max()))
        wy = max(wy, float(np.abs(y[0] - y2).max()))
        if (k + 1) % 10 == 0 or k + 1 == n:
            print("%s weights, %d images: worst pool2 %.3g  pool4 %.3g of scale, softmax %.3g; region labels differing where the oracle's margin exceeds 1e-3: %d of %d"
                  % (name, k + 1, w2, w4, wy, flips, cells), flush=True)
    bad += flips + (w2 > 1e-4) + (w4 > 1e-4)
sys.exit(1 if bad else 0)
