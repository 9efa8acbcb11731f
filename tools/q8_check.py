"""CK_CNN_F16Q8 beside CK_CNN_F16X2 against a float64 torch evaluation (trained and random weights): pooled maps (error
relative to the map's scale), softmax, labels.  Times per mode: tools/cnn_modes.py.  GPU box:  python tools/q8_check.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from camkifu_amd import capi, synth
    from camkifu_amd.stone.nn_manager import NNManager
    from oracle import oracle as ora
    from test_gpu_parity import _torch_fp64_maps, _torch_fp64_classifier
    ora.build()
    ck = capi.Context()
    dst = np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32)
    gobans = []
    for seed, dens in ((31, 0.45), (71, 0.15), (72, 0.3)):
        sc = synth.scene(480, 640, seed=seed, density=dens)
        gobans.append(ora.warp_perspective(sc["frame"].numpy(), ora.get_perspective_transform(sc["corners"], dst)))
    gobans = np.stack(gobans)
    for name, W in (("trained", NNManager.init_net()), ("random", synth.cnn_weights())):
        ck.cnn_set_weights(W)
        r2, r4 = _torch_fp64_maps(W, gobans)
        y64 = _torch_fp64_classifier(W, gobans)
        for mode, mname in ((capi.CK_CNN_F16X2, "f16x2"), (capi.CK_CNN_F16Q8, "f16q8")):
            ck.cnn_set_mode(mode)
            p2, p4 = ck.cnn_maps(gobans)
            y, labels, conf = ck.cnn_predict(gobans)
            rl, rc = ck.cnn_regions(gobans)
            e2 = np.abs(p2.reshape(r2.shape) - r2).max() / np.abs(r2).max()
            e4 = np.abs(p4.reshape(r4.shape) - r4).max() / np.abs(r4).max()
            top2 = np.sort(y64, axis=2)[..., -2:]
            clear = (top2[..., 1] - top2[..., 0]) > 1e-3
            flips = int((rl.reshape(len(gobans), 100)[clear] != y64.argmax(2)[clear]).sum())
            print("%s weights, %s: pool2 %.3g  pool4 %.3g of scale; softmax %.3g; label flips where float64 is clear: %d"
                  % (name, mname, e2, e4, np.abs(y - y64).max(), flips), flush=True)


if __name__ == "__main__":
    main()
