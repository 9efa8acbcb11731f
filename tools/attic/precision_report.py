"""How accurate are the classifier modes?  Softmax outputs of the HIP library in fp32 / f16x2 / bf16 mode against a
float64 evaluation of the same network (torch CPU, true convolution = flipped kernels), on rendered boards."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F

from camkifu_amd import capi, synth
from camkifu_amd.stone.nn_manager import NNManager
from oracle import oracle as ora


def ref64(W, goban):
    origins = [0, 40, 80, 120, 160, 200, 240, 280, 320, 340]
    x = np.stack([goban[r:r + 40, c:c + 40] for r in origins for c in origins]).astype(np.float64)
    x = torch.from_numpy(x).permute(0, 3, 1, 2)

    def conv(x, w, b):      # Keras-1/Theano true convolution: flip the kernel
        wt = torch.from_numpy(w.astype(np.float64)).flip(0, 1).permute(3, 2, 0, 1)
        return F.relu(F.conv2d(x, wt, torch.from_numpy(b.astype(np.float64))))
    x = conv(x, W["c1w"], W["c1b"]); x = conv(x, W["c2w"], W["c2b"]); x = F.max_pool2d(x, 2)
    x = conv(x, W["c3w"], W["c3b"]); x = conv(x, W["c4w"], W["c4b"]); x = F.max_pool2d(x, 2)
    x = x.permute(0, 2, 3, 1).reshape(100, -1)
    x = F.relu(x @ torch.from_numpy(W["d1w"].astype(np.float64)) + torch.from_numpy(W["d1b"].astype(np.float64)))
    x = x @ torch.from_numpy(W["d2w"].astype(np.float64)) + torch.from_numpy(W["d2b"].astype(np.float64))
    return torch.softmax(x, 1).numpy()


def main():
    ctx = capi.Context(0)
    dst = np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32)
    for name, W in (("trained (camkifu_amd/data/keras.h5)", NNManager.init_net()), ("seeded random", synth.cnn_weights())):
        ctx.cnn_set_weights(W)
        errs = {m: 0.0 for m in ("fp32", "f16x2", "bf16", "oracle")}
        flips = {m: 0 for m in errs}
        for seed in range(4):
            sc = synth.scene(480, 640, seed=300 + seed, density=0.15 * (seed + 1))
            gob = ora.warp_perspective(sc["frame"].numpy(), ora.get_perspective_transform(sc["corners"], dst))
            truth = ref64(W, gob)
            for m, mode in (("fp32", capi.CK_CNN_FP32), ("f16x2", capi.CK_CNN_F16X2), ("bf16", capi.CK_CNN_BF16)):
                ctx.cnn_set_mode(mode)
                y = ctx.cnn_predict(gob)[0][0]
                errs[m] = max(errs[m], float(np.abs(y - truth).max()))
                flips[m] += int((y.argmax(1) != truth.argmax(1)).sum())
            y = ora.cnn_predict_regions(W, gob)
            errs["oracle"] = max(errs["oracle"], float(np.abs(y - truth).max()))
            flips["oracle"] += int((y.argmax(1) != truth.argmax(1)).sum())
        print("%-34s max |softmax - float64|: " % name + ", ".join("%s %.2e (%d label flips)" % (m, errs[m], flips[m]) for m in errs))


if __name__ == "__main__":
    main()
