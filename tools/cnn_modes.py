"""The classifier alone, per mode: stage times (HIP events on the context's stream) and, for a few gobans, the pooled maps and
labels against the oracle.  usage: python tools/cnn_modes.py [frames per call = 128] [modes = bf16,f16x2] [--check]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from camkifu_amd import capi, synth
from camkifu_amd.stone.nn_manager import NNManager

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 128
modes = sys.argv[2].split(",") if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ["bf16", "f16x2"]
MODE = {"f16x2": capi.CK_CNN_F16X2, "fp32": capi.CK_CNN_FP32, "bf16": capi.CK_CNN_BF16, "f16q8": capi.CK_CNN_F16Q8}
ctx = capi.Context(0)
W = NNManager.init_net()
ctx.cnn_set_weights(W)

if "--check" in sys.argv:
    from oracle import oracle as ora
    ora.build()
    dst = np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32)
    gobs = []
    for k in range(3):
        sc = synth.scene(480, 640, seed=31 + k, density=0.2 * k)
        gobs.append(ora.warp_perspective(sc["frame"].numpy(), ora.get_perspective_transform(sc["corners"], dst)))
    gobs = np.stack(gobs)
    for m in modes:
        ctx.cnn_set_mode(MODE[m])
        p2, p4 = ctx.cnn_maps(gobs)
        y, labels, conf = ctx.cnn_predict(gobs)
        for k in range(len(gobs)):
            y2, o2, o4 = ora.cnn_region_maps(W, gobs[k])
            l2, _ = ora.decode_all(y2)
            e2 = float(np.abs(p2[k].reshape(o2.shape) - o2).max() / np.abs(o2).max())
            e4 = float(np.abs(p4[k].reshape(o4.shape) - o4).max() / np.abs(o4).max())
            print("%-6s goban %d: pool2 err %.3g of scale, pool4 err %.3g, softmax err %.3g, label differences %d"
                  % (m, k, e2, e4, float(np.abs(y[k] - y2).max()), int((labels[k] != l2).sum())), flush=True)

g = torch.randint(0, 256, (n, 380, 380, 3), dtype=torch.uint8, device="cuda:0")
for m in modes:
    ctx.cnn_set_mode(MODE[m])
    ctx.cnn_regions(g)
    ctx.timing_enable(True)
    ctx.timing_reset()
    reps = 5
    for _ in range(reps):
        ctx.cnn_regions(g)
    out = []
    tot = 0.0
    for name in ("cnn_conv1", "cnn_conv2", "cnn_conv3", "cnn_conv4", "cnn_tail"):
        ms, cnt = ctx.timing_get(name)
        if cnt:
            out.append("%s %.2f" % (name[4:], 1e3 * ms / (reps * n)))
            tot += 1e3 * ms / (reps * n)
    ctx.timing_enable(False)
    print("%-6s us per frame at %d frames per call: %s | classifier %.2f" % (m, n, "  ".join(out), tot), flush=True)
