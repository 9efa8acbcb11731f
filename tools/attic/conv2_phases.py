import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from camkifu_amd import capi, synth
from camkifu_amd.stone.nn_manager import NNManager
ctx = capi.Context(0)
ctx.cnn_set_weights(NNManager.init_net())
g = torch.randint(0, 256, (64, 380, 380, 3), dtype=torch.uint8, device="cuda:0")
ctx.cnn_regions(g)
ctx.timing_enable(True); ctx.timing_reset()
for _ in range(3):
    ctx.cnn_regions(g)
ms, cnt = ctx.timing_get("cnn_conv2"); ms4, _ = ctx.timing_get("cnn_conv4")
print("conv1+2: %.2f us per frame, conv3+4: %.2f" % (1e3 * ms / (cnt * 64), 1e3 * ms4 / (cnt * 64)))
