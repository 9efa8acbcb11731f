"""K1 (exact 15x15 median) across content classes: us per 1080p frame, 8 frames per launch, and -- for every class -- the
result compared pixel for pixel with a sort-based median (torch.median over unfolded windows) on two 240 x 320 crops.
usage: python tools/k1_content.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as TF

from camkifu_amd import capi, synth

R = int(sys.argv[1]) if len(sys.argv) > 1 else 5
H, W = 1080, 1920
dev = torch.device("cuda:0")


def reference_median(img):
    """(h, w, 3) uint8 on the GPU -> exact 15x15 median with replicate border, by sorting every window"""
    x = img.permute(2, 0, 1)[None].float()
    x = TF.pad(x, (7, 7, 7, 7), mode="replicate")
    win = x.unfold(2, 15, 1).unfold(3, 15, 1).reshape(1, 3, img.shape[0], img.shape[1], 225)
    return win.median(dim=-1).values[0].permute(1, 2, 0).to(torch.uint8)


g = torch.Generator(device=dev)
g.manual_seed(7)
frames = synth.film(8, H, W, seed=synth.SEED, device=dev, quiet=52, move_every=32, hand_frames=12)[0]
table = synth.natural_texture(H, W, seed=synth.SEED + 3, device=dev)
nat = synth.film(8, H, W, seed=synth.SEED, device=dev, quiet=52, move_every=32, hand_frames=12, background=table)[0]
noise = torch.randint(0, 256, (8, H, W, 3), generator=g, device=dev, dtype=torch.uint8)
coarse = torch.rand((8, 3, H // 12 + 2, W // 12 + 2), generator=g, device=dev)
tex = (TF.interpolate(coarse, size=(H, W), mode="bilinear") * 255).permute(0, 2, 3, 1).contiguous().to(torch.uint8)
# narrow-band noise (medians in a dozen levels, input range ~60) and a two-level checker (two far clusters)
band = (128 + 10 * torch.randn((8, H, W, 3), generator=g, device=dev)).clamp(0, 255).to(torch.uint8)
ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
checker = torch.where(((ys // 40 + xs // 40) % 2 == 0)[None, ..., None], 60, 200).expand(8, H, W, 3).contiguous().to(torch.uint8)
checker = (checker.float() + 3 * torch.randn((8, H, W, 3), generator=g, device=dev)).clamp(0, 255).to(torch.uint8)
ctx = capi.Context(0)
bad = 0
for name, batch in (("bench_scene", frames), ("natural_texture", nat), ("smooth_texture", tex), ("uniform_noise", noise),
                    ("band_noise", band), ("checker", checker)):
    ctx.median15(batch)
    ctx.timing_enable(True)
    ctx.timing_reset()
    for _ in range(R):
        med = ctx.median15(batch)
    ms, cnt = ctx.timing_get("median")
    ctx.timing_enable(False)
    wrong = 0
    for (y0, x0) in ((0, 0), (H - 240, W - 320), (400, 800)):
        # the reference median of a 240 x 320 window is computed on the window grown by the filter's reach (7 px, clipped at
        # the frame): inside the window the replicate border the reference pads with is then the frame's own
        ya, xa = max(0, y0 - 7), max(0, x0 - 7)
        grown = reference_median(batch[0, ya:min(H, y0 + 247), xa:min(W, x0 + 327)])
        want = grown[y0 - ya:y0 - ya + 240, x0 - xa:x0 - xa + 320]
        got = med[0, y0:y0 + 240, x0:x0 + 320]
        wrong += int((got != want).any(-1).sum())
    bad += wrong
    print("%-16s %7.2f us per frame   mismatching pixels on 3 crops: %d" % (name, 1e3 * ms / (cnt * 8), wrong), flush=True)
ctx.close()
sys.exit(1 if bad else 0)
