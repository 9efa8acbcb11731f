"""What the split-precision classifier would lose with its two cross terms in 8 (or 6) bits -- a float64 simulation.

The f16x2 mode writes a product a*w as a_hi*w_hi + a_hi*w_lo + a_lo*w_hi on the fp16 MFMA (three instructions, DESIGN.md
4).  The cross terms are 2^-11 of the main term, so their operands need few bits: this script evaluates the network of
nn_manager.py:277-298 in float64 with the cross terms' operands rounded to a narrow format (OCP e4m3, e2m3 with a power-of-
two scale per 32 channels, int8 with one scale per layer) and prints the error of the two pooled maps against the plain
float64 evaluation, relative to the map's scale -- the quantity tests/test_gpu_parity.py::
test_cnn_filter_maps_below_the_softmax holds to 1e-4.  CPU only (torch float64); no HIP, no oracle.

    python tools/sim_split_q8.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def f16_rtz(x):
    """round toward zero to fp16 (v_cvt_pkrtz), as float64"""
    h = x.to(torch.float16).double()
    over = h.abs() > x.abs()
    if over.any():
        hi = h.to(torch.float16)
        step = torch.nextafter(hi, torch.zeros_like(hi)).double()
        h = torch.where(over, step, h)
    return h


def f16(x):
    return x.to(torch.float16).double()


def fp_round(x, mant, emin, vmax):
    """round to nearest even to a float format with `mant` explicit mantissa bits, smallest normal exponent emin,
    subnormals below it, saturating at vmax"""
    ax = x.abs()
    e = torch.floor(torch.log2(torch.clamp(ax, min=1e-300)))
    e = torch.clamp(e, min=emin)
    q = torch.pow(2.0, e - mant)
    r = torch.round(ax / q) * q          # torch.round is half-to-even
    r = torch.clamp(r, max=vmax)
    return torch.sign(x) * r


def e4m3(x):
    return fp_round(x, 3, -6, 448.0)


def e2m3_block(x, dim):
    """e2m3 (values up to 7.5, step 0.125 below 1) with one power-of-two scale per 32 consecutive elements along dim,
    chosen so that the block's maximum lands in [4, 8) before rounding (saturating at 7.5)"""
    shp = x.shape
    x = x.movedim(dim, -1)
    n = x.shape[-1]
    pad = (-n) % 32
    xp = F.pad(x, (0, pad)).reshape(*x.shape[:-1], -1, 32)
    m = xp.abs().amax(-1, keepdim=True)
    s = torch.pow(2.0, torch.floor(torch.log2(torch.clamp(m, min=1e-300))) - 2)
    q = fp_round(xp / s, 3, 0, 7.5) * s
    q = q.reshape(*x.shape[:-1], -1)[..., :n].movedim(-1, dim)
    assert q.shape == shp
    return q


def e2m3_fixed(x, s):
    """e2m3 with one fixed scale s (a constant of the layer)"""
    return fp_round(x / s, 3, 0, 7.5) * s


def int8_layer(x):
    s = x.abs().max() / 127.0
    return torch.round(x / s).clamp(-127, 127) * s


def run(W, gobans, scheme):
    w = {k: torch.from_numpy(np.ascontiguousarray(v)).double() for k, v in W.items()}
    org = [0, 40, 80, 120, 160, 200, 240, 280, 320, 340]
    g = torch.from_numpy(gobans).permute(0, 3, 1, 2).double()
    patches = torch.stack([g[:, :, a:a + 40, b:b + 40] for a in org for b in org], 1).reshape(-1, 3, 40, 40)

    def kern(k):
        return k.flip(0, 1).permute(3, 2, 0, 1).contiguous()

    def conv(x, k, b, first=False):
        k = kern(k)
        if scheme == "f64":
            return F.relu(F.conv2d(x, k, b))
        # weights: hi + lo in fp16 of 256 w (the product's stored form), activations: hi (rtz) + lo
        ks = k * 256.0
        k_hi = f16(ks)
        k_lo = f16(ks - k_hi)
        if first:                           # u8 pixels are exact in fp16: two terms, as the product computes them
            return F.relu((F.conv2d(x, k_hi) + F.conv2d(x, k_lo)) / 256.0 + b.view(1, -1, 1, 1))
        x_hi = f16_rtz(x)
        x_lo = f16(x - x_hi)
        main = F.conv2d(x_hi, k_hi)
        if scheme == "f16x2":
            cross = F.conv2d(x_hi, k_lo) + F.conv2d(x_lo, k_hi)
        elif scheme == "e4m3":
            # block scales: 1 for the hi operands, 2^-11 for the lo operands (weights' lo: 2^-11 too)
            cross = (F.conv2d(e4m3(x_hi), e4m3(k_lo * 2048.0)) + F.conv2d(e4m3(x_lo * 2048.0), e4m3(k_hi))) / 2048.0
        elif scheme == "e4m3_hi_only_w":
            # weights' hi operand taken in fp16 for the second cross term is not possible on one instruction; this variant
            # rounds only the activations (what the error would be with exact weights): a diagnostic
            cross = (F.conv2d(e4m3(x_hi), k_lo) + F.conv2d(e4m3(x_lo * 2048.0), k_hi) / 2048.0)
        elif scheme == "e2m3":
            cross = (F.conv2d(e2m3_block(x_hi, 1), e2m3_block(k_lo, 1)) + F.conv2d(e2m3_block(x_lo, 1), e2m3_block(k_hi, 1)))
        elif scheme == "int8":
            cross = (F.conv2d(int8_layer(x_hi), int8_layer(k_lo)) + F.conv2d(int8_layer(x_lo), int8_layer(k_hi)))
        elif scheme == "none":
            cross = 0.0
        else:
            raise ValueError(scheme)
        return F.relu((main + cross) / 256.0 + b.view(1, -1, 1, 1))

    x = F.max_pool2d(conv(conv(patches, w["c1w"], w["c1b"], first=True), w["c2w"], w["c2b"]), 2)
    p2 = x
    x = F.max_pool2d(conv(conv(x, w["c3w"], w["c3b"]), w["c4w"], w["c4b"]), 2)
    return p2, x


def main():
    from camkifu_amd import synth
    from camkifu_amd.stone.nn_manager import NNManager
    from oracle import oracle as ora
    ora.build()
    dst = np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32)
    gobans = []
    for seed, dens in ((31, 0.45), (71, 0.15)):
        sc = synth.scene(480, 640, seed=seed, density=dens)
        gobans.append(ora.warp_perspective(sc["frame"].numpy(), ora.get_perspective_transform(sc["corners"], dst)))
    gobans = np.stack(gobans)
    for name, W in (("trained", NNManager.init_net()), ("random", synth.cnn_weights())):
        ref2, ref4 = run(W, gobans, "f64")
        print("%s weights: map scales %.3g / %.3g" % (name, ref2.abs().max(), ref4.abs().max()))
        for scheme in ("f16x2", "e4m3", "e2m3", "int8", "none"):
            p2, p4 = run(W, gobans, scheme)
            e2 = ((p2 - ref2).abs().max() / ref2.abs().max()).item()
            e4 = ((p4 - ref4).abs().max() / ref4.abs().max()).item()
            print("  %-8s pool2 %.3g  pool4 %.3g  of scale (bar 1e-4)" % (scheme, e2, e4))


if __name__ == "__main__":
    main()
