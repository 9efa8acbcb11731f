"""Freeze the CPU oracle: digests of its outputs on committed inputs.

The oracle (oracle/*.c, oracle/ora_*.py) is the working pin of the HIP path for K1-K9, K11 and the rank-3 pieces -- the
reference holds no vectors for them -- and the GPU tests compare HIP against whatever the oracle says TODAY.  This script
writes what it says into tests/golden/: an edit of the oracle that changes an answer then shows up in review as a changed
fixture, not as silence.  tests/test_oracle_frozen.py recomputes and compares (CPU only).

    python tools/freeze_oracle.py            # (re)write tests/golden/oracle_frozen_inputs.npz and oracle_frozen.json
    python tools/freeze_oracle.py --check    # recompute from the committed inputs and diff against the committed digests

Inputs: synthetic scenes (camkifu_amd.synth) at 48 x 64 and 480 x 640, STORED (the renderer goes through torch's float
kernels and RNG: the fixture must not depend on them); background-model sequences and foreground masks built from
integer arithmetic; classifier weights from synth.cnn_weights (numpy Generator; their digest is part of the fixture, a
changed stream says so before anything else is compared).  Integer / byte outputs are SHA-256 digests; float outputs
(transform, Hough list, pooled maps, softmax) are stored as sampled values and compared to 1e-6 of their scale, labels
and line counts exactly.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
INPUTS = os.path.join(GOLDEN, "oracle_frozen_inputs.npz")
DIGESTS = os.path.join(GOLDEN, "oracle_frozen.json")
DST = np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32)
SCENES = [(48, 64, 11, 0.2), (48, 64, 12, 0.5), (48, 64, 13, 0.0), (480, 640, 21, 0.3), (480, 640, 22, 0.55)]


def sha(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()[:32] + ":%s:%s" % (a.dtype.str, "x".join(map(str, a.shape)))


def sample(a, n=48):
    """a float array as (shape, sum, sum of magnitudes, n evenly spaced values): compared with a tolerance"""
    a = np.ascontiguousarray(a, np.float64).ravel()
    idx = np.linspace(0, a.size - 1, min(n, a.size)).astype(np.int64)
    return dict(size=int(a.size), sum=float(a.sum()), abs_sum=float(np.abs(a).sum()), max=float(np.abs(a).max()) if a.size else 0.0,
                values=[float(v) for v in a[idx]])


def make_inputs():
    from camkifu_amd import synth
    out = {}
    for k, (h, w, seed, density) in enumerate(SCENES):
        sc = synth.scene(h, w, seed=seed, density=density)
        out["frame%d" % k] = sc["frame"].numpy()
        out["corners%d" % k] = np.asarray(sc["corners"], np.float32)
    return out


def hash_noise(shape, seed, bits=3):
    """deterministic small noise from integer arithmetic (no RNG): values 0 .. 2^bits - 1"""
    idx = np.arange(int(np.prod(shape)), dtype=np.uint64).reshape(shape)
    v = (idx + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
    v ^= v >> np.uint64(29)
    v *= np.uint64(0xBF58476D1CE4E5B9)
    v ^= v >> np.uint64(32)
    return (v & np.uint64((1 << bits) - 1)).astype(np.int64)


def mog2_sequence(n=24, side=96):
    """a wood-coloured image with noise, a dark block that appears at frame 8 and a bright one that crosses from frame 14"""
    seq = []
    base = np.empty((side, side, 3), np.int64)
    base[..., 0], base[..., 1], base[..., 2] = 65, 100, 128
    for f in range(n):
        img = base + hash_noise(base.shape, 1000 + f) - 3
        if f >= 8:
            img[20:44, 30:50] = 25 + hash_noise((24, 20, 3), 2000 + f, 2)
        if f >= 14:
            x = 4 * (f - 14)
            img[60:80, x:x + 18] = 228 + hash_noise((20, 18, 3), 3000 + f, 2)
        seq.append(np.clip(img, 0, 255).astype(np.uint8))
    return np.stack(seq)


def fg_mask(side=380):
    """a foreground mask with two stone-sized discs, a hand-sized bar and speckles"""
    y, x = np.mgrid[0:side, 0:side]
    m = np.zeros((side, side), np.uint8)
    for cy, cx in ((70, 110), (250, 190)):
        m[(y - cy) ** 2 + (x - cx) ** 2 <= 81] = 255
    m[150:380, 300:330] = 255
    m[(hash_noise((side, side), 77, 8) == 0)] = 255
    return m


def compute(inputs):
    from camkifu_amd import synth
    from oracle import oracle as ora, ora_grid, ora_stones
    ora.build()
    res = {"_inputs": {k: sha(v) for k, v in sorted(inputs.items())}}
    gobans = []
    for k, (h, w, seed, density) in enumerate(SCENES):
        fr, corners = inputs["frame%d" % k], inputs["corners%d" % k]
        med = ora.median(fr, 15)
        edges, nms, mag, dx, dy = ora.canny(med, 25, 75, want_map=True)
        bl = ora.board_lines(edges)
        M = ora.get_perspective_transform(corners, DST)
        goban = ora.warp_perspective(fr, M)
        gobans.append(goban)
        res["scene%d_%dx%d" % (k, h, w)] = dict(
            median=sha(med), canny_edges=sha(edges), canny_nms_map=sha(nms), canny_magnitude=sha(mag),
            n_contours=int(bl["n_contours"]), biggest_area=float(bl["biggest_area"]), ghost=sha(bl["ghost"]),
            n_lines=int(bl["status"]), lines=sha(bl["lines"]), lines_head=[[float(a), float(b)] for a, b in bl["lines"][:6]],
            transform=[float(v) for v in M.ravel()], goban=sha(goban), median7=sha(ora.median(fr, 7)),
            goban_canny=sha(ora.goban_canny(goban)))
    # K9: the background model over a sequence, learning rates as the stones finder sets them
    seq = mog2_sequence()
    bg = ora.MOG2(seq.shape[1], seq.shape[2], 3)
    masks = np.stack([bg.apply(seq[f], 0.01 if f < 12 else 0.005) for f in range(len(seq))])
    res["mog2"] = dict(sequence=sha(seq), masks=sha(masks), foreground_pixels=[int((m > 0).sum()) for m in masks])
    # K10-K12: random weights and the shipped trained ones on the two 480 x 640 scenes' goban images
    from camkifu_amd.stone.nn_manager import NNManager
    for name, W in (("random", synth.cnn_weights()), ("trained", NNManager.init_net())):
        entry = {"weights": {k: sha(np.asarray(v, np.float32)) for k, v in sorted(W.items())}}
        for g in (3, 4):
            y, p2, p4 = ora.cnn_region_maps(W, gobans[g])
            labels, conf = ora.decode_all(y)
            entry["goban%d" % g] = dict(pool2=sample(p2), pool4=sample(p4), softmax=sample(y), labels=sha(np.asarray(labels, np.uint8)),
                                        region_argmax=sha(y.argmax(1).astype(np.int32)), confidence=sample(conf))
        res["cnn_" + name] = entry
    # rank 3: SfContours.find_stones and StonesFinder.find_intersections on a goban image
    from camkifu_amd.stone.stonesfinder import PosGrid
    pg = PosGrid(380)
    stones, zones, mask, _ = ora_stones.find_stones(gobans[3], fg_mask(), want_all=True)
    grid, found, canny = ora_grid.find_intersections(gobans[3], pg.mtx, pg.zones(1.0), want_lines=True)
    res["rank3"] = dict(fg_mask=sha(fg_mask()), stones=sha(np.asarray(stones, np.uint8)), zones=sha(np.asarray(zones, np.int16)),
                        hull_mask=sha(mask), grid=sha(np.asarray(grid, np.int16)), grid_canny=sha(canny), zones_with_lines=len(found),
                        lines=sha(np.array([[r, c] + [int(v) for v in ln] for (r, c), lns in sorted(found.items()) for ln in lns], np.int32)))
    return res


def differences(want, got, path="", tol=1e-6):
    """-> list of human-readable differences between two digest trees"""
    out = []
    if isinstance(want, dict) and isinstance(got, dict):
        if set(want) == {"size", "sum", "abs_sum", "max", "values"}:                     # a sampled float array
            if want["size"] != got["size"]:
                return ["%s: size %d -> %d" % (path, want["size"], got["size"])]
            scale = max(want["max"], 1e-30)
            worst = max([abs(a - b) for a, b in zip(want["values"], got["values"])] + [abs(want["max"] - got["max"])]) / scale
            mean_d = abs(want["abs_sum"] - got["abs_sum"]) / max(want["abs_sum"], 1e-30)
            if worst > tol or mean_d > tol:
                out.append("%s: values moved by %.3g of scale (mean magnitude by %.3g)" % (path, worst, mean_d))
            return out
        for k in sorted(set(want) | set(got)):
            if k not in want or k not in got:
                out.append("%s/%s: %s" % (path, k, "new" if k not in want else "gone"))
            else:
                out += differences(want[k], got[k], path + "/" + k, tol)
    elif isinstance(want, list) and isinstance(got, list) and len(want) == len(got):
        for i, (a, b) in enumerate(zip(want, got)):
            out += differences(a, b, "%s[%d]" % (path, i), tol)
    elif isinstance(want, float) or isinstance(got, float):
        if abs(float(want) - float(got)) > tol * max(1.0, abs(float(want))):
            out.append("%s: %r -> %r" % (path, want, got))
    elif want != got:
        out.append("%s: %r -> %r" % (path, want, got))
    return out


def main():
    if "--check" in sys.argv:
        inputs = dict(np.load(INPUTS))
        diff = differences(json.load(open(DIGESTS)), json.loads(json.dumps(compute(inputs))))
        print("\n".join(diff) if diff else "oracle outputs equal the committed digests")
        return 1 if diff else 0
    os.makedirs(GOLDEN, exist_ok=True)
    inputs = make_inputs()
    np.savez_compressed(INPUTS, **inputs)
    with open(DIGESTS, "w") as f:
        json.dump(compute(dict(np.load(INPUTS))), f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote %s (%d KB) and %s" % (os.path.relpath(INPUTS, ROOT), os.path.getsize(INPUTS) // 1024, os.path.relpath(DIGESTS, ROOT)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
