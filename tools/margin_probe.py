"""Where can the reduced-precision classifier modes actually flip?  

The shipped model is so sure of itself on rendered boards that operand precision cannot show: no region of 60 gobans had a
top-2 softmax margin under 0.05 (profiles/r05_fuzz_board_bf16.txt).  Here the margins and the confidences are MADE small:
random weights, and blends (1 - a) random + a trained that behave like a net early in its training, on gobans with up to
sigma = 12 of sensor noise.  Per model and mode (bf16, f16q8, and the f32-equivalent f16x2 as a control) against the CPU
oracle's f32 chain:
  * labels (arg max of a region's 81 outputs) that differ, by the ORACLE's top-2 margin;
  * `conf > 0.6` decisions (the policy's gate, stone/sf_neural.py:18) that differ, by the oracle's |conf - 0.6|;
  * the mode's largest |softmax - oracle| (its measured error e).
Gate (exit code): no label differs where the oracle's margin exceeds 2 e, no decision differs where |conf - 0.6| exceeds
2 e -- an output cannot cross a gap wider than the error on both sides of it.  Everything below that is reported, not judged:
there the f32 chain itself is one rounding away from the other answer.
usage: python tools/margin_probe.py [gobans per model = 16] [seed = 1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from camkifu_amd import capi, synth
from camkifu_amd.stone.nn_manager import NNManager
from oracle import oracle as ora

MODES = (("bf16", capi.CK_CNN_BF16), ("f16q8", capi.CK_CNN_F16Q8), ("f16x2", capi.CK_CNN_F16X2))
MARGIN_EDGES = (1e-4, 1e-3, 1e-2, 0.05)
CONF_GATE = 0.6


def models(seed):
    trained = NNManager.init_net()
    rnd = synth.cnn_weights(seed=synth.SEED + seed)
    out = [("trained", trained)]
    for a in (0.75, 0.5, 0.25):
        out.append(("blend %.2f trained" % a, {k: ((1 - a) * rnd[k] + a * trained[k]).astype(np.float32) for k in trained}))
    out.append(("random", rnd))
    return out


def bucket(v, edges):
    return int(np.searchsorted(np.asarray(edges), v, side="right"))


def probe(ck, W, gobans):
    """-> per mode dict(regions, err, label_diff[by margin bucket], conf_diff[by distance bucket], worst margin of a label
    difference, worst distance of a decision difference) and the oracle's own statistics"""
    yo = np.stack([ora.cnn_predict_regions(W, g) for g in gobans])              # (n, 100, 81)
    top2 = np.sort(yo, axis=2)[:, :, -2:]
    margin = (top2[:, :, 1] - top2[:, :, 0]).reshape(-1)
    lab_o = yo.argmax(2).reshape(-1)
    conf_o = (yo.max(2).astype(np.float64) / yo.sum(2, dtype=np.float64)).reshape(-1)
    dist = np.abs(conf_o - CONF_GATE)
    stats = dict(regions=len(lab_o), margin_below=[int((margin < e).sum()) for e in MARGIN_EDGES],
                 conf_in_05_07=int(((conf_o > 0.5) & (conf_o < 0.7)).sum()))
    ck.cnn_set_weights(W)
    res = {}
    for name, mode in MODES:
        ck.cnn_set_mode(mode)
        y = np.asarray(ck.cnn_predict(gobans)[0])
        err = float(np.abs(y - yo).max())
        lab = y.argmax(2).reshape(-1)
        conf = (y.max(2).astype(np.float64) / y.sum(2, dtype=np.float64)).reshape(-1)
        ld = lab != lab_o
        cd = (conf > CONF_GATE) != (conf_o > CONF_GATE)
        res[name] = dict(err=err, label_diff=int(ld.sum()), conf_diff=int(cd.sum()),
                         label_diff_by_margin=np.bincount([bucket(m, MARGIN_EDGES) for m in margin[ld]], minlength=len(MARGIN_EDGES) + 1).tolist(),
                         worst_margin=float(margin[ld].max()) if ld.any() else 0.0,
                         worst_dist=float(dist[cd].max()) if cd.any() else 0.0,
                         ok=bool((not ld.any() or margin[ld].max() <= 2 * err) and (not cd.any() or dist[cd].max() <= 2 * err)))
    ck.cnn_set_mode(capi.CK_CNN_DEFAULT)
    return stats, res


def gobans_for(n, seed):
    dst = np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32)
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        sc = synth.scene(480, 640, seed=seed * 1000 + k, density=float(rng.uniform(0.0, 0.7)), noise=float(rng.choice([3.0, 6.0, 12.0])))
        out.append(ora.warp_perspective(sc["frame"].numpy(), ora.get_perspective_transform(sc["corners"], dst)))
    return np.stack(out)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ora.build()
    ck = capi.Context(0)
    gob = gobans_for(n, seed)
    ok = True
    print("margin buckets (oracle top-2 softmax margin): < %s, >= %s" % (", < ".join("%g" % e for e in MARGIN_EDGES), MARGIN_EDGES[-1]))
    for name, W in models(seed):
        stats, res = probe(ck, W, gob)
        print("\n%s: %d regions; oracle margins below %s: %s; confidences in (0.5, 0.7): %d"
              % (name, stats["regions"], "/".join("%g" % e for e in MARGIN_EDGES), stats["margin_below"], stats["conf_in_05_07"]), flush=True)
        for mode, r in res.items():
            print("  %-6s max |softmax - oracle| %.3g   labels differing %d %s (widest margin crossed %.3g)   conf > 0.6 decisions differing %d "
                  "(furthest from 0.6: %.3g)   %s" % (mode, r["err"], r["label_diff"], r["label_diff_by_margin"], r["worst_margin"], r["conf_diff"],
                                                      r["worst_dist"], "ok" if r["ok"] else "BEYOND 2 x ITS ERROR"), flush=True)
            ok = ok and r["ok"]
    ck.close()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
