"""Pin the oracle against every known answer the reference's own tests/doctests hold for the
hot path: label codec (test/camkifu/stone/test_tmanager.py:18-27) and the
geometry tables derived from nn_manager.py / stonesfinder.py."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "reference_known_answers.json")) as f:
    GOLD = json.load(f)


def test_compute_stones(ora):
    for k, v in GOLD["compute_stones"].items():
        assert ora.compute_stones(int(k)) == v


def test_compute_label_roundtrip(ora):
    for lab in range(81):
        assert ora.compute_label(ora.compute_stones(lab)) == lab


def test_class_indices(ora):
    ci = ora.class_indices()
    assert ci.shape == (4, 3, 27)
    for k, v in GOLD["class_indices"].items():
        d, c = map(int, k.split(","))
        assert list(ci[d, c]) == v


def test_patch_origins(ora):
    origins = []
    for i in range(10):
        rs, re, cs, ce = ora.subregion(i, i)
        x0, x1, y0, y1 = ora.nn_rect(rs, re, cs, ce)
        assert x1 - x0 == 40 and y1 - y0 == 40 and (x0, x1) == (y0, y1)
        origins.append(x0)
    assert origins == GOLD["patch_origins"]
    assert ora.subregion(9, 0) == (17, 19, 0, 2)


def test_sf_getrect_and_posgrid(ora):
    for k, v in GOLD["sf_getrect"].items():
        r, c = map(int, k.split(","))
        assert list(ora.sf_getrect(r, c)) == v
    g = ora.posgrid()
    for i in range(19):
        for j in range(19):
            assert tuple(g[i, j]) == (10 + 20 * i, 10 + 20 * j)


def test_decode_all_matches_codec(ora):
    rng = np.random.default_rng(5)
    y = rng.random((100, 81)).astype(np.float32)
    labels, conf = ora.decode_all(y)
    sym = {'E': 0, 'B': 1, 'W': 2}
    exp = np.zeros((19, 19), np.uint8)
    expc = np.zeros((19, 19))
    for i in range(10):
        for j in range(10):
            yy = y[i * 10 + j]
            lab = int(np.argmax(yy))
            rs, re, cs, ce = ora.subregion(i, j)
            st = np.array([sym[s] for s in ora.compute_stones(lab)]).reshape(2, 2)
            exp[rs:re, cs:ce] = st
            tot = 0.0              # numpy-1.x: python sum() of float32 scalars promotes to
            for v in yy:           # float64 and accumulates in index order
                tot = tot + float(v)
            expc[rs:re, cs:ce] = float(max(yy)) / tot
    assert np.array_equal(labels, exp)
    assert np.array_equal(conf, expc)
