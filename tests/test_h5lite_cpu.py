"""h5lite (from-scratch HDF5 reader) and the Keras-1 model loader, checked against
files written by the REAL h5py / libhdf5 (tools/make_h5_fixtures.py, run once in the build container with
/opt/conda/bin/python3.9; expected values in tests/golden/h5_structures.json)."""
import json
import os

import numpy as np
import pytest

from camkifu_amd.stone import h5lite, keras1
from camkifu_amd.stone.nn_manager import NNManager, KERAS_MODEL_FILE

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _plain(v):
    if isinstance(v, np.ndarray):
        return [_plain(x) for x in v.tolist()]
    if isinstance(v, (bytes, np.bytes_)):
        return v.decode("utf-8")
    if isinstance(v, np.generic):
        return v.item()
    return v


def test_structures_written_by_libhdf5():
    f = h5lite.File(os.path.join(GOLD, "h5_structures.h5"))
    man = json.load(open(os.path.join(GOLD, "h5_structures.json"), encoding="utf-8"))
    for path, exp in man["datasets"].items():
        a = f[path].read()
        assert list(a.shape) == exp["shape"] and a.dtype.str == exp["dtype"], path
        assert float(np.asarray(a, np.float64).sum()) == pytest.approx(exp["sum"], rel=1e-9, abs=1e-9), path
        assert np.asarray(a).reshape(-1)[:4].tolist() == pytest.approx(exp["first"]), path
    for path, attrs in man["attrs"].items():
        got = {k: _plain(v) for k, v in f[path].attrs.items()} if path != "/" else {k: _plain(v) for k, v in f.attrs.items()}
        assert got == attrs, path
    assert sorted(f["big"].keys()) == man["big_keys"]          # 40 links: more than one symbol-table node
    assert f["big/item_17"].read().tolist() == [17, 17, 17]
    assert sorted(p for p, _ in h5lite.visit(f["nested"])) == ["/deeper/chunked", "/f16", "/zipped"]
    with pytest.raises(KeyError):
        f["nested/nope"]


def test_rejects_what_it_does_not_read(tmp_path):
    with pytest.raises(h5lite.H5Error):
        h5lite.File(b"not an hdf5 file at all" * 40)
    raw = bytearray(open(os.path.join(GOLD, "h5_structures.h5"), "rb").read())
    raw[8] = 9                                                   # unknown superblock version
    with pytest.raises(h5lite.H5Error):
        h5lite.File(bytes(raw))
    # a user block in front of the superblock (signature at 512) is fine
    moved = bytes(512) + open(os.path.join(GOLD, "h5_structures.h5"), "rb").read()
    assert h5lite.File(moved)["compact"].read().tolist() == [1.5, -2.5, 3.5, 4.5]


def test_keras1_model_file_loads_like_the_reference_expects():
    """camkifu_amd/data/keras.h5 has the layout Keras 1.2 model.save() writes (root attrs, model_weights group,
    layer_names / weight_names, 'tf' kernels); NNManager.init_net loads it as the reference loads keras.h5"""
    layers = keras1.read_layer_weights(KERAS_MODEL_FILE)
    assert [n for n, _ in layers] == ["convolution2d_1", "convolution2d_2", "convolution2d_3", "convolution2d_4",
                                      "dense_1", "dense_2"]
    W = NNManager.init_net()
    assert tuple(W) == keras1.WEIGHT_ORDER
    assert all(W[k].shape == keras1.WEIGHT_SHAPES[k] and W[k].dtype == np.float32 for k in W)
    assert sum(v.size for v in W.values()) == 658665             # parameter count of create_net
    f = h5lite.File(KERAS_MODEL_FILE)
    assert _plain(f.attrs["keras_version"]) == "1.2.2"
    assert json.loads(_plain(f.attrs["model_config"]))["class_name"] == "Sequential"


def test_theano_dim_ordering_is_converted(monkeypatch):
    W = NNManager.init_net()
    th = []
    for name, (kw, kb) in zip(["c1", "c2", "c3", "c4"], [("c1w", "c1b"), ("c2w", "c2b"), ("c3w", "c3b"), ("c4w", "c4b")]):
        th.append((name, [W[kw].transpose(3, 2, 0, 1).copy(), W[kb]]))                # [out, in, rows, cols]
    d1_th = W["d1w"].reshape(6, 6, 90, 160).transpose(2, 0, 1, 3).reshape(3240, 160)  # rows in (c, h, w) order
    th += [("d1", [d1_th, W["d1b"]]), ("d2", [W["d2w"], W["d2b"]])]
    monkeypatch.setattr(keras1, "read_layer_weights", lambda path: th)
    got = keras1.load_model("whatever.h5")
    assert all(np.array_equal(got[k], W[k]) for k in W)
    bad = th[:-1]
    monkeypatch.setattr(keras1, "read_layer_weights", lambda path: bad)
    with pytest.raises(keras1.ModelFormatError):
        keras1.load_model("whatever.h5")
