#!/opt/conda/bin/python3.9
"""Generates the HDF5 fixtures of tests/golden/ with the REAL h5py / libhdf5 (only available in this
container's /opt/conda python3.9 -- run:  /opt/conda/bin/python3.9 tools/make_h5_fixtures.py ).
The files pin camkifu_amd/stone/h5lite.py (a from-scratch reader) against what libhdf5 writes.

  tests/golden/h5_structures.h5 (+ .json manifest of expected values)
      groups (a 40-entry one: several symbol-table nodes), contiguous / compact / chunked / deflate+shuffle
      datasets of several types, fixed and variable-length string attributes, numeric attributes.
  camkifu_amd/data/keras.h5
      the trained stone classifier (tools/train_cnn.py -> tools/out/cnn_weights.npz) written the way Keras 1.2 `model.save`
      lays a model out (the file the reference loads: stone/nn_manager.py:22, 65-73): root attributes
      keras_version / model_config, group model_weights with attribute layer_names, one group per layer
      with attribute weight_names and float32 datasets in 'tf' dim ordering.
"""
import json
import os
import sys

import h5py
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def structures():
    rng = np.random.default_rng(7)
    path = os.path.join(GOLD, "h5_structures.h5")
    man = {"datasets": {}, "attrs": {}}
    with h5py.File(path, "w") as f:
        f.attrs["title"] = np.bytes_("h5lite fixture")
        f.attrs["vlen_title"] = "variable length ünïcode"
        f.attrs["answer"] = np.int32(42)
        f.attrs["pi"] = np.float64(3.141592653589793)
        f.attrs["names"] = np.array([b"alpha", b"be", b"gamma_delta"])
        f.attrs["vlen_names"] = np.array(["one", "three", ""], dtype=h5py.string_dtype())
        f.attrs["matrix"] = np.arange(6, dtype=np.int16).reshape(2, 3)
        man["attrs"]["/"] = {"title": "h5lite fixture", "vlen_title": "variable length ünïcode", "answer": 42,
                             "pi": 3.141592653589793, "names": ["alpha", "be", "gamma_delta"],
                             "vlen_names": ["one", "three", ""], "matrix": [[0, 1, 2], [3, 4, 5]]}

        def put(name, arr, **kw):
            d = f.create_dataset(name, data=arr, **kw)
            man["datasets"]["/" + name] = {"dtype": arr.dtype.str, "shape": list(arr.shape),
                                           "sum": float(np.asarray(arr, np.float64).sum()) if arr.dtype.kind in "fiu" else None,
                                           "first": np.asarray(arr).reshape(-1)[:4].tolist() if arr.dtype.kind in "fiu" else None}
            return d
        put("f32", rng.standard_normal((5, 7)).astype(np.float32))
        put("f64_be", rng.standard_normal(9).astype(">f8"))
        put("u8", rng.integers(0, 256, (3, 4, 5), dtype=np.uint8))
        put("i64", rng.integers(-2 ** 40, 2 ** 40, 11, dtype=np.int64))
        put("scalar", np.float32(2.5))
        d = put("nested/deeper/chunked", rng.standard_normal((33, 10)).astype(np.float32), chunks=(8, 4))
        d.attrs["weight_names"] = np.array([b"w_W", b"w_b"])
        man["attrs"]["/nested/deeper/chunked"] = {"weight_names": ["w_W", "w_b"]}
        put("nested/zipped", rng.integers(0, 50, (40, 40)).astype(np.int32), chunks=(16, 16), compression="gzip",
            shuffle=True)
        put("nested/f16", rng.standard_normal(6).astype(np.float16))
        # compact layout needs the low-level API
        space = h5py.h5s.create_simple((4,))
        dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
        dcpl.set_layout(h5py.h5d.COMPACT)
        arr = np.array([1.5, -2.5, 3.5, 4.5], np.float32)
        did = h5py.h5d.create(f.id, b"compact", h5py.h5t.IEEE_F32LE, space, dcpl=dcpl)
        did.write(h5py.h5s.ALL, h5py.h5s.ALL, arr)
        man["datasets"]["/compact"] = {"dtype": "<f4", "shape": [4], "sum": float(arr.sum()), "first": arr.tolist()}
        big = f.create_group("big")
        for i in range(40):
            big.create_dataset("item_%02d" % i, data=np.full(3, i, np.int32))
        man["big_keys"] = ["item_%02d" % i for i in range(40)]
        f.create_dataset("never_written", shape=(2, 2), dtype=np.float32)
        man["datasets"]["/never_written"] = {"dtype": "<f4", "shape": [2, 2], "sum": 0.0, "first": [0.0, 0.0, 0.0, 0.0]}
    json.dump(man, open(os.path.join(GOLD, "h5_structures.json"), "w"), indent=1, ensure_ascii=False)
    print("wrote", path, os.path.getsize(path), "bytes")


def keras_model():
    W = np.load(os.path.join(ROOT, "tools", "out", "cnn_weights.npz"))
    path = os.path.join(ROOT, "camkifu_amd", "data", "keras.h5")          # the model ships inside the package
    layers = [("convolution2d_1", ["c1w", "c1b"]), ("convolution2d_2", ["c2w", "c2b"]), ("maxpooling2d_1", []),
              ("dropout_1", []), ("convolution2d_3", ["c3w", "c3b"]), ("convolution2d_4", ["c4w", "c4b"]),
              ("maxpooling2d_2", []), ("dropout_2", []), ("flatten_1", []), ("dense_1", ["d1w", "d1b"]),
              ("dropout_3", []), ("dense_2", ["d2w", "d2b"])]
    cfg = {"class_name": "Sequential", "config": [{"class_name": "Convolution2D", "config": {
        "name": "convolution2d_1", "nb_filter": 32, "nb_row": 5, "nb_col": 5, "dim_ordering": "tf",
        "activation": "relu", "batch_input_shape": [None, 40, 40, 3], "border_mode": "valid"}}]}
    with h5py.File(path, "w") as f:
        f.attrs["keras_version"] = "1.2.2".encode("utf8")
        f.attrs["model_config"] = json.dumps(cfg).encode("utf8")
        g = f.create_group("model_weights")
        g.attrs["layer_names"] = [name.encode("utf8") for name, _ in layers]
        for name, keys in layers:
            lg = g.create_group(name)
            wnames = [("%s_%s" % (name, "W" if k.endswith("w") else "b")).encode("utf8") for k in keys]
            lg.attrs["weight_names"] = wnames
            for wn, k in zip(wnames, keys):
                val = W[k].astype(np.float32)
                ds = lg.create_dataset(wn.decode(), val.shape, dtype=val.dtype)
                ds[...] = val
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    structures()
    keras_model()
