"""Loading the stone classifier from a Keras-1 HDF5 file, as the reference does with
`keras.models.load_model(KERAS_MODEL_FILE)` (stone/nn_manager.py:22, 65-73) -- without Keras or h5py:
`h5lite` reads the file, this module maps the layers of `NNManager.create_net`
(nn_manager.py:277-298) onto the twelve arrays `ck_cnn_set_weights` takes.

Accepted layouts: a full `model.save()` file (weights under /model_weights) or a `save_weights()` file
(layer groups at the root); convolution kernels in 'tf' dim ordering [rows, cols, in, out] (what the
reference's input_shape=(40, 40, 3) implies) or in 'th' ordering [out, in, rows, cols], in which case
the kernels are transposed and the rows of the first dense layer are re-ordered from (c, h, w) to
(h, w, c) flattening.  Kernels stay un-flipped: the library applies them as true convolutions, like
Keras-1 on Theano."""
import numpy as np

from . import h5lite

WEIGHT_ORDER = ("c1w", "c1b", "c2w", "c2b", "c3w", "c3b", "c4w", "c4b", "d1w", "d1b", "d2w", "d2b")
WEIGHT_SHAPES = dict(c1w=(5, 5, 3, 32), c1b=(32,), c2w=(5, 5, 32, 32), c2b=(32,),
                     c3w=(3, 3, 32, 90), c3b=(90,), c4w=(3, 3, 90, 90), c4b=(90,),
                     d1w=(3240, 160), d1b=(160,), d2w=(160, 81), d2b=(81,))


class ModelFormatError(ValueError):
    pass


def _text(x):
    return x.decode("utf-8") if isinstance(x, (bytes, np.bytes_)) else str(x)


def read_layer_weights(path):
    """-> [(layer name, [arrays in weight_names order])] for every layer that has weights"""
    f = h5lite.File(path)
    g = f["model_weights"] if "model_weights" in f else f
    if "layer_names" not in g.attrs:
        raise ModelFormatError("%s: no layer_names attribute (not a Keras model / weights file)" % path)
    out = []
    for ln in np.atleast_1d(g.attrs["layer_names"]):
        lg = g[_text(ln)]
        names = [_text(w) for w in np.atleast_1d(lg.attrs.get("weight_names", []))]
        if names:
            out.append((_text(ln), [np.asarray(lg[w].read(), np.float32) for w in names]))
    return out


def load_model(path):
    """-> dict of the twelve float32 arrays in Keras-1 'tf' layout (capi.WEIGHT_ORDER)"""
    layers = read_layer_weights(path)
    if len(layers) != 6 or any(len(ws) != 2 for _, ws in layers):
        raise ModelFormatError("%s: expected 4 convolution + 2 dense layers with (W, b), found %s"
                               % (path, [(n, [w.shape for w in ws]) for n, ws in layers]))
    arrays = [a for _, ws in layers for a in ws]
    weights, th = {}, False
    for key, a in zip(WEIGHT_ORDER, arrays):
        want = WEIGHT_SHAPES[key]
        if key in ("c1w", "c2w", "c3w", "c4w") and a.shape != want and a.shape == (want[3], want[2], want[0], want[1]):
            a = a.transpose(2, 3, 1, 0)                      # 'th' [out, in, rows, cols] -> 'tf'
            th = True
        if a.shape != want:
            raise ModelFormatError("%s: %s has shape %s, the classifier needs %s" % (path, key, a.shape, want))
        weights[key] = np.ascontiguousarray(a, np.float32)
    if th:                                                   # Flatten() saw (c, h, w): re-order to (h, w, c)
        weights["d1w"] = np.ascontiguousarray(weights["d1w"].reshape(90, 6, 6, 160).transpose(1, 2, 0, 3).reshape(3240, 160))
    return weights
