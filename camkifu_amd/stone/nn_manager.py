"""NNManager: patch geometry, base-3 label codec and the classifier itself (mirror of the
inference side of the reference's stone/nn_manager.py:29-131, 216-298, 360-382).  The Keras
model becomes a weight dictionary handed to the HIP library (K10..K12); training and the
labelling GUI are out of scope."""
import math
import os
import threading

import numpy as np

from .. import capi, cvconf
from ..golib_shim import gsize, E, B, W

colors = {E: 0, B: 1, W: 2}
rcolors = {0: E, 1: B, 2: W}
# The trained model file (reference: KERAS_MODEL_FILE = cvconf.train_dir + "/model/keras.h5",
# stone/nn_manager.py:22).  The author's file cannot be fetched here; the repository ships a classifier
# trained on the synthetic renderer (tools/train_cnn.py) in the same Keras-1 HDF5 layout.
KERAS_MODEL_FILE = os.environ.get("CAMKIFU_KERAS_MODEL") or os.path.join(
    os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "keras.h5")
GOLDEN_WEIGHTS = KERAS_MODEL_FILE


class NNManager:
    _network = None          # weight dict, shared like the reference's lazily created Keras model
    _netlock = threading.RLock()
    _depth = 3

    def __init__(self):
        self.canonical_shape = (cvconf.canonical_size, cvconf.canonical_size)
        self.split = 10
        self.step = (gsize + 1) // self.split
        self.nb_classes = 3 ** (int((gsize + 1) / self.split) ** 2)
        x0, x1, y0, y1 = self._get_rect_nn(*self._subregion(0, 0))
        self.r_width = y1 - y0
        self.c_width = x1 - x0
        self.c_indices = None

    @staticmethod
    def get_net(download=False):
        with NNManager._netlock:
            if NNManager._network is None:
                NNManager._network = NNManager.init_net()
            return NNManager._network

    @staticmethod
    def init_net(download=False):
        """weights in Keras-1 'tf' layout: the model file when present (stone/nn_manager.py:65-73 loads it
        with keras.models.load_model; here stone/keras1.py + h5lite read it), else create_net():
        seeded He-normal"""
        if os.path.isfile(KERAS_MODEL_FILE):
            return NNManager.load_model(KERAS_MODEL_FILE)
        return NNManager.create_net()

    @staticmethod
    def load_model(path):
        from . import keras1
        return keras1.load_model(path)

    @staticmethod
    def create_net():
        from .. import synth
        return synth.cnn_weights()

    def _subregion(self, i, j):
        assert 0 <= i < self.split and 0 <= j < self.split
        step = self.step
        rs, re = i * step, (i + 1) * step
        if gsize - rs < step:
            rs, re = gsize - step, gsize
        cs, ce = j * step, (j + 1) * step
        if gsize - cs < step:
            cs, ce = gsize - step, gsize
        return rs, re, cs, ce

    def get_region_indices(self, r, c):
        return r // self.step, c // self.step

    def getrect(self, r, c, re=0, ce=0):
        x0 = int(r * self.canonical_shape[0] / gsize)
        y0 = int(c * self.canonical_shape[1] / gsize)
        re = (re + 1) if 0 < re else (r + 1)
        ce = (ce + 1) if 0 < ce else (c + 1)
        return x0, y0, int(re * self.canonical_shape[0] / gsize), int(ce * self.canonical_shape[1] / gsize)

    def _get_rect_nn(self, rs, re, cs, ce):
        x0, y0, _, _ = self.getrect(rs, cs)
        _, _, x1, y1 = self.getrect(re - 1, ce - 1)
        if hasattr(self, 'c_width'):
            if x1 - x0 != self.c_width:
                x0 = x1 - self.c_width
            if y1 - y0 != self.r_width:
                y0 = y1 - self.r_width
        return x0, x1, y0, y1

    def _get_x(self, i, j, img):
        x0, x1, y0, y1 = self._get_rect_nn(*self._subregion(i, j))
        return img[x0:x1, y0:y1]

    @staticmethod
    def compute_label(rs, re, cs, ce, stones):
        label = 0
        for r in range(rs, re):
            for c in range(cs, ce):
                label += colors[stones[r, c]] * 3 ** ((r - rs) * (ce - cs) + (c - cs))
        return label

    @staticmethod
    def compute_stones(label, dimension=4):
        k = label
        stones = np.ndarray(dimension, dtype=object)
        for i in reversed(range(dimension)):
            digit = int(k / (3 ** i))
            stones[i] = rcolors[digit]
            k %= 3 ** i
        return stones

    def class_indices(self):
        if self.c_indices is None:
            dimension = int(math.log(self.nb_classes, 3))
            table = [NNManager.compute_stones(c, dimension) for c in range(self.nb_classes)]
            self.c_indices = np.ndarray((dimension, 3, self.nb_classes // 3), dtype=np.uint8)
            for d in range(dimension):
                for stone, ci in colors.items():
                    self.c_indices[d, ci] = [c for c in range(self.nb_classes) if table[c][d] == stone]
        return self.c_indices
