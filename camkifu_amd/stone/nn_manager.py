"""NNManager: region geometry, the base-3 label codec and the classifier's weights -- the inference side of the
reference's stone/nn_manager.py (29-131, 216-298, 360-382) under the same method names, built on small lookup
tables instead of per-call loops.  The Keras model is a dictionary of twelve float32 arrays handed to the HIP
library (K10..K12, ck_cnn_set_weights); training and the labelling GUI are out of scope.

Tables (gsize 19, 10 x 10 regions of 2 x 2 intersections, the last region pulled back onto rows 17-18):
    REGION_START[i]  first row (or column) of region i            0, 2, ..., 16, 17
    DIGITS[label]    the four base-3 digits of a label, least significant first = intersections
                     (0,0) (0,1) (1,0) (1,1) of the region         nn_manager.py:236-254
    PATCH_ORIGIN[i]  first pixel of region i's 40-pixel window     0, 40, ..., 320, 340
"""
import os
from threading import RLock

import numpy as np

from .. import cvconf
from ..golib_shim import gsize, E, B, W

SPLIT = 10
STEP = (gsize + 1) // SPLIT
NB_CLASSES = 3 ** (STEP * STEP)
REGION_START = np.minimum(np.arange(SPLIT) * STEP, gsize - STEP)
DIGITS = ((np.arange(NB_CLASSES)[:, None] // 3 ** np.arange(STEP * STEP)[None, :]) % 3).astype(np.uint8)
SYMBOLS = np.array([E, B, W], dtype=object)
CODE = {E: 0, B: 1, W: 2}
CELL_PX = cvconf.canonical_size // gsize
PATCH_ORIGIN = REGION_START * CELL_PX
PATCH_SIDE = STEP * CELL_PX

# The trained model file.  The reference downloads the author's keras.h5 into its training directory
# (stone/nn_manager.py:22, 65-90); that file cannot be fetched here, so the package ships a classifier trained on the
# synthetic renderer (tools/train_cnn.py) in the same Keras-1 HDF5 layout.  $CAMKIFU_KERAS_MODEL overrides the path.
PACKAGED_MODEL = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "keras.h5")
KERAS_MODEL_FILE = os.environ.get("CAMKIFU_KERAS_MODEL") or PACKAGED_MODEL
GOLDEN_WEIGHTS = KERAS_MODEL_FILE


class ModelMissing(FileNotFoundError):
    pass


def _seeded_weights():
    from .. import synth
    return synth.cnn_weights()


class NNManager:
    _network, _guard = None, RLock()      # the weights, loaded once per process like the reference's lazily built model

    def __init__(self):
        self.canonical_shape = (cvconf.canonical_size,) * 2
        self.split, self.step, self.nb_classes = SPLIT, STEP, NB_CLASSES
        self.r_width = self.c_width = PATCH_SIDE
        self.c_indices = self._class_table = None

    # ---- model ---------------------------------------------------------------------------------
    @classmethod
    def get_net(cls, download=False):
        with cls._guard:
            if cls._network is None:
                cls._network = cls.init_net()
            return cls._network

    def init_net(download=False, allow_random=False):
        """the twelve weight arrays (Keras-1 'tf' layout) from KERAS_MODEL_FILE.  A missing file is an error -- a
        classifier with random weights reads garbage -- unless seeded random weights are asked for explicitly
        (numerics tests use them)."""
        if os.path.isfile(KERAS_MODEL_FILE):
            return NNManager.load_model(KERAS_MODEL_FILE)
        if allow_random:
            return _seeded_weights()
        raise ModelMissing("stone classifier model not found: %s (set CAMKIFU_KERAS_MODEL)" % KERAS_MODEL_FILE)

    def load_model(path):
        from . import keras1
        return keras1.load_model(path)

    def create_net():
        """the architecture of nn_manager.py:277-298 with seeded He-normal weights (untrained)"""
        return _seeded_weights()

    init_net, load_model, create_net = staticmethod(init_net), staticmethod(load_model), staticmethod(create_net)

    # ---- geometry --------------------------------------------------------------------------------
    def _subregion(self, ri, cj):
        """(row start, row end, col start, col end) of region (ri, cj), ends exclusive"""
        r0, c0 = int(REGION_START[ri]), int(REGION_START[cj])
        return r0, r0 + STEP, c0, c0 + STEP

    def get_region_indices(self, row, col):
        return row // STEP, col // STEP

    def getrect(self, row, col, re=0, ce=0):
        """pixel box (x0, y0, x1, y1) of intersections row..re, col..ce (inclusive; 0 = just row / col), canonical image"""
        last_r, last_c = (re if re > 0 else row), (ce if ce > 0 else col)
        return row * CELL_PX, col * CELL_PX, (last_r + 1) * CELL_PX, (last_c + 1) * CELL_PX

    def _get_rect_nn(self, r0, r1, c0, c1):
        """(x0, x1, y0, y1) of the classifier's window for a block of intersections: always PATCH_SIDE wide,
        anchored at the block's far edge"""
        x1, y1 = r1 * CELL_PX, c1 * CELL_PX
        return x1 - PATCH_SIDE, x1, y1 - PATCH_SIDE, y1

    def _get_x(self, ri, cj, image):
        a, b = int(PATCH_ORIGIN[ri]), int(PATCH_ORIGIN[cj])
        return image[a:a + PATCH_SIDE, b:b + PATCH_SIDE]

    # ---- codec -----------------------------------------------------------------------------------
    def compute_label(r0, r1, c0, c1, stones):
        block = np.asarray(stones)[r0:r1, c0:c1].reshape(-1)
        return int(sum(CODE[s] * 3 ** k for k, s in enumerate(block)))

    def compute_stones(label, dimension=STEP * STEP):
        if dimension == STEP * STEP:
            return SYMBOLS[DIGITS[int(label)]]
        return SYMBOLS[(int(label) // 3 ** np.arange(dimension)) % 3]

    compute_label, compute_stones = staticmethod(compute_label), staticmethod(compute_stones)

    def class_indices(self):
        """[intersection k, colour, :] = the labels that put that colour on that intersection (27 each)"""
        table = self._class_table
        if table is None:
            per_colour = [[np.flatnonzero(DIGITS[:, k] == col) for col in range(3)] for k in range(STEP * STEP)]
            table = self.c_indices = self._class_table = np.array(per_colour, np.uint8)
        return table
