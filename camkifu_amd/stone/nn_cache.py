"""NNCache: one frame's classifier answers, asked for once.

The reference memoises one batch-1 Keras predict per 40x40 patch (stone/nn_cache.py:8-58).  Here the first
question about a frame runs all 100 patches in ONE GPU call (ck_cnn_regions, K10..K12) that returns what every
query of the class boils down to -- per region its argmax label and max(y)/sum(y) -- and the three
predict_* methods are table lookups on that."""
import numpy as np

from ..golib_shim import gsize
from . import nn_manager as nm


class NNCache:
    def __init__(self, nn_manager, img, ctx):
        if tuple(img.shape[:2]) != tuple(nn_manager.canonical_shape):
            raise ValueError("goban image must be %s, got %s" % (nn_manager.canonical_shape, img.shape))
        self.manager, self.img, self.ctx = nn_manager, img, ctx
        self._regions = None

    def regions(self):
        """-> (labels uint8 (10, 10), confidences float64 (10, 10)) of this frame"""
        if self._regions is None:
            lab, conf = self.ctx.cnn_regions(self.img)
            self._regions = np.asarray(lab)[0], np.asarray(conf)[0]
        return self._regions

    def predict_4_stones(self, ri, cj):
        lab, conf = self.regions()
        return nm.SYMBOLS[nm.DIGITS[lab[ri, cj]]].reshape(nm.STEP, nm.STEP), float(conf[ri, cj])

    def predict_stone(self, row, col):
        # NB: the region is (row // 2, col // 2) and the entry 2 * (row % 2) + col % 2 of its four stones, which on row / column
        # 18 is the region's FIRST row / column -- the reference's arithmetic (nn_cache.py:16-23), kept as it is
        lab, conf = self.regions()
        i, j = row // nm.STEP, col // nm.STEP
        return nm.SYMBOLS[nm.DIGITS[lab[i, j], nm.STEP * (row % nm.STEP) + col % nm.STEP]], float(conf[i, j])

    def predict_all_stones(self):
        """(19, 19, 2) object array [colour, confidence]; regions are laid down in raster order, so on row / column
        17 the last region's answer stands"""
        lab, conf = self.regions()
        out = np.empty((gsize, gsize, 2), dtype=object)
        for i, rs in enumerate(nm.REGION_START):
            for j, cs in enumerate(nm.REGION_START):
                out[rs:rs + nm.STEP, cs:cs + nm.STEP, 0] = nm.SYMBOLS[nm.DIGITS[lab[i, j]]].reshape(nm.STEP, nm.STEP)
                out[rs:rs + nm.STEP, cs:cs + nm.STEP, 1] = float(conf[i, j])
        return out
