"""NNCache: per-frame memo of the classifier's outputs (mirror of the reference's
stone/nn_cache.py:8-58).  The reference runs one batch-1 Keras predict per 40x40 patch; here the
first request of a frame runs all 100 patches in ONE GPU call (K10..K12) and the rest are reads."""
import numpy as np

from ..golib_shim import gsize


class NNCache:
    def __init__(self, nn_manager, img, ctx):
        assert img.shape[0:2] == nn_manager.canonical_shape
        self.manager = nn_manager
        self.img = img
        self.ctx = ctx
        self._y = None                       # (100, 81) softmax outputs, filled on first use

    def _all_y(self):
        if self._y is None:
            y, _, _ = self.ctx.cnn_predict(self.img)
            self._y = y[0]
        return self._y

    def predict_y(self, i, j):
        return self._all_y()[i * self.manager.split + j]

    def predict_all_y(self):
        return self._all_y().reshape(self.manager.split, self.manager.split, -1)

    @staticmethod
    def _confidence(y):
        tot = 0.0                            # python sum() over float32 scalars: float64, in order
        for v in y:
            tot = tot + float(v)
        return float(max(y)) / tot

    def predict_4_stones(self, i, j):
        y = self.predict_y(i, j)
        rs, re, cs, ce = self.manager._subregion(i, j)
        stones = self.manager.compute_stones(int(np.argmax(y))).reshape((re - rs, ce - cs))
        return stones, self._confidence(y)

    def predict_stone(self, r, c):
        i, j = self.manager.get_region_indices(r, c)
        y = self.predict_y(i, j)
        stones = self.manager.compute_stones(int(np.argmax(y)))
        step = self.manager.step
        return stones[step * (r % step) + c % step], self._confidence(y)

    def predict_all_stones(self):
        stones = np.ndarray((gsize, gsize, 2), dtype=object)
        for i in range(self.manager.split):
            for j in range(self.manager.split):
                rs, re, cs, ce = self.manager._subregion(i, j)
                square, confidence = self.predict_4_stones(i, j)
                stones[rs:re, cs:ce, 0] = square
                stones[rs:re, cs:ce, 1] = confidence
        return stones
