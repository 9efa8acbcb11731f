"""Test-only stand-in for camkifu_amd.capi.Context that answers with the CPU oracle, so the
host-side finder logic can be exercised without a GPU.  Lives under tests/: the product never
routes through the oracle."""
import numpy as np

from oracle import oracle as ora


class OracleCtx:
    def __init__(self):
        self.weights = None
        self._mog = {}

    def board_detect(self, frames, hough_thresh=-1, cap=1024):
        frames = np.asarray(frames)
        if frames.ndim == 3:
            frames = frames[None]
        out = []
        for fr in frames:
            o = ora.board_lines(ora.canny(ora.median(fr, 15), 25, 75),
                                hough_thresh=None if hough_thresh < 0 else hough_thresh, cap=cap)
            st = {-1: 1, -2: 2}.get(o["status"], 0)
            out.append(dict(status=st, n_contours=o["n_contours"], n_lines=max(o["status"], 0),
                            biggest_area=o["biggest_area"], lines=o["lines"]))
        return out

    def warp_perspective(self, frame, M, dsize=380):
        return ora.warp_perspective(np.asarray(frame), M, (dsize, dsize))

    def mog2_create(self, h=380, w=380):
        k = len(self._mog)
        self._mog[k] = ora.MOG2(h, w, 3)
        return k

    def mog2_apply(self, handle, img, lr):
        return self._mog[handle].apply(img, lr)

    def cnn_set_weights(self, weights):
        self.weights = {k: np.asarray(v, np.float32) for k, v in weights.items()}

    def cnn_predict(self, goban, want_y=True):
        g = np.asarray(goban)
        if g.ndim == 3:
            g = g[None]
        ys, ls, cs = [], [], []
        for x in g:
            y = ora.cnn_predict_regions(self.weights, x)
            lab, cf = ora.decode_all(y)
            ys.append(y)
            ls.append(lab)
            cs.append(cf)
        return (np.stack(ys), np.stack(ls), np.stack(cs)) if want_y else (np.stack(ls), np.stack(cs))

    def stones_detect(self, frames, M):
        frames = np.asarray(frames)
        if frames.ndim == 3:
            frames = frames[None]
        gob = np.stack([self.warp_perspective(f, M) for f in frames])
        return self.cnn_predict(gob, want_y=False)
