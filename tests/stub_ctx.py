"""Test-only stand-in for camkifu_amd.capi.Context that answers with the CPU oracle, so the
host-side finder logic can be exercised without a GPU.  Lives under tests/: the product never
routes through the oracle."""
import numpy as np

from oracle import oracle as ora


class OracleCtx:
    def __init__(self):
        self.weights = None
        self._mog = {}

    @staticmethod
    def _batch(a, nd):
        a = np.asarray(a)
        return a[None] if a.ndim == nd else a

    def board_detect(self, frames, hough_thresh=-1, cap=1024, raw=False):
        out = []
        for fr in self._batch(frames, 3):
            o = ora.board_lines(ora.canny(ora.median(fr, 15), 25, 75),
                                hough_thresh=None if hough_thresh < 0 else hough_thresh, cap=cap)
            st = {-1: 1, -2: 2}.get(o["status"], 0)
            out.append(dict(status=st, n_contours=o["n_contours"], n_lines=max(o["status"], 0),
                            biggest_area=o["biggest_area"], lines=o["lines"]))
        if raw:
            from camkifu_amd import capi
            res = np.zeros(len(out), capi.BOARD_DTYPE)
            lines = np.zeros((len(out), cap, 2), np.float32)
            for f, b in enumerate(out):
                res[f] = (b["status"], b["n_contours"], b["n_lines"], 0, b["biggest_area"])
                k = min(b["n_lines"], cap)
                lines[f, :k] = b["lines"][:k]
            return res, lines
        return out

    def warp_perspective(self, frame, M, dsize=380, out=None):
        frame = np.asarray(frame)
        if frame.ndim == 4:
            res = np.stack([ora.warp_perspective(f, M, (dsize, dsize)) for f in frame])
        else:
            res = ora.warp_perspective(frame, M, (dsize, dsize))
        if out is not None:
            out[...] = res
            return out
        return res

    def mog2_create(self, h=380, w=380):
        k = len(self._mog)
        self._mog[k] = ora.MOG2(h, w, 3)
        return k

    def mog2_apply(self, handle, img, lr):
        return self._mog[handle].apply(img, lr)

    def mog2_band_run(self, handle, band, learning_rates, last_band):
        band = np.asarray(band)
        out = np.zeros((len(band), (band.shape[1] + 19) // 20, 19), np.int32)
        for f, img in enumerate(band):
            fg = self._mog[handle].apply(np.ascontiguousarray(img), float(learning_rates[f]))
            padded = np.zeros((out.shape[1] * 20, 380), np.uint8)
            padded[:fg.shape[0]] = fg
            padded[:, 379] = 0
            if last_band:
                padded[fg.shape[0] - 1] = 0
            out[f] = (padded != 0).reshape(out.shape[1], 20, 19, 20).sum((1, 3))
        return out

    def zone_counts(self, mask):
        mask = np.asarray(mask)
        if mask.ndim == 2:
            return ora.zone_counts(mask)
        return np.stack([ora.zone_counts(m) for m in mask])

    def contour_stones(self, goban, fg, rects, rs=0, re=19, cs=0, ce=19, want_all=False):
        from oracle import ora_stones
        assert np.array_equal(np.asarray(rects).reshape(19, 19, 4), [[ora.sf_getrect(r, c) for c in range(19)] for r in range(19)])
        goban, fg = np.asarray(goban), np.asarray(fg)
        if goban.ndim == 3:
            return ora_stones.find_stones(goban, fg, rs, re, cs, ce, want_all)[:3] if want_all else ora_stones.find_stones(goban, fg, rs, re, cs, ce)
        res = [ora_stones.find_stones(g, m, rs, re, cs, ce, want_all) for g, m in zip(goban, fg)]
        return tuple(np.stack([r[k] for r in res]) for k in range(3)) if want_all else np.stack(res)

    def find_intersections(self, goban, mtx, rects, want_lines=False):
        from oracle import ora_grid
        goban = np.asarray(goban)
        if goban.ndim == 3:
            return ora_grid.find_intersections(goban, mtx, rects, want_lines)
        res = [ora_grid.find_intersections(g, mtx, rects, want_lines) for g in goban]
        return tuple(list(x) for x in zip(*res)) if want_lines else np.stack(res)

    def cnn_set_weights(self, weights):
        self.weights = {k: np.asarray(v, np.float32) for k, v in weights.items()}

    def cnn_predict(self, goban, want_y=True):
        ys, ls, cs = [], [], []
        for x in self._batch(goban, 3):
            y = ora.cnn_predict_regions(self.weights, x)
            lab, cf = ora.decode_all(y)
            ys.append(y)
            ls.append(lab)
            cs.append(cf)
        return (np.stack(ys), np.stack(ls), np.stack(cs)) if want_y else (np.stack(ls), np.stack(cs))

    def cnn_regions(self, goban):
        labs, confs = [], []
        for x in self._batch(goban, 3):
            lab, cf = ora.decode_regions(ora.cnn_predict_regions(self.weights, x))
            labs.append(lab.reshape(10, 10))
            confs.append(cf.reshape(10, 10))
        return np.stack(labs), np.stack(confs)

    def stones_detect(self, frames, M):
        gob = np.stack([self.warp_perspective(f, M) for f in self._batch(frames, 3)])
        return self.cnn_predict(gob, want_y=False)

    def stones_run(self, frames, M, mog2=None, learning_rates=None, want_grid=False):
        gob = np.stack([self.warp_perspective(f, M) for f in self._batch(frames, 3)])
        rl, rc = self.cnn_regions(gob)
        fg = None
        if mog2 is not None:
            fg = np.stack([ora.zone_counts(self._mog[mog2].apply(g, float(lr))) for g, lr in zip(gob, learning_rates)])
        out = dict(region_label=rl, region_conf=rc, fgcount=fg, labels=None, conf=None)
        if want_grid:
            out["labels"], out["conf"] = self.cnn_predict(gob, want_y=False)
        return out
