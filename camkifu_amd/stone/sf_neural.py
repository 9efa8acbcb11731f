"""SfNeural for the MI355X: the CNN stones finder under its registration name (reference stone/sf_neural.py:26-244).

Per frame the classifier runs ONCE on the GPU for all 100 regions (NNCache -> ck_cnn_regions) and the foreground
mask is reduced to one count per intersection; what to do with those numbers -- the one-off full assessment after
the background frames, agitation targets, re-reading of calm regions with the colour-ratio veto, the three-check
lookback -- is decided by the library's ordered policy (ck_policy_run, camkifu_amd/csrc/ck_fold.cpp), the same code
the batch pipeline folds whole batches with.  This class only moves arrays in and requests out:
suggest() for a single new stone, bulk_update() otherwise."""
import numpy as np

from .. import capi
from ..core.exceptions import DeletedError
from ..golib_shim import gsize, E, B, W
from ..host import stones_finder_base
from . import nn_manager as nm
from .nn_cache import NNCache

WIN_NAME = 'Neural'
MIN_CONFIDENCE, TARGET_THRESH, TARGET_INCR, NB_LOOKBACK = 0.6, 15, 5, 3      # compiled into the policy; here for reference
_SYMBOL = (E, B, W)


def zone_counts_host(fg, zones):
    """foreground pixels per intersection zone, one vectorised reduction (SfNeural.is_agitated's 361 box sums,
    sf_neural.py:72-83, 178-180).  With the default grid the zones tile the mask: two reduceat calls; a learnt
    (irregular) grid falls back to an integral image."""
    on = (np.asarray(fg) != 0)
    x0, y0, x1, y1 = (zones[..., k] for k in range(4))
    regular = (x0[:, :1] == x0).all() and (y0[:1] == y0).all() and (x1[:-1, 0] == x0[1:, 0]).all() and (y1[0, :-1] == y0[0, 1:]).all()
    if regular:
        body = on[:x1[-1, 0], :y1[0, -1]]
        return np.add.reduceat(np.add.reduceat(body, x0[:, 0], 0, dtype=np.int32), y0[0], 1).astype(np.int32)
    ii = np.zeros((on.shape[0] + 1, on.shape[1] + 1), np.int64)
    ii[1:, 1:] = on.cumsum(0).cumsum(1)
    return (ii[x1, y1] - ii[x0, y1] - ii[x1, y0] + ii[x0, y0]).astype(np.int32)


class SfNeural(stones_finder_base()):
    def __init__(self, manager, ctx=None):
        try:
            super().__init__(manager, learn_bg=True, ctx=ctx)
        except TypeError:                                   # the host application's base takes no ctx
            super().__init__(manager, learn_bg=True)
            self.ctx = ctx if ctx is not None else capi.Context(getattr(manager, "device", 0))
        self.manager = nm.NNManager()
        self.indices = self.manager.class_indices()
        self.cache = None
        self.policy = capi.PolicyCore(getattr(self, "bg_init_frames", 50))
        self._weights_on_gpu = False

    # ---- the reference's attributes, read from the policy's state ------------------------------------
    @property
    def targets(self):
        return self.policy.state()["targets"]

    @targets.setter
    def targets(self, values):
        self.policy.set_targets(values)

    @property
    def heatmap(self):
        """(19, 19) object array: None, or (colour, energy, confidence) of the prediction under watch there"""
        st = self.policy.state()
        out = np.full((gsize, gsize), None, dtype=object)
        for r, c in np.argwhere(st["heat_color"] > 0):
            out[r, c] = (_SYMBOL[st["heat_color"][r, c]], int(st["heat_energy"][r, c]), float(st["heat_conf"][r, c]))
        return out

    @property
    def has_sampled(self):
        return self.policy.state()["has_sampled"]

    # ---- per frame ---------------------------------------------------------------------------------
    def _upload_net(self):
        if not self._weights_on_gpu:
            self.ctx.cnn_set_weights(self.manager.get_net(download=True))
            self._weights_on_gpu = True

    def _find(self, goban_img):
        f = self.total_f_processed
        self.cache = NNCache(self.manager, goban_img, self.ctx)
        self._upload_net()
        if f == 0 or f < self.bg_init_frames:
            return                                          # net loading frame / background sampling: nothing to decide
        labels, conf = self.cache.regions()
        fg = self.get_foreground()
        counts = None if fg is None else zone_counts_host(fg, self._zones())
        self.policy.run(f, labels[None], conf[None], None if counts is None else counts[None], self._board_codes, self._apply)

    def _zones(self):
        grid = getattr(self, "_posgrid", None)
        if hasattr(grid, "zones"):
            return grid.zones()
        return np.array([[self.getrect(r, c) for c in range(gsize)] for r in range(gsize)])      # host application's grid

    def _board_codes(self):
        stones = self.get_stones()
        return (stones == B).astype(np.uint8) + 2 * (stones == W).astype(np.uint8)

    def _apply(self, kind, moves, frame_index):
        named = [(_SYMBOL[color], r, c) for color, r, c in moves]
        if kind == capi.PolicyCore.SUGGEST:
            try:
                self.suggest(*named[0], doprint=False)
            except DeletedError as locked:
                print(locked)
        else:
            self.bulk_update(named)

    def _window_name(self):
        return WIN_NAME
