"""SfContours for the MI355X: the contour-analysis stones finder under its registration name (reference
stone/sf_contours.py:12-340).

`find_stones` -- hulls of the contours found in the image (median 13 + 7, Otsu, Canny) and in the foreground mask
(opening, Canny, size / orientation / fill / distance-transform filters) drawn into a mask, mean colours of each
intersection zone under or outside that mask, colour by comparison with the neighbouring zones -- is ONE library call
per goban image, or per batch of them (`ck_contour_stones`): the pixel work runs on the GPU, the per-contour geometry
and the raster-ordered colour decision in the library's host half (camkifu_amd/csrc/k_stonefind.hip,
ck_stonegeom.cpp).  `get_canny` is one call too (`ck_goban_canny`).  As in the reference the finder, run on its own,
looks and shows but submits nothing (`_find`, sf_contours.py:26-43): SfMeta is the caller that acts on the result."""
import numpy as np

from .. import capi
from ..golib_shim import gsize, E, B, W
from ..host import stones_finder_base

_SYMBOL = np.array([E, B, W], dtype=object)
_shared_ctx = []


def get_canny(img, ctx=None):
    """Smooth with two median blurs, then Canny with Otsu thresholds: an (h, w, 3) uint8 BGR image -> (h, w) uint8 edge
    map of 0 / 255 (same argument and result as the reference's static method)"""
    if ctx is None:
        if not _shared_ctx:
            _shared_ctx.append(capi.Context(0))
        ctx = _shared_ctx[0]
    return ctx.goban_canny(np.ascontiguousarray(img, np.uint8)[None])[0]


class SfContours(stones_finder_base()):
    get_canny = staticmethod(get_canny)

    def __init__(self, manager, ctx=None):
        try:
            super().__init__(manager, ctx=ctx)
        except TypeError:                                   # the host application's base takes no ctx
            super().__init__(manager)
            self.ctx = ctx if ctx is not None else capi.Context(getattr(manager, "device", 0))
        self.last_stones = None                             # what the reference draws into its window

    def _find(self, goban_img):
        if self.bg_init_frames < self.total_f_processed:
            self.last_stones = self.find_stones(goban_img)
        # (else: "BACKGROUND SAMPLING" on the reference's display)

    def _learn(self):
        pass

    def zone_table(self):
        """StonesFinder.getrect for the 361 intersections, (19, 19, 4) int32 -- from the grid the finder holds"""
        grid = getattr(self, "_posgrid", None)
        if hasattr(grid, "zones"):
            return np.ascontiguousarray(grid.zones(1.0), np.int32)
        return np.array([[self.getrect(r, c) for c in range(gsize)] for r in range(gsize)], np.int32)

    def find_stones(self, img, rs=0, re=gsize, cs=0, ce=gsize, canvas=None, **_):
        """-> (19, 19) object array of B / W / E, E outside rows [rs, re) x columns [cs, ce).  `canvas` (the reference's
        optional drawing surface) is accepted and left alone."""
        codes = self.ctx.contour_stones(np.ascontiguousarray(img, np.uint8), np.ascontiguousarray(self.get_foreground(), np.uint8),
                                        self.zone_table(), rs, re, cs, ce)
        return _SYMBOL[codes]

    def find_stones_batch(self, imgs, fgs, rs=0, re=gsize, cs=0, ce=gsize):
        """the same for n goban images and their foreground masks at once (host arrays or device tensors) ->
        (n, 19, 19) uint8 of 0 E / 1 B / 2 W"""
        return self.ctx.contour_stones(imgs, fgs, self.zone_table(), rs, re, cs, ce)

    def _window_name(self):
        return SfContours.__name__
