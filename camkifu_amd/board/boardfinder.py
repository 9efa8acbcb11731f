"""Standalone protocol base of a board finder, and the four-corner bookkeeping it exposes.

Contract (reference board/boardfinder.py:12-147): a subclass answers `_detect(frame) -> bool` and keeps
`self.corners` up to date; whenever `_detect` says yes the base derives `self.mtx`, the 3x3 transform
(K7, ck_get_perspective_transform) that maps the four corners onto the canonical square
`transform_dst`; the stones finder reads `mtx` and nothing else."""
import numpy as np

from .. import capi, cvconf
from ..core.video import VidProcessor


class GobanCorners:
    """Up to four corner candidates and, once they form a proper quadrilateral, their ordered hull.

    Points live in an (n, 2) integer array.  `submit` keeps candidates apart: while fewer than four are known a
    new one is only accepted further than min(h, w) / 5 from all others; afterwards it replaces its nearest
    neighbour.  `hull` is None until four points span a 4-vertex convex hull."""

    def __init__(self, points=()):
        self.frame = self.hull = None
        self._xy = np.zeros((0, 2), np.int64)
        for p in (points or ()):
            self._xy = np.vstack([self._xy, np.asarray(p, np.int64)[:2]])
        self._refresh()

    @property
    def _points(self):
        return [tuple(int(v) for v in p) for p in self._xy]

    def is_ready(self):
        return bool(self.hull)

    def submit(self, candidate):
        p = np.asarray(candidate, np.int64)[:2]
        if len(self._xy):
            d = np.sqrt(((self._xy - p) ** 2).sum(1).astype(np.float64))
            nearest = int(d.argmin())
        if len(self._xy) >= 4:
            self._xy[nearest] = p
        elif len(self._xy) == 0 or self.frame is None or d[nearest] > min(self.frame.shape[0], self.frame.shape[1]) / 5:
            self._xy = np.vstack([self._xy, p])
        self._refresh()

    def clear(self):
        self._xy, self.hull = np.zeros((0, 2), np.int64), None

    def paint(self, canvas):
        """display only: nothing to draw on in the headless build"""

    def _refresh(self):
        quad = capi.ordered_hull(self._xy) if len(self._xy) > 3 else ()
        self.hull = quad if len(quad) == 4 else None

    def __str__(self):
        return "Corners:%s" % (self._points,)


class BoardFinder(VidProcessor):
    def __init__(self, manager):
        VidProcessor.__init__(self, manager)
        side = cvconf.canonical_size
        self.transform_dst = np.array([(0, 0), (side, 0), (side, side), (0, side)], np.float32)
        self.corners, self.mtx = GobanCorners(), None

    def _detect(self, frame):
        raise NotImplementedError("a board finder implements _detect(frame) -> bool")

    def _doframe(self, frame):
        corners = self.corners
        corners.frame = frame
        if not self._detect(frame):
            return
        try:
            self.mtx = capi.get_perspective_transform(np.asarray(corners.hull, np.float32), self.transform_dst)
        except (capi.CkError, ValueError, TypeError) as why:      # no usable quadrilateral: the stones finder must wait
            print("board located but not usable: %s" % why)
            self.mtx = None

    def _show(self, img, name=None, loc=None, max_freq=2, **kw):
        super()._show(img, name=name, loc=loc if loc is not None else cvconf.bf_loc, max_frequ=max_freq)
