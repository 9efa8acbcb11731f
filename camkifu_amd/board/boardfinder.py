"""BoardFinder base class and GobanCorners (mirror of the reference's board/boardfinder.py:12-147).
Subclasses implement `_detect(frame) -> bool` and fill `self.corners`; on success the base
class computes `self.mtx`, the 3x3 perspective transform the stones finder reads."""
import sys
import traceback

import numpy as np

from .. import capi, cvconf
from ..core import imgutil
from ..core.video import VidProcessor


class BoardFinder(VidProcessor):
    def __init__(self, vmanager):
        super().__init__(vmanager)
        self.corners = GobanCorners()
        size = cvconf.canonical_size
        self.transform_dst = np.array([(0, 0), (size, 0), (size, size), (0, size)], dtype=np.float32)
        self.mtx = None

    def _doframe(self, frame):
        self.corners.frame = frame
        if self._detect(frame):
            source = np.array(self.corners.hull, dtype=np.float32)
            try:
                self.mtx = capi.get_perspective_transform(source, self.transform_dst)   # K7
            except capi.CkError:
                self.mtx = None          # the stones finder must stop
                traceback.print_exc()

    def _detect(self, frame):
        raise NotImplementedError("Abstract method meant to be extended")

    def _show(self, img, name=None, frame=True, latency=True, thread=False, loc=None, max_freq=2):
        super()._show(img, name, frame, latency, thread, loc=loc or cvconf.bf_loc, max_frequ=max_freq)


class GobanCorners:
    """The corner points found so far and their 4-vertex convex hull (None until complete)."""

    def __init__(self, points=None):
        self.hull = None
        self.frame = None
        self._points = list(points) if points is not None else []
        self._check_hull()

    def is_ready(self):
        return self.hull is not None

    def submit(self, point):
        """append while fewer than 4 points are known (rejecting a point that sits too close to
        another one); afterwards replace the closest point"""
        closest_d, closest_i = sys.maxsize, None
        for i, pt in enumerate(self._points):
            d = imgutil.norm(pt, point)
            if d < closest_d:
                closest_d, closest_i = d, i
        if len(self._points) < 4:
            if closest_i is None or self.frame is None or min(*self.frame.shape[0:2]) / 5 < closest_d:
                self._points.append(point)
        else:
            self._points[closest_i] = point
        self._check_hull()

    def clear(self):
        self._points = []
        self._check_hull()

    def paint(self, img):
        pass                              # drawing is display-only and out of scope

    def _check_hull(self):
        self.hull = None
        if 3 < len(self._points):
            hull = imgutil.get_ordered_hull(self._points)
            if len(hull) == 4:
                self.hull = hull

    def __str__(self):
        return "Corners:" + str(self._points)
