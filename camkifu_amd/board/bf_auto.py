"""BoardFinderAuto on the MI355X path (mirror of the reference's board/bf_auto.py:11-219).

`_detect` keeps the reference's control flow; the image chain
    cv2.medianBlur(frame, 15) -> cv2.Canny(., 25, 75) -> cv2.findContours(RETR_EXTERNAL)
    -> minAreaRect sort, 3 biggest -> drawContours -> cv2.HoughLines(1, pi/180, min(h,w)/5)
is ONE call through the C-ABI (ck_board_detect, kernels K1..K6); what comes back is what the
reference's Python then consumes: the contour count, the biggest minAreaRect area and the
(rho, theta) list in OpenCV's order.  The temporal logic (4-frame line accumulation,
intersection grouping, cluster merging, corner update) is the reference's and stays on the host.
"""
import bisect
import math
import time

from .. import capi
from ..core import imgutil
from .boardfinder import BoardFinder


class BoardFinderAuto(BoardFinder):
    def __init__(self, vmanager, ctx=None):
        super().__init__(vmanager)
        self.ctx = ctx if ctx is not None else capi.Context(getattr(vmanager, "device", 0))
        self.lines_accu = []
        self.groups_accu = []
        self.auto_refresh = 10          # seconds to sleep after a positive detection
        self.last_positive = -1.0
        self.hough_cap = 1024

    def _doframe(self, frame):
        elapsed = time.time() - self.last_positive
        if self.auto_refresh < elapsed:
            super()._doframe(frame)
        else:
            self.metadata["Last detection {}s ago"] = int(elapsed)
            self._show(frame)

    # the stateless per-frame core, also used by the batch pipeline with precomputed results
    def detect_core(self, frame):
        return self.ctx.board_detect(frame, cap=self.hough_cap)[0]

    def _detect(self, frame, core=None):
        length_ref = min(frame.shape[0], frame.shape[1])
        res = core if core is not None else self.detect_core(frame)
        if res["status"] == capi.CK_BOARD_NO_CONTOUR:
            return False
        found = False
        if res["status"] == capi.CK_BOARD_LINES:             # frame_area / 3 < biggest.area
            if res["n_lines"] > len(res["lines"]):
                raise capi.CkError("Hough line capacity exceeded: %d lines" % res["n_lines"])
            # NB: cv2.HoughLines returns None when nothing passes the threshold and the reference
            # would raise on it (bf_auto.py:135); here an empty result just adds no segment.
            segments = [imgutil.segment_from_hough(l, frame.shape[0:2]) for l in res["lines"]]
            self.lines_accu.extend(segments)
            if not self.total_f_processed % 4:
                self.group_intersections(frame.shape)
                while 4 < len(self.groups_accu):
                    before = len(self.groups_accu)
                    imgutil.connect_clusters(self.groups_accu, (length_ref / 50) ** 2)
                    if len(self.groups_accu) == before:
                        break
                found = self.updt_corners(length_ref)
        if not self.total_f_processed % 4:
            self.metadata["Board  : {}"] = "found" if found else "searching"
            self._show(frame)
        if found:
            self.last_positive = time.time()
        return found

    def group_intersections(self, shape):
        """pairwise intersections of sufficiently non-parallel accumulated lines, greedily grouped
        (x-only proximity test -- reference quirk, bf_auto.py:161: the squared "distance" is
        (dx)^2 + (dx)^2).  Same groups in the same order as the reference's loops; the membership test
        `any(2 dx^2 < thresh for p1 in g)` is answered from the sorted x values of the group (the
        nearest x decides), which keeps the every-4th-frame cost flat when hundreds of near-duplicate
        intersections pile up while the board is not found."""
        length_ref = min(shape[0], shape[1])
        margin = -length_ref / 15
        thresh = (length_ref / 80) ** 2
        ordered = sorted(self.lines_accu, key=lambda s: s.theta)
        xs_of = {id(g): sorted(p[0] for p in g) for g in self.groups_accu}
        for s1 in ordered:
            for s2 in reversed(ordered):
                if not (math.pi / 3 < s1.line_angle(s2)):
                    break                # remaining s2 are even more parallel to s1
                p0 = s1.intersection(s2)
                if not imgutil.within_margin(p0, (0, 0, shape[1], shape[0]), margin):
                    continue
                x = p0[0]
                for g in self.groups_accu:
                    xs = xs_of[id(g)]
                    k = bisect.bisect_left(xs, x)
                    near = False
                    if k < len(xs):
                        d = xs[k] - x
                        near = d * d + d * d < thresh
                    if not near and k > 0:
                        d = x - xs[k - 1]
                        near = d * d + d * d < thresh
                    if near:
                        g.append(p0)
                        xs.insert(k, x)
                        break
                else:
                    g = [p0]
                    self.groups_accu.append(g)
                    xs_of[id(g)] = [x]

    def updt_corners(self, length_ref):
        found = False
        if len(self.groups_accu) == 4:
            centers = []
            for group in self.groups_accu:
                sx = sum(pt[0] for pt in group)
                sy = sum(pt[1] for pt in group)
                centers.append((int(sx / len(group)), int(sy / len(group))))
            centers = imgutil.get_ordered_hull(centers)
            found = all(not (imgutil.norm(centers[i - 1], centers[i]) < length_ref / 3)
                        for i in range(len(centers)))
            update = self.corners.hull is None
            if found and not update:
                # both hulls are spatially sorted: compare corner by corner (reference indexes
                # the new hull over range(4); a degenerate hull raises there and here alike)
                update = any(5 < imgutil.norm(centers[i], self.corners.hull[i]) for i in range(4))
            if update:
                self.corners.clear()
                for pt in centers:
                    self.corners.submit(pt)
        self.metadata["Clusters : {}"].append(len(self.groups_accu))
        self.metadata["Line intersections: {}"] = sum(len(g) for g in self.groups_accu)
        self.lines_accu.clear()
        self.groups_accu.clear()
        return found

    def _window_name(self):
        return "Board Finder Auto"
