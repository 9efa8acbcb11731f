"""BoardFinderAuto for the MI355X: same registration name and hooks as the reference's automatic board finder
(board/bf_auto.py:11-219), every piece of arithmetic behind the C-ABI.

    per frame, stateless (GPU, ONE call -- ck_board_detect, K1..K6):
        medianBlur 15 -> Canny 25/75 -> findContours(RETR_EXTERNAL) -> 3 largest minAreaRect -> drawContours
        -> HoughLines(1, pi/180, min(h, w) / 5)                                   bf_auto.py:72-84, 105-133
    per frame, ordered (host, ck_boardfold_step):
        4-frame line accumulation -> intersections -> grouping -> 4 corners        bf_auto.py:85-102, 143-217

`_detect(frame, record=None)` accepts a precomputed stateless result so that the batch pipeline's fold
(pipeline.BoardFold) runs the very same method over gathered records."""
import time

from .. import capi
from ..host import board_finder_base


class BoardFinderAuto(board_finder_base()):
    HOUGH_CAP = 1024                 # lines fetched per frame; more than that is reported, not truncated silently

    def __init__(self, manager, ctx=None):
        super().__init__(manager)
        self._ctx = ctx                              # created on first use: a fold-only instance never touches the GPU
        self.core = capi.BoardFoldCore()
        self.auto_refresh, self.last_positive = 10, float("-inf")      # seconds without looking after a hit (live input)
        self.last_stats = None                       # (clusters, intersections) of the latest grouping round

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = capi.Context(getattr(self.vmanager, "device", 0))
        return self._ctx

    def _doframe(self, frame):
        idle = time.monotonic() - self.last_positive
        if idle > self.auto_refresh:
            return super()._doframe(frame)
        self.metadata["Last detection {}s ago"] = int(idle)
        self._show(frame)

    def detect_core(self, frame):
        """the stateless half on the GPU -> dict(status, n_contours, n_lines, biggest_area, lines)"""
        return self.ctx.board_detect(frame, cap=self.HOUGH_CAP)[0]

    def _detect(self, frame, record=None):
        h, w = frame.shape[0], frame.shape[1]
        rec = self.detect_core(frame) if record is None else record
        if rec["n_lines"] > len(rec["lines"]):
            raise capi.CkError("%d Hough lines found, %d fetched" % (rec["n_lines"], len(rec["lines"])))
        # cv2.HoughLines answers None when no cell reaches the threshold and the reference then raises on the
        # iteration (bf_auto.py:135); here such a frame simply contributes no line.
        found, update, centers, stats = self.core.step(h, w, rec["status"], rec["lines"], self.total_f_processed,
                                                       self.corners.hull)
        if stats is not None:
            self.last_stats = stats
            self.metadata["Clusters : {}"].append(stats[0])
            self.metadata["Line intersections: {}"] = stats[1]
        if update:
            corners = self.corners
            corners.clear()
            for p in centers:
                corners.submit(p)
        if self.total_f_processed % 4 == 0:
            self.metadata["Board  : {}"] = ("searching", "found")[bool(found)]
            self._show(frame)
        if found:
            self.last_positive = time.monotonic()
        return found

    def _window_name(self):
        return "Board Finder Auto"
