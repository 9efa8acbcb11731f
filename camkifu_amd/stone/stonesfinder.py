"""Standalone protocol base of a stones finder: the straightened goban image, the background model, the result
sink and the grid geometry a finder builds on (contract: reference stone/stonesfinder.py:95-176, 250-349,
412-450, 950-981).

    frame --K8 ck_warp_perspective(board_finder.mtx)--> goban_img 380x380x3
          --K9 ck_mog2_apply--> foreground mask --> _learn() (user corrections) --> _find(goban_img)

Pieces, each usable on its own (the batch pipeline's fold uses the sink without a finder):
    PosGrid        pixel position of the 361 intersections, and the zone (rectangle) around each as one table
    DeletionWatch  intersections the user emptied: averaged appearance, refusal of new stones until it changes
    StoneSink      (colour, row, col) decisions -> controller instructions "append" / "delete" / "bulk" / "auto_save"
    StonesFinder   the per-frame protocol: _doframe, ready_to_read, _find (abstract), suggest / remove / bulk_update,
                   getrect, get_foreground, corrected, is_empty, get_stones
Coordinates handed to suggest / remove / bulk_update are numpy (row, col)."""
import queue

import numpy as np

from .. import capi, cvconf
from ..core.exceptions import CorrectionWarning, DeletedError
from ..core.video import VidProcessor
from ..golib_shim import gsize, E, B, W, Move, NP_TYPE

correc_size = 10
_CODE = {E: 0, B: 1, W: 2}


class PosGrid:
    """`mtx[i, j]` = (x, y) pixel of intersection (i, j) in the canonical image: cell centres, 10 + 20 k at the
    default size.  `zones(cursor)` is the (19, 19, 4) table of rectangles (x0, y0, x1, y1) the reference computes one
    at a time in StonesFinder.getrect: bounded halfway (cursor 1.0) towards the diagonal neighbours, mirrored at the
    first line, stopping one pixel short of the image at the last."""

    def __init__(self, size):
        self.size = size
        half = size / gsize / 2
        line = (half * (gsize - 1 - np.arange(gsize)) + (size - half) * np.arange(gsize)) / (gsize - 1)
        self.mtx = np.stack(np.meshgrid(line, line, indexing="ij"), -1).astype(np.int16)
        self.adjust_vect, self.adjust_contribs = np.zeros(2, np.float32), 0
        self._zones = {}

    def zones(self, cursor=1.0):
        key = (float(cursor), self.mtx.tobytes())
        hit = self._zones.get(key)
        if hit is None:
            p = self.mtx.astype(np.int16)
            before = np.roll(p, (1, 1), (0, 1)).copy()                       # mtx[r - 1, c - 1] (index -1 wraps, as in Python)
            after = p[np.minimum(np.arange(gsize) + 1, gsize - 1)][:, np.minimum(np.arange(gsize) + 1, gsize - 1)].copy()
            before[0, :, 0] = -p[0, :, 0]
            before[:, 0, 1] = -p[:, 0, 1]
            after[-1, :, 0] = 2 * self.size - p[-1, :, 0] - 2
            after[:, -1, 1] = 2 * self.size - p[:, -1, 1] - 2
            k = 0.5 * cursor
            lo = np.maximum(0, (k * before + (1 - k) * p).astype(np.int64))  # astype truncates toward zero like int()
            hi = np.minimum(self.size, ((1 - k) * p + k * after).astype(np.int64))
            hit = np.concatenate([lo, hi], -1)
            self._zones = {key: hit}
        return hit

    def learn(self, grid, rate=0.2):
        """Drift correction (reference stonesfinder.py:1015-1043): `grid` is an updated copy of mtx (find_intersections'
        answer, made positive); the mean displacement of the intersections that moved is blended into `adjust_vect` at
        `rate`, and once more than 20 intersections have contributed the whole grid is shifted by it (truncated)."""
        if not 0 < rate <= 1:
            raise AssertionError("rate must lie in ]0, 1]")
        shift = np.asarray(grid, np.int16) - self.mtx
        if shift.min() < -200:
            raise ValueError("Provided grid seems too far from original, at least for one point.")
        movers = int(np.count_nonzero((shift != 0).any(-1)))
        if not movers:
            return
        step = shift.sum(axis=(0, 1), dtype=np.float32)
        step /= movers
        keep = 1.0 - rate
        self.adjust_vect *= keep                                   # float32 in place, as the reference's accumulator
        self.adjust_vect += step * rate
        self.adjust_contribs += movers
        if self.adjust_contribs <= 20:
            return
        whole = self.adjust_vect.astype(np.int16)                   # truncated toward zero
        print("Grid adjust vector : %s" % self.adjust_vect)
        self.mtx += whole
        self.adjust_vect.fill(0)
        self.adjust_contribs = 0


class DeletionWatch:
    """The user took a stone off (or moved it): that intersection is watched.  Over the next `samples` frames in which
    its zone is not entirely foreground the zone's pixels are averaged; from then on a stone is only accepted there
    once the zone differs from that average by 40 grey levels per pixel (all channels summed).
    reference: stonesfinder.py:178-245."""

    def __init__(self, shape, samples=50):
        self.samples = samples
        self.left = np.full((gsize, gsize), -1, np.int32)        # frames still to sample; -1 = not watched
        self.saved_bg = np.zeros(tuple(shape) + (3,), np.float32)

    def start(self, r, c):
        self.left[r, c] = self.samples

    def watched(self):
        return {(int(r), int(c)): int(self.left[r, c]) for r, c in np.argwhere(self.left >= 0)}

    def sample(self, goban_img, fg, zones):
        for r, c in np.argwhere(self.left > 0):
            x0, y0, x1, y1 = zones[r, c]
            # reference quirk kept: `np.sum(fg[zone] < 0.1 * area)` counts the pixels BELOW the bound, so the zone is
            # sampled as soon as one of its pixels is background
            if fg is None or (fg[x0:x1, y0:y1] < 0.1 * (x1 - x0) * (y1 - y0)).any():
                self.saved_bg[x0:x1, y0:y1] += goban_img[x0:x1, y0:y1] / self.samples
                self.left[r, c] -= 1

    def check(self, r, c, goban_img, zones):
        """raises DeletedError while (r, c) is locked; forgets the watch once the zone has changed enough"""
        state = self.left[r, c]
        if state < 0:
            return
        if state > 0:
            raise DeletedError(((r, c),), "The zone has been marked as deleted too recently.")
        x0, y0, x1, y1 = zones[r, c]
        delta = self.saved_bg[x0:x1, y0:y1] - goban_img[x0:x1, y0:y1]
        if np.sum(np.absolute(delta)) / (delta.shape[0] * delta.shape[1]) < 40:
            raise DeletedError(((r, c),), "The zone has not changed enough since last deletion.")
        print("intersection %s looks different from when the user emptied it: unlocked" % ((r, c),))
        self.left[r, c] = -1


class StoneSink:
    """What a finder decides -> what the controller is told.  `guard(r, c)` may raise DeletedError to veto a stone."""

    def __init__(self, controller_of, guard=None):
        self._controller_of, self._guard = controller_of, guard

    @property
    def controller(self):
        return self._controller_of()

    def is_empty(self, r, c):
        return self.controller.is_empty_blocking(c, r)

    def get_stones(self):
        return self.controller.get_stones()

    def board_codes(self):
        """the goban as uint8 (19, 19): 0 empty, 1 black, 2 white"""
        stones = self.get_stones()
        return (stones == B).astype(np.uint8) + 2 * (stones == W).astype(np.uint8)

    def _vet(self, r, c):
        if self._guard is not None:
            self._guard(r, c)

    def suggest(self, color, r, c, doprint=True):
        self._vet(r, c)
        stone = Move(NP_TYPE, (color, r, c))
        if doprint:
            print(stone)
        ctl = self.controller
        ctl.pipe("append", stone)
        ctl.pipe("auto_save")

    def remove(self, r, c):
        if self.is_empty(r, c):
            raise AssertionError("Can't remove stone from empty intersection.")
        gone = Move(NP_TYPE, ("", r, c))
        self.controller.pipe("delete", gone.x, gone.y)

    def bulk_update(self, tuples):
        """several changes as ONE controller instruction: E empties, B / W places (recolouring = empty, then place;
        an identical stone already there is skipped).  Vetoed places are left out and reported together afterwards.
        The goban is looked at as it WILL be after the changes already scheduled in this call, so a point named twice
        -- the regions of rows / columns 16-17 and 17-18 overlap and may both report it -- is sent once (or, if they
        disagree, settled in favour of the later one) instead of reaching the controller as two stones on one point."""
        ctl, batch, refused, scheduled = self.controller, [], [], {}

        def colour_at(r, c):
            if (r, c) in scheduled:
                return scheduled[(r, c)]
            return E if self.is_empty(r, c) else ctl.locate(c, r).color
        for color, r, c in tuples:
            now = colour_at(r, c)
            if color == E:
                if now != E:
                    batch.append(Move(NP_TYPE, (E, r, c)))
                    scheduled[(r, c)] = E
                continue
            if color not in (B, W) or now == color:
                continue
            if now != E:
                batch.append(Move(NP_TYPE, (E, r, c)))               # the clearing half goes out even if the stone is vetoed
                scheduled[(r, c)] = E
            try:
                self._vet(r, c)
            except DeletedError as veto:
                refused.append(veto)
                continue
            batch.append(Move(NP_TYPE, (color, r, c)))
            scheduled[(r, c)] = color
        if batch:
            ctl.pipe("bulk", batch)
            ctl.pipe("auto_save")
        if refused:
            raise DeletedError(refused, message="Bulk_update:warning: All non-conflicting locations have been sent.")


class StonesFinder(VidProcessor):
    def __init__(self, manager, learn_bg=True, ctx=None):
        VidProcessor.__init__(self, manager)
        self.ctx = ctx if ctx is not None else capi.Context(getattr(manager, "device", 0))
        side = cvconf.canonical_size
        self.canonical_shape, self.goban_img, self.intersections, self._fg = (side, side), None, None, None
        self._posgrid = PosGrid(side)
        self.watch = DeletionWatch(self.canonical_shape)
        self.corrections = queue.Queue(correc_size)
        self.sink = StoneSink(lambda: self.vmanager.controller, self._check_dels)
        if learn_bg:
            self.bg_model = self.ctx.mog2_create(side, side)
            video = getattr(manager, "current_video", None)
            still = isinstance(video, str) and video.lower().endswith((".png", ".jpg"))
            self.bg_init_frames = 0 if still else 50

    # ---- per frame -------------------------------------------------------------------------------
    def ready_to_read(self):
        bf = getattr(self.vmanager, "board_finder", None)
        return VidProcessor.ready_to_read(self) and getattr(bf, "mtx", None) is not None

    def _doframe(self, frame):
        bf, self.intersections = self.vmanager.board_finder, None
        mtx = None if bf is None else bf.mtx
        if mtx is None:
            return
        goban = self.goban_img = self.ctx.warp_perspective(frame, mtx, cvconf.canonical_size)   # K8
        for stage in (self._learn_bg, self._learn):
            stage()
        self._find(goban)

    def _find(self, goban_img):
        raise NotImplementedError("a stones finder implements _find(goban_img)")

    def _learn_bg(self):
        if getattr(self, "bg_model", None) is not None:
            rate = 0.01 if self.total_f_processed < self.bg_init_frames else 0.005
            self._fg = self.ctx.mog2_apply(self.bg_model, self.goban_img, rate)                  # K9

    def get_foreground(self):
        return self._fg

    # ---- empty intersections from grid lines ----------------------------------
    def find_intersections(self, img, canvas=None):
        """Which intersections show grid lines, hence no stone (reference stonesfinder.py:516-552): one library call
        (ck_find_intersections: grey / Otsu / Canny and one HoughLinesP wave per zone on the GPU, update_grid on the
        host) -> a copy of the grid with the positions where a line was found negated, and moved where a cross was.
        `canvas` (the reference's drawing surface) is accepted and left alone."""
        return self.ctx.find_intersections(np.ascontiguousarray(img, np.uint8), self._posgrid.mtx, self._posgrid.zones(1.0))

    def get_intersections(self, img, display=False):
        """cached per frame (reset in _doframe); the grid learns from every fresh answer (stonesfinder.py:554-576)"""
        cached = self.intersections
        if cached is None:
            cached = self.intersections = self.find_intersections(img)
            self._posgrid.learn(np.abs(cached))
        return cached

    # ---- consistency checks on a candidate result (stonesfinder.py:597-783; vectorised in stone/checks.py) ---------
    def check_against(self, stones, reference=None, rs=0, re=gsize, cs=0, ce=gsize, **kwargs):
        from . import checks
        return checks.check_against(stones, reference, rs, re, cs, ce)

    def check_lines(self, stones, img=None, rs=0, re=gsize, cs=0, ce=gsize, **kwargs):
        from . import checks
        return checks.check_lines(stones, self.get_intersections(img), rs, re, cs, ce)

    def check_thickness(self, stones, rs=0, re=gsize, cs=0, ce=gsize, **kwargs):
        from . import checks
        return checks.check_thickness(stones, rs, re, cs, ce)

    def check_flow(self, stones, rs=0, re=gsize, cs=0, ce=gsize, **kwargs):
        from . import checks
        return checks.check_flow(stones, self.sink.board_codes().reshape(gsize, gsize) == 0, rs, re, cs, ce)

    def first_line_lonelies(self, stones, reference=None, rs=0, re=gsize, cs=0, ce=gsize, **kwargs):
        from . import checks
        return checks.first_line_lonelies(stones, reference, rs, re, cs, ce)

    # ---- user corrections ------------------------------------------------------------------------
    def corrected(self, err_move, exp_move):
        pending = self.corrections
        try:
            pending.put_nowait((err_move, exp_move))
        except queue.Full:
            print("Corrections queue full (%s), ignoring %s -> %s" % (correc_size, err_move, exp_move))

    def _learn(self):
        """drain the corrections: a deletion (or the origin of a moved stone) goes under watch; anything else cannot be
        learnt from here and is reported (CorrectionWarning) after this frame's sampling"""
        leftover = []
        while True:
            try:
                wrong, right = self.corrections.get_nowait()
            except queue.Empty:
                break
            moved = wrong is not None and right is not None and (wrong.x, wrong.y) != (right.x, right.y)
            if right is None or moved:
                self.watch.start(wrong.y, wrong.x)                 # Move.x is the column, Move.y the row
            else:
                leftover.append((wrong, right))
        self.watch.sample(self.goban_img, self.get_foreground(), self._posgrid.zones())
        if leftover:
            raise CorrectionWarning(leftover, message="Unhandled corrections")

    def _check_dels(self, r, c):
        self.watch.check(r, c, self.goban_img, self._posgrid.zones())

    @property
    def deleted(self):
        return self.watch.watched()

    @property
    def saved_bg(self):
        return self.watch.saved_bg

    @property
    def nb_del_samples(self):
        return self.watch.samples

    @nb_del_samples.setter
    def nb_del_samples(self, n):
        self.watch.samples = n

    # ---- results ---------------------------------------------------------------------------------
    def suggest(self, color, r, c, doprint=True):
        self.sink.suggest(color, r, c, doprint)

    def remove(self, r, c):
        self.sink.remove(r, c)

    def bulk_update(self, tuples):
        self.sink.bulk_update(tuples)

    def is_empty(self, r, c):
        return self.sink.is_empty(r, c)

    def get_stones(self):
        return self.sink.get_stones()

    # ---- geometry --------------------------------------------------------------------------------
    def getrect(self, r, c, cursor=1.0):
        if not isinstance(cursor, float):
            raise TypeError("cursor must be a float in ]0, 2[")
        return tuple(int(v) for v in self._posgrid.zones(cursor)[r, c])

    def _window_name(self):
        return "camkifu.stone.stonesfinder.StonesFinder"


def update_grid(lines, box, result_slot):
    """reference stonesfinder.py:888-947, as a module function like there: the library's host routine"""
    capi.update_grid(lines, box, result_slot)
