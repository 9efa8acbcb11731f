"""StonesFinder base class (mirror of the hot-path part of the reference's
stone/stonesfinder.py:95-176, 250-349, 412-450, 950-981).

_doframe = K8 warpPerspective(frame, board_finder.mtx, (380, 380)) -> K9 MOG2 background model
-> _learn -> _find(goban_img); both image stages run on the GPU through the C-ABI.  Results go
to the controller through suggest / remove / bulk_update in numpy (row, col) coordinates.
Line-based emptiness checks and grid learning are "next" rows."""
import queue

import numpy as np

from .. import capi, cvconf
from ..core.exceptions import CorrectionWarning, DeletedError
from ..core.video import VidProcessor
from ..golib_shim import gsize, E, B, W, Move, NP_TYPE

correc_size = 10


class StonesFinder(VidProcessor):
    def __init__(self, vmanager, learn_bg=True, ctx=None):
        super().__init__(vmanager)
        self.ctx = ctx if ctx is not None else capi.Context(getattr(vmanager, "device", 0))
        self.goban_img = None
        self.canonical_shape = (cvconf.canonical_size, cvconf.canonical_size)
        self._posgrid = PosGrid(cvconf.canonical_size)
        self.intersections = None
        self._fg = None
        if learn_bg:
            self.bg_model = self.ctx.mog2_create(*self.canonical_shape)
            video = getattr(self.vmanager, "current_video", None)
            is_img = isinstance(video, str) and video.lower().endswith((".png", ".jpg"))
            self.bg_init_frames = 0 if is_img else 50
        # (quite primal) "learning" attributes, see _learn()
        self.corrections = queue.Queue(correc_size)
        self.saved_bg = np.zeros(self.canonical_shape + (3,), dtype=np.float32)
        self.deleted = {}
        self.nb_del_samples = 50

    def _doframe(self, frame):
        self.intersections = None
        transform = None
        if self.vmanager.board_finder is not None:
            transform = self.vmanager.board_finder.mtx
        if transform is not None:
            self.goban_img = self.ctx.warp_perspective(frame, transform, cvconf.canonical_size)   # K8
            self._learn_bg()
            self._learn()
            self._find(self.goban_img)

    def ready_to_read(self):
        try:
            return super().ready_to_read() and self.vmanager.board_finder.mtx is not None
        except AttributeError:
            return False

    def _find(self, goban_img):
        raise NotImplementedError("Abstract method meant to be extended")

    def _learn_bg(self):
        if hasattr(self, "bg_model"):
            learning = 0.01 if self.total_f_processed < self.bg_init_frames else 0.005
            self._fg = self.ctx.mog2_apply(self.bg_model, self.goban_img, learning)          # K9

    def _learn(self):
        """User corrections (stonesfinder.py:178-222).  A deletion (or a moved stone) puts the emptied
        intersection under watch: its pixels are averaged over nb_del_samples calm frames into
        saved_bg, and _check_dels refuses new suggestions there until the zone looks different."""
        unprocessed = []
        try:
            while True:
                err, exp = self.corrections.get_nowait()
                if exp is None:
                    self.deleted[(err.y, err.x)] = self.nb_del_samples      # Move.x / Move.y are image coordinates
                elif err is not None and (err.x, err.y) != (exp.x, exp.y):
                    self.deleted[(err.y, err.x)] = self.nb_del_samples      # a stone has been moved
                else:
                    unprocessed.append((err, exp))                          # a missed stone: left to subclasses
        except queue.Empty:
            pass
        for (r, c), nb_left in self.deleted.items():
            if nb_left:
                fg = self.get_foreground()
                x0, y0, x1, y1 = self.getrect(r, c)
                # reference quirk kept: the comparison sits INSIDE the sum (count of pixels below the
                # bound), so the zone is sampled unless every pixel of it is foreground
                if fg is None or np.sum(fg[x0:x1, y0:y1] < 0.1 * (x1 - x0) * (y1 - y0)):
                    self.saved_bg[x0:x1, y0:y1] += self.goban_img[x0:x1, y0:y1] / self.nb_del_samples
                    self.deleted[(r, c)] = nb_left - 1
        if 0 < len(unprocessed):
            raise CorrectionWarning(unprocessed, message="Unhandled corrections")

    def get_foreground(self):
        return self._fg

    def _check_dels(self, r, c):
        """stonesfinder.py:223-245: has this intersection been deleted by the user recently?"""
        try:
            nb_samples_left = self.deleted[(r, c)]
        except KeyError:
            return
        if 0 == nb_samples_left:                       # only check when sampling has completed
            x0, y0, x1, y1 = self.getrect(r, c)
            diff = self.saved_bg[x0:x1, y0:y1] - self.goban_img[x0:x1, y0:y1]
            if np.sum(np.absolute(diff)) / (diff.shape[0] * diff.shape[1]) < 40:
                raise DeletedError(((r, c),), "The zone has not changed enough since last deletion.")
            print("previously user-deleted location: {} now unlocked".format((r, c)))
            del self.deleted[(r, c)]
        else:
            raise DeletedError(((r, c),), "The zone has been marked as deleted too recently.")

    # ---- result sink ---------------------------------------------------------------------------
    def suggest(self, color, r, c, doprint=True):
        self._check_dels(r, c)
        move = Move(NP_TYPE, (color, r, c))
        if doprint:
            print(move)
        self.vmanager.controller.pipe("append", move)
        self.vmanager.controller.pipe("auto_save")

    def remove(self, r, c):
        assert not self.is_empty(r, c), "Can't remove stone from empty intersection."
        move = Move(NP_TYPE, ("", r, c))
        self.vmanager.controller.pipe("delete", move.x, move.y)

    def bulk_update(self, tuples):
        moves, del_errors = [], []
        for color, r, c in tuples:
            if color == E:
                if not self.is_empty(r, c):
                    moves.append(Move(NP_TYPE, (color, r, c)))
            elif color in (B, W):
                if not self.is_empty(r, c):
                    existing = self.vmanager.controller.locate(c, r)
                    if color != existing.color:
                        moves.append(Move(NP_TYPE, (E, r, c)))      # clear first, then recolour
                    else:
                        continue
                try:
                    self._check_dels(r, c)
                    moves.append(Move(NP_TYPE, (color, r, c)))
                except DeletedError as de:
                    del_errors.append(de)
        if moves:
            self.vmanager.controller.pipe("bulk", moves)
            self.vmanager.controller.pipe("auto_save")
        if del_errors:
            raise DeletedError(del_errors, message="Bulk_update:warning: All non-conflicting locations have been sent.")

    def corrected(self, err_move, exp_move):
        try:
            self.corrections.put_nowait((err_move, exp_move))
        except queue.Full:
            print("Corrections queue full (%s), ignoring %s -> %s" % (correc_size, err_move, exp_move))

    def is_empty(self, r, c):
        return self.vmanager.controller.is_empty_blocking(c, r)

    def get_stones(self):
        return self.vmanager.controller.get_stones()

    # ---- grid geometry ---------------------------------------------------------------------------
    def getrect(self, r, c, cursor=1.0):
        """pixel rectangle (x0, y0, x1, y1) around intersection (r, c); the last row/column
        ends at 379 (reference: stonesfinder.py:412-450)"""
        assert isinstance(cursor, float)
        g = self._posgrid
        p = g.mtx[r][c]
        before = g.mtx[r - 1][c - 1].copy()
        after = g.mtx[min(r + 1, gsize - 1)][min(c + 1, gsize - 1)].copy()
        if r == 0:
            before[0] = -p[0]
        elif r == gsize - 1:
            after[0] = 2 * g.size - p[0] - 2
        if c == 0:
            before[1] = -p[1]
        elif c == gsize - 1:
            after[1] = 2 * g.size - p[1] - 2
        w = cursor / 2
        x0 = max(0, int(w * before[0] + (1 - w) * p[0]))
        y0 = max(0, int(w * before[1] + (1 - w) * p[1]))
        x1 = min(g.size, int((1 - w) * p[0] + w * after[0]))
        y1 = min(g.size, int((1 - w) * p[1] + w * after[1]))
        return x0, y0, x1, y1

    def _window_name(self):
        return "camkifu.stone.stonesfinder.StonesFinder"


class PosGrid:
    """pixel position of each goban intersection in the canonical image: (10 + 20 i, 10 + 20 j)"""

    def __init__(self, size):
        self.size = size
        self.mtx = np.zeros((gsize, gsize, 2), dtype=np.int16)
        start = size / gsize / 2
        end = size - start
        for i in range(gsize):
            xi = (start * (gsize - 1 - i) + end * i) / (gsize - 1)
            for j in range(gsize):
                self.mtx[i][j][0] = xi
                self.mtx[i][j][1] = (start * (gsize - 1 - j) + end * j) / (gsize - 1)
        self.adjust_vect = np.zeros(2, dtype=np.float32)
        self.adjust_contribs = 0
