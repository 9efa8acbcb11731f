"""Fast video-file processing: batches of frames through the stateless GPU core, sharded over
the GPUs of one node, one RCCL gather of fixed-size per-frame records, then the ordered host
fold that carries the reference's temporal logic.

What shards and what does not: per frame, K1..K6 (frame -> Hough lines) and, given
a transform, K8 + K10..K12 (frame -> 19x19 labels) are stateless, so frame f of a batch goes to
rank f mod world.  Line accumulation over 4 frames, corner clustering, the hold-off after a hit
and the move-emission policy are stateful and are replayed in frame order on the gathered records.
There is no data-path collective besides the gather: frames never cross xGMI.
"""
import numpy as np

from . import cvconf
from .golib_shim import gsize, E, B, W

LMAX = 64                                   # Hough lines kept per frame record
REC_LABELS = 0
REC_CONF = 368                              # 8-byte aligned
REC_HDR = REC_CONF + 361 * 8                # status, n_contours, n_lines, pad : 4 x int32
REC_AREA = REC_HDR + 16                     # biggest_area float64
REC_LINES = REC_AREA + 8                    # LMAX x (rho, theta) float32
REC_BYTES = REC_LINES + LMAX * 8            # = 3792
_COLORS = (E, B, W)


def shard_indices(n, rank, world):
    """frame f -> rank f mod world"""
    return np.arange(rank, n, world)


def pack_records(board, labels, conf):
    """board: list of per-frame dicts from Context.board_detect; labels (n,19,19) u8; conf (n,19,19) f64
    -> (n, REC_BYTES) uint8"""
    n = len(board)
    rec = np.zeros((n, REC_BYTES), np.uint8)
    if n == 0:
        return rec
    rec[:, REC_LABELS:REC_LABELS + 361] = np.asarray(labels, np.uint8).reshape(n, 361)
    rec[:, REC_CONF:REC_CONF + 361 * 8] = np.ascontiguousarray(conf, np.float64).reshape(n, 361).view(np.uint8)
    for f, b in enumerate(board):
        if b["n_lines"] > LMAX:
            raise ValueError("frame record holds %d Hough lines, %d found" % (LMAX, b["n_lines"]))
        hdr = np.array([b["status"], b["n_contours"], b["n_lines"], 0], np.int32)
        rec[f, REC_HDR:REC_HDR + 16] = hdr.view(np.uint8)
        rec[f, REC_AREA:REC_AREA + 8] = np.array([b["biggest_area"]], np.float64).view(np.uint8)
        k = b["n_lines"]
        if k:
            rec[f, REC_LINES:REC_LINES + 8 * k] = np.ascontiguousarray(b["lines"][:k], np.float32).reshape(-1).view(np.uint8)
    return rec


def unpack_record(rec):
    hdr = rec[REC_HDR:REC_HDR + 16].view(np.int32)
    k = int(hdr[2])
    return dict(status=int(hdr[0]), n_contours=int(hdr[1]), n_lines=k,
                biggest_area=float(rec[REC_AREA:REC_AREA + 8].view(np.float64)[0]),
                lines=rec[REC_LINES:REC_LINES + 8 * k].view(np.float32).reshape(k, 2).copy(),
                labels=rec[REC_LABELS:REC_LABELS + 361].reshape(19, 19).copy(),
                conf=rec[REC_CONF:REC_CONF + 361 * 8].view(np.float64).reshape(19, 19).copy())


def gather_records(local, n_total, rank, world, device=None):
    """One all-gather of this rank's records; returns the (n_total, REC_BYTES) array in frame order.
    `torch.distributed` must be initialised when world > 1 (backend nccl = RCCL on GPUs, gloo on CPU)."""
    if world == 1:
        return local
    import torch
    import torch.distributed as dist
    per = (n_total + world - 1) // world                       # ranks with fewer frames pad
    buf = torch.zeros((per, REC_BYTES), dtype=torch.uint8, device=device)
    if len(local):
        buf[:len(local)] = torch.from_numpy(local).to(buf.device)
    out = torch.empty((world * per, REC_BYTES), dtype=torch.uint8, device=buf.device)
    dist.all_gather_into_tensor(out, buf)
    out = out.cpu().numpy().reshape(world, per, REC_BYTES)
    full = np.zeros((n_total, REC_BYTES), np.uint8)
    for r in range(world):
        idx = shard_indices(n_total, r, world)
        full[idx] = out[r, :len(idx)]
    return full


class _Shape:
    """what BoardFinderAuto._detect needs from a frame once the image chain has run: its shape"""

    def __init__(self, h, w):
        self.shape = (h, w, 3)


class BoardFold:
    """Ordered replay of BoardFinderAuto's temporal logic on per-frame records.  The wall-clock
    hold-off after a hit (bf_auto.py:43-49, 10 s) becomes a frame count at the file read rate."""

    def __init__(self, h, w, refresh_frames=None):
        from .board.bf_auto import BoardFinderAuto

        class _VM:
            imqueue = None
        self.finder = BoardFinderAuto(_VM(), ctx=False)        # ctx=False: records only, no GPU calls
        self.frame = _Shape(h, w)
        self.refresh_frames = 10 * cvconf.file_fps if refresh_frames is None else refresh_frames
        self.hold = 0

    @property
    def mtx(self):
        return self.finder.mtx

    def step(self, rec):
        f = self.finder
        if self.hold > 0:
            self.hold -= 1
        else:
            f.corners.frame = self.frame
            if f._detect(self.frame, core=rec):
                from . import capi
                f.mtx = capi.get_perspective_transform(np.array(f.corners.hull, np.float32), f.transform_dst)
                self.hold = self.refresh_frames
        f.total_f_processed += 1
        return f.mtx


class StonesFold:
    """Per-frame full-board assessment (NNCache.predict_all_stones + SfNeural.predict_all's
    acceptance rule: colour != E and confidence > 0.6) pushed to the controller with
    StonesFinder.bulk_update semantics."""

    MIN_CONFIDENCE = 0.6

    def __init__(self, controller):
        self.controller = controller

    def step(self, labels, conf):
        from .golib_shim import Move, NP_TYPE
        moves = []
        for r in range(gsize):
            for c in range(gsize):
                color = _COLORS[labels[r, c]]
                if color == E or not (conf[r, c] > self.MIN_CONFIDENCE):
                    continue
                existing = self.controller.locate(c, r)
                if existing is not None:
                    if existing.color == color:
                        continue
                    moves.append(Move(NP_TYPE, (E, r, c)))
                moves.append(Move(NP_TYPE, (color, r, c)))
        if moves:
            self.controller.pipe("bulk", moves)
            self.controller.pipe("auto_save")
        return moves


class FastFilePipeline:
    """compute(frames, mtx) -> (board list, labels, conf) is the stateless per-shard core; by
    default it is the HIP context (ck_board_detect + ck_stones_detect)."""

    def __init__(self, h, w, controller, ctx=None, rank=0, world=1, device=None, compute=None, ctx_board=None):
        """ctx runs the stones path; ctx_board (optional second context = second HIP stream) lets the
        board path run concurrently on its own host thread, as the reference's two finder threads do"""
        self.h, self.w = h, w
        self.rank, self.world, self.device = rank, world, device
        self.ctx = ctx
        self.ctx_board = ctx_board
        self._pool = None
        self.compute = compute or self._gpu_compute
        self.board = BoardFold(h, w)
        self.stones = StonesFold(controller)
        self.frames_done = 0

    def _gpu_compute(self, frames, mtx):
        if self.ctx_board is not None and mtx is not None:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(1)
            fut = self._pool.submit(self.ctx_board.board_detect, frames, -1, LMAX)
            labels, conf = self.ctx.stones_detect(frames, mtx)
            board = fut.result()
        else:
            board = (self.ctx_board or self.ctx).board_detect(frames, cap=LMAX)
            if mtx is None:
                n = len(board)
                return board, np.zeros((n, 19, 19), np.uint8), np.zeros((n, 19, 19), np.float64)
            labels, conf = self.ctx.stones_detect(frames, mtx)
        if hasattr(labels, "cpu"):
            labels, conf = labels.cpu().numpy(), conf.cpu().numpy()
        return board, labels, conf

    def process_batch(self, my_frames, n_total):
        """my_frames: this rank's shard (frames rank, rank+world, ... of the batch).  The transform
        used for the stones path is the one known at the start of the batch (board assumed fixed
        within a batch; it is re-estimated by the fold for the next one).  Returns the per-frame
        move lists emitted by the fold (identical on every rank)."""
        mtx = self.board.mtx
        board, labels, conf = self.compute(my_frames, mtx)
        rec = pack_records(board, labels, conf)
        full = gather_records(rec, n_total, self.rank, self.world, self.device)
        emitted = []
        for f in range(n_total):
            r = unpack_record(full[f])
            self.board.step(r)
            emitted.append(self.stones.step(r["labels"], r["conf"]) if mtx is not None else [])
            self.frames_done += 1
        return emitted
