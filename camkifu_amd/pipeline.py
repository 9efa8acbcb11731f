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


def pack_records_raw(res, lines, labels, conf):
    """vectorised pack_records for Context.board_detect(raw=True) output"""
    n = len(res)
    if (res["n_lines"] > LMAX).any():
        raise ValueError("frame record holds %d Hough lines, %d found" % (LMAX, int(res["n_lines"].max())))
    rec = np.zeros((n, REC_BYTES), np.uint8)
    rec[:, REC_LABELS:REC_LABELS + 361] = np.asarray(labels, np.uint8).reshape(n, 361)
    rec[:, REC_CONF:REC_CONF + 361 * 8] = np.ascontiguousarray(conf, np.float64).reshape(n, 361).view(np.uint8)
    hdr = np.stack([res["status"], res["n_contours"], res["n_lines"], np.zeros(n, np.int32)], 1).astype(np.int32)
    rec[:, REC_HDR:REC_HDR + 16] = hdr.view(np.uint8).reshape(n, 16)
    rec[:, REC_AREA:REC_AREA + 8] = np.ascontiguousarray(res["biggest_area"], np.float64).view(np.uint8).reshape(n, 8)
    keep = np.arange(LMAX)[None, :] < res["n_lines"][:, None]               # lines beyond n_lines are zeroed
    ln = np.where(keep[..., None], np.ascontiguousarray(lines[:, :LMAX], np.float32), np.float32(0))
    rec[:, REC_LINES:REC_LINES + LMAX * 8] = np.ascontiguousarray(ln).view(np.uint8).reshape(n, LMAX * 8)
    return rec


def unpack_record(rec):
    hdr = rec[REC_HDR:REC_HDR + 16].view(np.int32)
    k = int(hdr[2])
    return dict(status=int(hdr[0]), n_contours=int(hdr[1]), n_lines=k,
                biggest_area=float(rec[REC_AREA:REC_AREA + 8].view(np.float64)[0]),
                lines=rec[REC_LINES:REC_LINES + 8 * k].view(np.float32).reshape(k, 2).copy(),
                labels=rec[REC_LABELS:REC_LABELS + 361].reshape(19, 19).copy(),
                conf=rec[REC_CONF:REC_CONF + 361 * 8].view(np.float64).reshape(19, 19).copy())


def unpack_records(rec):
    """vectorised view of an (n, REC_BYTES) record array -> dict of arrays (lines stay packed:
    use lines[f, :n_lines[f]])"""
    n = len(rec)
    hdr = np.ascontiguousarray(rec[:, REC_HDR:REC_HDR + 16]).view(np.int32).reshape(n, 4)
    return dict(status=hdr[:, 0], n_contours=hdr[:, 1], n_lines=hdr[:, 2],
                biggest_area=np.ascontiguousarray(rec[:, REC_AREA:REC_AREA + 8]).view(np.float64).reshape(n),
                lines=np.ascontiguousarray(rec[:, REC_LINES:REC_LINES + LMAX * 8]).view(np.float32).reshape(n, LMAX, 2),
                labels=rec[:, REC_LABELS:REC_LABELS + 361].reshape(n, 19, 19),
                conf=np.ascontiguousarray(rec[:, REC_CONF:REC_CONF + 361 * 8]).view(np.float64).reshape(n, 19, 19))


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
        self.cur = None                        # cached goban as uint8 (19,19): 0 E, 1 B, 2 W
        # prisoners the rule engine took off the goban but the camera may still show: (r, c) -> label.
        # They are not suggested again until the intersection has been seen empty or recoloured.
        self.prisoners = {}

    def resync(self):
        """re-read the goban from the controller (once per batch: somebody else may edit it)"""
        st = self.controller.get_stones()
        self.cur = np.zeros((gsize, gsize), np.uint8)
        self.cur[st == B] = 1
        self.cur[st == W] = 2

    def step(self, labels, conf):
        from .golib_shim import Move, NP_TYPE
        if self.cur is None:
            self.resync()
        change = (labels != 0) & (conf > self.MIN_CONFIDENCE) & (labels != self.cur)
        if self.prisoners:
            for (r, c), lab in list(self.prisoners.items()):
                if labels[r, c] == lab:
                    change[r, c] = False                     # still lying on the board: not a new stone
                else:
                    del self.prisoners[(r, c)]               # taken away (or replaced): watch over
        if not change.any():
            return []
        moves = []
        for r, c in np.argwhere(change):          # raster order, like the reference's double loop
            r, c = int(r), int(c)
            color = _COLORS[labels[r, c]]
            if self.cur[r, c] != 0:
                moves.append(Move(NP_TYPE, (E, r, c)))       # clear first, then recolour
            moves.append(Move(NP_TYPE, (color, r, c)))
            self.cur[r, c] = labels[r, c]
        if getattr(self.controller, "rules", None) is None:
            self.controller.pipe("bulk", moves)
        else:
            # one instruction per stone so the prisoners of each can be taken off the cached goban
            for mv in moves:
                self.controller.pipe("bulk", [mv])
                for col, cx, cy in (self.controller.last_captured if mv.color != E else ()):
                    self.prisoners[(cy, cx)] = 1 if col == B else 2
                    self.cur[cy, cx] = 0
        self.controller.pipe("auto_save")
        return moves

    def step_batch(self, labels, conf):
        """ordered fold of a whole batch; frames whose accepted labels equal the cached goban are
        skipped without touching Python per frame -> list of per-frame move lists"""
        if self.cur is None:
            self.resync()
        n = len(labels)
        acc = np.where(conf > self.MIN_CONFIDENCE, labels, 0)           # accepted colour or 0
        # a frame can only emit moves if it differs from the goban as left by its predecessor;
        # cheap superset: differs from the previous frame's accepted labels or from the cache
        prev = np.concatenate([self.cur[None], acc[:-1]])
        cand = ((acc != prev) & (acc != 0)).reshape(n, -1).any(1)
        out = [[] for _ in range(n)]
        for f in np.flatnonzero(cand | (np.arange(n) == 0)):
            out[f] = self.step(labels[f], conf[f])
        return out


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
        return self.fold(full, mtx is not None)

    def process_y4m(self, capture, batch=256, file_fps=None, torch_device=None):
        """Fast processing of a video file (README "Fast video file processing"; frame selection as
        CaptureReaderBase.skip, core/vmanager.py:511-525).  `capture` is a core.capture.Y4MCapture: the
        frames to analyse are dealt to the ranks batch by batch (frame k of a batch -> rank k mod world),
        each rank uploads ITS frames as I420 (1.5 B/px through a reused pinned buffer) and converts them
        to BGR in HBM (ck_i420_to_bgr), then the batch goes through process_batch.
        Returns the concatenated per-frame move lists."""
        import torch
        from .core.capture import file_frame_indices
        idx = file_frame_indices(len(capture), capture.fps, file_fps)
        dev = torch_device if torch_device is not None else torch.device("cuda", getattr(self.ctx, "device", 0))
        pinned, emitted = None, []
        for b0 in range(0, len(idx), batch):
            chunk = idx[b0:b0 + batch]
            mine = [chunk[k] for k in shard_indices(len(chunk), self.rank, self.world)]
            if pinned is None:
                cap_rows = len(shard_indices(batch, 0, self.world))
                pinned = torch.empty((cap_rows, capture.fsize), dtype=torch.uint8).pin_memory()
            raw = capture.read_raw_batch(mine, out=pinned.numpy())
            if len(mine):
                frames = self.ctx.i420_to_bgr(raw, capture.h, capture.w, to_device=dev)
            else:
                frames = torch.empty((0, capture.h, capture.w, 3), dtype=torch.uint8, device=dev)
            emitted.extend(self.process_batch(frames, len(chunk)))
        return emitted

    def fold(self, full, have_mtx=True):
        """ordered replay of the temporal logic on the gathered records of one batch"""
        u = unpack_records(full)
        n = len(full)
        self.stones.resync()
        f = 0
        while f < n:
            if self.board.hold > 0:                    # hold-off after a hit: nothing to look at
                skip = min(self.board.hold, n - f)
                self.board.hold -= skip
                self.board.finder.total_f_processed += skip
                f += skip
                continue
            k = int(u["n_lines"][f])
            self.board.step(dict(status=int(u["status"][f]), n_contours=int(u["n_contours"][f]), n_lines=k,
                                 biggest_area=float(u["biggest_area"][f]), lines=u["lines"][f, :k]))
            f += 1
        emitted = self.stones.step_batch(u["labels"], u["conf"]) if have_mtx else [[] for _ in range(n)]
        self.frames_done += n
        return emitted
