"""Fast video-file processing: batches of frames through the stateless GPU core, sharded over the GPUs of
one node, fixed-size per-frame records gathered TO RANK 0 over RCCL, then the ORDERED fold there.

What shards and what does not.  Per frame, K1..K6 (frame -> Hough lines) and, given a
transform, K8 + K10..K12 (frame -> the classifier's answer for its 100 regions) are stateless: frame f of a
batch goes to rank f mod world and never leaves its GPU.  Stateful, and therefore replayed in frame order on the
gathered records by ONE rank (the library's ck_boardfold_step / ck_policy_run, the very code the per-frame
finders call): 4-frame line accumulation, corner clustering, the hold-off after a hit, and the stones policy
(full assessment, agitation targets, colour-ratio veto, lookback).  The board transform the fold arrives at is
broadcast back (72 bytes) for the next batch.

The background model (K9, MOG2) is per-pixel state over TIME, so it cannot follow the frame sharding; it shards
by PIXEL instead: with world > 1 every rank keeps the mixtures of a band of intersection rows, receives that
band of every goban image of the batch in one all-to-all (xGMI), runs the band through the whole batch in frame
order (ck_mog2_band_run) and contributes its foreground counts to the gather.  With world == 1 the whole chain
is one call (ck_stones_run).

Where the records live.  The library writes both halves of a frame's record (ck_frame_record, 1440 bytes) IN PLACE into
one buffer per shard -- in HBM when the records go through RCCL, in host memory otherwise -- so nothing is packed by
numpy and nothing crosses PCIe before the gather; only rank 0 brings the gathered records (and the foreground counts) to
the host, in one copy each.  The other ranks learn a batch's outcome from the transform broadcast (16 doubles) and a
one-word all-reduce after the background model.

A failed batch (finish() raises on every rank) leaves a gap: the board fold on rank 0 advances over the frames whose
records are missing as frames without a contour, so the running frame count -- which places the every-4th-frame grouping
and, in hold-off-aware mode, every rank's plan of the next batch -- stays that of the film."""
import os

import numpy as np

from . import capi, cvconf
from .golib_shim import gsize, E, B, W
from .stone import nn_manager as nm

LMAX = capi.REC_LMAX                         # Hough lines carried per frame record (more are flagged, not fatal)
FLAG_LINES_CUT, FLAG_FAILED = capi.REC_LINES_CUT, capi.REC_FAILED
REC = capi.REC_DTYPE                         # ck_frame_record (include/camkifu_amd.h)
REC_BYTES = capi.REC_BYTES
assert REC_BYTES % 8 == 0
FLAGS_AT = REC.fields["flags"][1]            # byte offset of a record's flags word
_SYMBOL = (E, B, W)


def shard_indices(n, rank, world):
    """frame f -> rank f mod world"""
    return np.arange(rank, n, world)


def band_rows(world):
    """intersection rows [a, b) whose background model lives on each rank (19 rows dealt as evenly as possible)"""
    cuts = np.linspace(0, gsize, world + 1).round().astype(int)
    return [(int(cuts[k]), int(cuts[k + 1])) for k in range(world)]


def pack_records(board, region_label, region_conf, failed=False):
    """board: (structured BOARD_DTYPE array, lines (n, cap, 2)) as Context.board_detect(raw=True) returns them, or a
    list of per-frame dicts; region_label (n, 10, 10) u8; region_conf (n, 10, 10) f64 -> REC array (n,).
    Never raises on content: a frame with more than LMAX lines keeps the first LMAX (OpenCV's order = most votes
    first) and is flagged, so that no rank can fail before a collective the others are already waiting in."""
    dict_form = isinstance(board, (list, tuple)) and (len(board) == 0 or isinstance(board[0], dict))
    n = len(board) if dict_form else len(board[0])
    rec = np.zeros(n, REC)
    if n == 0:
        return rec
    fill_board(rec, board)
    if failed:
        rec["flags"] |= FLAG_FAILED
    rec["region_label"] = np.asarray(region_label, np.uint8).reshape(n, 10, 10)
    rec["region_conf"] = np.asarray(region_conf, np.float64).reshape(n, 10, 10)
    return rec


def fill_board(rec, board):
    """the board half of the REC rows `rec` (a numpy view, written in place) from what Context.board_detect hands over -- the
    host-side twin of ck_board_detect_records, for the stand-in contexts of the CPU tests"""
    if isinstance(board, (list, tuple)) and (len(board) == 0 or isinstance(board[0], dict)):
        res = np.zeros(len(board), capi.BOARD_DTYPE)
        lines = np.zeros((len(board), LMAX, 2), np.float32)
        for f, b in enumerate(board):
            res[f] = (b["status"], b["n_contours"], b["n_lines"], 0, b["biggest_area"])
            k = min(int(b["n_lines"]), LMAX, len(b["lines"]))
            lines[f, :k] = np.asarray(b["lines"], np.float32).reshape(-1, 2)[:k]
    else:
        res, lines = board
    if len(res) == 0:
        return rec
    for name in ("status", "n_contours", "n_lines", "biggest_area"):
        rec[name] = res[name]
    rec["flags"] = np.where(res["n_lines"] > LMAX, FLAG_LINES_CUT, 0)
    kept = np.minimum(res["n_lines"], LMAX)
    width = min(LMAX, lines.shape[1])
    live = np.arange(width)[None, :] < kept[:, None]
    rec["lines"][:, :width] = np.where(live[..., None], lines[:, :width], np.float32(0))
    rec["lines"][:, width:] = 0
    return rec


def record_buffer(rows, device=None):
    """`rows` zeroed records in one buffer: a numpy REC array (host), or a torch uint8 tensor (rows, REC_BYTES) on `device`"""
    if device is None:
        return np.zeros(rows, REC)
    import torch
    return torch.zeros((rows, REC_BYTES), dtype=torch.uint8, device=device)


def shard_buffer(n, device=None):
    """the record buffer of a shard of n frames: row 0 is the shard's HEADER (only its flags word is used: a rank whose
    shard is empty can still say that it failed), rows 1 .. n the frames' records, and one spare row, so that the first
    ceil(n_total / world) + 1 rows are what this rank contributes to the gather whatever the deal gave it"""
    return record_buffer(n + 2, device)


def records_view(buf):
    """a host record buffer (numpy REC array, or uint8 bytes of it) as a REC array"""
    if not isinstance(buf, np.ndarray):
        buf = (buf.cpu() if buf.is_cuda else buf).numpy()     # (records forced into HBM on a rank that folds them itself)
    return buf if buf.dtype == REC else buf.view(REC).reshape(buf.shape[:-1])


def board_into(ctx, frames, rows):
    """K1-K6 of `frames` -> the board half of the record rows, in place; a context without the record entry points (the
    stand-ins of the CPU tests) answers in the old form and the rows are filled here"""
    if hasattr(ctx, "board_detect_records"):
        return ctx.board_detect_records(frames, rows)
    return fill_board(rows, ctx.board_detect(frames, -1, LMAX, True))


def regions_into(ctx, gobans, rows):
    """K10-K12 of the goban images -> the stones half of the record rows, in place"""
    if hasattr(ctx, "cnn_regions_records"):
        return ctx.cnn_regions_records(gobans, rows)
    lab, conf = ctx.cnn_regions(gobans)
    rows["region_label"] = np.asarray(GpuCore._host(lab), np.uint8).reshape(len(rows), 10, 10)
    rows["region_conf"] = np.asarray(GpuCore._host(conf), np.float64).reshape(len(rows), 10, 10)
    return rows


def grid_of(region_label, region_conf=None):
    """(n, 10, 10) region labels -> the (n, 19, 19) grid NNCache.predict_all_stones builds (later regions win on
    row / column 17); with region_conf also the per-intersection confidences"""
    lab = np.asarray(region_label)
    n = lab.shape[0]
    cell_region = np.minimum(np.arange(gsize) // nm.STEP, nm.SPLIT - 1)
    cell_region[gsize - nm.STEP:] = nm.SPLIT - 1
    local = np.arange(gsize) - nm.REGION_START[cell_region]
    digit = local[:, None] * nm.STEP + local[None, :]
    per_cell = lab[:, cell_region[:, None], cell_region[None, :]]
    grid = nm.DIGITS[per_cell, np.broadcast_to(digit, (n, gsize, gsize))]
    if region_conf is None:
        return grid
    return grid, np.asarray(region_conf)[:, cell_region[:, None], cell_region[None, :]]


class _Group:
    """the few collectives the pipeline needs, on whatever backend torch.distributed was initialised with
    (nccl = RCCL over xGMI on GPUs, gloo on CPU); world == 1 needs no process group at all.

    What is gathered is gathered TO RANK 0 (the rank that folds) and comes to the host there, in one copy; the other
    ranks bring nothing to the host but the broadcast wire and one flag word per batch.  `host_bytes` counts what each
    kind of collective brought to THIS rank's host memory (tests hold ranks != 0 to the wire and the flags)."""

    def __init__(self, rank, world, device):
        self.rank, self.world, self.device = rank, world, device
        self.host_bytes = dict(gather=0, bcast=0, flag=0)
        self._landing = {}                                    # free pinned host buffers the gathers land in, by (shape, dtype)

    def _to_host(self, t, lease):
        """a device tensor -> numpy, through PINNED memory (one DMA, no staging through a pageable bounce buffer).  The
        buffer is taken from a free list (allocated when that is empty: a handful over a run) and noted on `lease`; whoever
        holds the lease gives the buffers back (release) when nothing reads the arrays any more -- a batch's gathered records
        are read by its folds while the next batches are being gathered."""
        import torch
        if not t.is_cuda:
            return t.numpy()
        key = (tuple(t.shape), t.dtype)
        free = self._landing.setdefault(key, [])
        buf = free.pop() if free else torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        lease.append((key, buf))
        buf.copy_(t)                                          # (blocking: complete when it returns)
        return buf.numpy()

    def release(self, lease):
        while lease:
            key, buf = lease.pop()
            self._landing.setdefault(key, []).append(buf)

    def on_wire(self, a):
        """a numpy array or a torch tensor -> a tensor where the collectives' buffers live (no copy when it is there already)"""
        import torch
        if isinstance(a, np.ndarray) and a.dtype.fields is not None:
            a = np.ascontiguousarray(a).view(np.uint8).reshape(a.shape + (a.dtype.itemsize,))       # records: as bytes, no copy
        t = a if hasattr(a, "is_cuda") else torch.from_numpy(np.ascontiguousarray(a))
        want = torch.device(self.device) if self.device is not None else torch.device("cpu")
        return t if t.device == want else t.to(want)

    def gather_rows(self, mine, lease):
        """every rank contributes the same number of rows (a tensor or an array, any dtype, first axis = rows) -> on rank 0 a
        numpy uint8 array (world, rows, bytes per row), None elsewhere.  ONE device-to-host copy, on rank 0 only, into a
        pinned buffer noted on `lease` (see _to_host)."""
        import torch
        import torch.distributed as dist
        t = self.on_wire(mine)
        t = t.reshape(t.shape[0], -1)
        if t.dtype != torch.uint8:
            t = t.view(torch.uint8)
        if not t.is_contiguous():
            t = t.contiguous()
        if self.rank != 0:
            dist.gather(t, None, dst=0)
            return None
        out = torch.empty((self.world,) + tuple(t.shape), dtype=torch.uint8, device=t.device)
        dist.gather(t, list(out.unbind(0)), dst=0)
        self.host_bytes["gather"] += out.numel()
        return self._to_host(out, lease)

    def _small(self, key, shape, dtype):
        """a persistent small tensor where the collectives' buffers live (and, for device buffers, its pinned host twin): the
        wire and the flag word are a few bytes per batch -- no allocation, no pageable staging copy"""
        import torch
        slot = self._landing.get(("small", key, shape))
        if slot is None:
            dev = torch.device(self.device) if self.device is not None else torch.device("cpu")
            t = torch.zeros(shape, dtype=dtype, device=dev)
            slot = (t, torch.zeros(shape, dtype=dtype, pin_memory=True) if t.is_cuda else None)
            self._landing[("small", key, shape)] = slot
        return slot

    def broadcast_array(self, a, src=0):
        """-> the source's array (float64) on every rank; the source itself reads nothing back (it holds the array) and
        uploads through pinned memory without waiting"""
        import torch
        import torch.distributed as dist
        a = np.ascontiguousarray(a, np.float64)
        t, pinned = self._small("wire", tuple(a.shape), torch.float64)
        if self.rank == src:
            if pinned is not None:
                pinned.numpy()[...] = a
                t.copy_(pinned, non_blocking=True)
            else:
                t.copy_(torch.from_numpy(a))
        dist.broadcast(t, src)
        if self.rank == src:
            return a
        self.host_bytes["bcast"] += t.numel() * t.element_size()
        return t.cpu().numpy()

    def max_flag_start(self, value):
        """one int32 word all-reduced with MAX: every rank learns whether (and which) rank raised a flag.  Issued here, read
        with max_flag_read -- after whatever else the caller queues behind it, so that ONE host wait covers both"""
        import torch
        import torch.distributed as dist
        t, _ = self._small("flag", (1,), torch.int32)
        t.fill_(int(value))                                   # (a fill on the device: nothing to upload)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t

    def max_flag_read(self, t):
        self.host_bytes["flag"] += 4
        return int(t.item())

    def max_flag(self, value):
        return self.max_flag_read(self.max_flag_start(value))

    def all_to_all_bands(self, send_parts, recv_sizes, self_through_collective=False):
        """send_parts[d]: torch uint8 tensor (any strides) for rank d; recv_sizes[s]: bytes rank s sends here (both sides
        can work them out from the batch size) -> list of what every rank sent here.  Each part is copied ONCE, straight
        into its slot of the send buffer; this rank's own part does not go through the collective at all (a device copy),
        unless `self_through_collective` (kept for exercising the call with a single rank)."""
        import torch
        import torch.distributed as dist
        me = self.rank
        own = None if self_through_collective else send_parts[me].contiguous()
        sizes = [int(p.numel()) if (d != me or own is None) else 0 for d, p in enumerate(send_parts)]
        recv_sizes = [int(r) if (s != me or own is None) else 0 for s, r in enumerate(recv_sizes)]
        send = torch.empty(sum(sizes), dtype=torch.uint8, device=send_parts[0].device)
        at = 0
        for p, k in zip(send_parts, sizes):
            if k:
                send[at:at + k].view(p.shape).copy_(p)
            at += k
        recv = torch.empty(int(sum(recv_sizes)), dtype=torch.uint8, device=send.device)
        if self.world > 1 or own is None:                     # (the same decision on every rank)
            dist.all_to_all_single(recv, send, recv_sizes, sizes)
        parts = list(torch.split(recv, recv_sizes))
        if own is not None:
            parts[me] = own.reshape(-1)
        return parts


def rccl_group_options():
    """What a host passes as `pg_options=` to `init_process_group("nccl", ...)` for the group it hands to
    FastFilePipeline: the communicator's own stream at high priority, like the exchange thread's (see _exchange)."""
    import os
    import torch.distributed as dist
    opts = dist.ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = os.environ.get("CK_EXCHANGE_PRIORITY") != "0"
    return opts


def _on_device(device):
    """initializer of every worker thread of the pipeline: torch's current device is per-thread state and starts at
    device 0 in a new thread -- on rank r > 0 a worker would otherwise allocate, copy and synchronise on GPU 0"""
    if device is None:
        return None

    def init():
        import torch
        if torch.cuda.is_available():
            torch.cuda.set_device(device)
    return init


def _pool(workers, device):
    from concurrent.futures import ThreadPoolExecutor
    init = _on_device(device)
    return ThreadPoolExecutor(workers, initializer=init) if init else ThreadPoolExecutor(workers)


def _device_of(ctx):
    """the GPU index of a real context, None for the stand-ins the CPU tests use"""
    return ctx.device if isinstance(ctx, capi.Context) else None


def _take(frames, indices, keep=None):
    """-> (the frames with those indices, contiguous; the torch stream their gather was queued on, or None when there is
    nothing to wait for: a slice of consecutive frames, host arrays).  No host wait: the gather runs on this thread's
    current torch stream and whoever hands the result to the library on another thread orders the context's stream behind
    that stream first (Context.wait_stream -- capi._in only knows the consumer thread's own current stream).  The index
    tensor goes onto `keep` (the batch's ticket holds it until finish())."""
    if len(indices) and indices[-1] - indices[0] + 1 == len(indices):
        return frames[indices[0]:indices[-1] + 1], None
    if hasattr(frames, "index_select"):
        import torch
        index = torch.as_tensor(indices, device=frames.device)
        out = frames.index_select(0, index)
        if keep is not None:
            keep.append(index)
        return out, (torch.cuda.current_stream(out.device) if out.is_cuda else None)
    return np.ascontiguousarray(np.asarray(frames)[list(indices)]), None


class BoardFold:
    """Ordered replay of the board finder on per-frame records: BoardFinderAuto._detect is called with the record in
    place of the GPU call.  The 10 s wall-clock hold-off after a hit (bf_auto.py:43-49) is a frame count here."""

    def __init__(self, h, w, refresh_frames=None):
        from .board.bf_auto import BoardFinderAuto
        manager = type("FoldOnly", (), {"imqueue": None})()
        self.finder = BoardFinderAuto(manager)
        self.frame = np.zeros((h, w, 0), np.uint8)          # what _detect needs of a frame once the image chain ran: its shape
        self.refresh_frames = 10 * cvconf.file_fps if refresh_frames is None else refresh_frames
        self.hold = 0
        self.seen = self.looked = self.fetched = self.calls = 0   # records offered / looked at / computed lazily / fetch calls
        self.episode = 8                                     # frames the last detection took
        self.recent = [1]                                    # extra grouping rounds the last few detections needed (run_lazy's first request)
        self._run = self._opened = 0                         # frames looked at since the window in progress opened / the count it opened on
        self._mtx_hull = None                                # the hull the transform in force was derived from (_run_records)
        self.generosity = 1                                  # grouping rounds a window's first request covers beyond the recent maximum
        self.rounds_seen = {}                                # how many detections needed 0, 1, 2 ... extra grouping rounds (diagnostics)

    @property
    def mtx(self):
        return self.finder.mtx

    def step(self, rec):
        """rec: one REC row (or a dict with status / n_lines / lines ...)"""
        f = self.finder
        self.seen += 1
        if self.hold > 0:
            self.hold -= 1
        else:
            self.looked += 1
            k = min(int(rec["n_lines"]), LMAX, len(rec["lines"]))
            f.corners.frame = self.frame
            hit = f._detect(self.frame, record=dict(status=int(rec["status"]), n_lines=k, lines=np.asarray(rec["lines"])[:k]))
            if hit:
                self._hit()
        f.total_f_processed += 1
        return f.mtx

    def _hit(self):
        """what BoardFinder._doframe does after a detection (board/boardfinder.py:43-48): the transform from the corners'
        hull; the hold-off starts when there is one"""
        f = self.finder
        try:
            f.mtx = capi.get_perspective_transform(np.asarray(f.corners.hull, np.float32), f.transform_dst)
            self.hold, self._mtx_hull = self.refresh_frames, f.corners.hull
        except (capi.CkError, TypeError, ValueError):
            f.mtx, self._mtx_hull = None, None               # degenerate quadrilateral: keep looking

    def run(self, recs, order=None):
        """the fold over a batch of records.  A contiguous REC array (what the gather leaves on rank 0; frame f at
        recs[order[f]] when `order` is given) goes through ck_boardfold_run: the loop over the frames, the hold-off and the
        per-frame step are the library's, and Python is entered only where the corners change; anything else frame by
        frame (`step`)."""
        if isinstance(recs, np.ndarray) and recs.dtype == REC and recs.flags.c_contiguous:
            return self._run_records(recs, order)
        if order is not None:
            recs = [recs[int(j)] for j in order]
        k, n = 0, len(recs)
        while k < n:
            if self.hold > 0:                                # nothing is looked at during the hold-off
                skip = min(self.hold, n - k)
                self.hold -= skip
                self.seen += skip
                self.finder.total_f_processed += skip
                k += skip
                continue
            self.step(recs[k])
            k += 1
        return self.mtx

    def _run_records(self, recs, order=None):
        f, k = self.finder, 0
        n = len(recs) if order is None else len(order)
        f.corners.frame = self.frame
        h, w = self.frame.shape[0], self.frame.shape[1]
        while k < n:
            # a hit that does not move the corners re-derives the transform in force from the same hull: the library then
            # only starts the hold-off and goes on; otherwise (no valid transform from this hull yet) every hit comes back
            same = self.refresh_frames if (f.mtx is not None and f.corners.hull is not None and f.corners.hull == self._mtx_hull) else -1
            try:
                k, count, self.hold, self.seen, self.looked, found, update, centers, stats = f.core.run(
                    h, w, recs, k, f.total_f_processed, self.hold, self.seen, self.looked, f.corners.hull, order, same)
            except IndexError as why:                        # (the counters stand where the failing frame left them)
                _, f.total_f_processed, self.hold, self.seen, self.looked = why.fold_state
                raise
            f.total_f_processed = count
            if stats is not None:
                f.last_stats = stats
            if update:                                       # what _detect does with the step's answer (board/bf_auto.py)
                f.corners.clear()
                for p in centers:
                    f.corners.submit(p)
            if found and (update or same < 0):
                self._hit()
        return self.mtx

    def state(self):
        """what the next batch's first request depends on (plan_request / first_request): running frame count, frames of
        hold-off left, frames looked at in the window in progress, the count that window opened on"""
        return (int(self.finder.total_f_processed), int(self.hold), int(self._run), int(self._opened))

    @staticmethod
    def plan_request(n, H, k, c, run, opened):
        """The frames of a batch of n a plan-ahead request covers when the fold stands at frame k with running count c and
        the record of k is missing (run_lazy's docstring): a pure function of the fold's state, so that every rank of a
        multi-GPU run can work out the FIRST request of a batch by itself (first_request) and start K1-K6 on the planned
        frames it owns without being told.  H: frames of hold-off after a hit; run / opened: frames looked at in the
        window in progress and the count it opened on (run == 0: a window opens at k)."""
        def chain(k_, c_, j):
            """frames looked at from frame k_ on count c_ if the window in progress hits on its j-th upcoming opportunity
            and every later window of the batch on its first"""
            out = []
            span = (-c_) % 4 + 1 + 4 * j
            while k_ < n:
                out.extend(range(k_, min(n, k_ + span)))
                k_, c_ = k_ + span + H, c_ + span + H
                span = (-c_) % 4 + 1
            return out
        if run == 0:
            return chain(k, c, 0)
        # opportunities this window has had: its first after (-opened) % 4 + 1 frames, then one every four
        first = (-opened) % 4 + 1
        had = 0 if run < first else 1 + (run - first) // 4
        # hit in round 3 or 4 of the window; a window already past that: two or four opportunities from here
        ahead = [3 - had, 4 - had] if had <= 2 else [1, 3]
        return sorted(set().union(*[chain(k, c, j) for j in ahead]))

    @staticmethod
    def first_request(state, n, H):
        """the first request run_lazy(plan_ahead=True) makes for a batch of n frames that starts in `state` ([] when the
        whole batch lies inside the hold-off)"""
        c, hold, run, opened = state
        k = min(hold, n)
        return [] if k >= n else BoardFold.plan_request(n, H, k, c + k, run, opened)

    def run_lazy(self, n, fetch, chunk=8, plan_ahead=True):
        """The same fold over a batch of n frames whose board records do not exist yet: `fetch(indices)` computes the
        records of those frames (-> BOARD_DTYPE array, lines (len(indices), cap, 2)) and is only asked for frames this
        fold is going to look at.  During the hold-off the reference does not run K1..K6 at all (bf_auto.py:43-49); this
        is that, batch-wise.  Same calls to `step` in the same order as `run` over the full records, hence the same
        corners: what is computed never changes what is folded (the cache is keyed by frame).

        A window that opens on running count c can only close on a multiple of 4 of that count -- the library looks for
        corners only there (bf_auto.py:85-94): on its first opportunity, (-c) % 4 + 1 frames in, or r rounds of four frames
        later.  Where window j + 1 opens depends on where window j's hit fell, so fetching window by window is a chain of
        dependent GPU round trips (4.6 per 256-frame batch, 2 ms each next to the classifier: as long as the eager step),
        and fetching every place a window MIGHT be computes most of the records (round 3: 60 %).

        plan_ahead (round 4): two detections in three come on the first opportunity and nearly all others three or four
        rounds later (bench film: r = 0 for 57 of 86, r >= 3 for 28).  So a request covers a HYPOTHESIS for the rest of
        the batch: when a window opens and its first frame is not there, the frames up to the first opportunity of this
        window and of every later one, each placed as if its predecessor hits at once; when a first opportunity has
        passed without a hit, the window's frames up to round 4 plus the same chain of later windows placed after a
        hit in round 3 and after one in round 4.  A batch then costs one request plus one per deviation (~2.5) instead
        of one per window and follow-up, for ~20 % of the records.

        Without it: one request per window up to its probable end (the typical r of the last detections plus `generosity`
        rounds), further requests of `chunk` frames for a later hit."""
        cache = {}

        def load(want):
            want = [f for f in want if 0 <= f < n and f not in cache]
            if not want:
                return
            res, lines = fetch(want)
            self.fetched += len(want)
            self.calls += 1
            for j, f in enumerate(want):
                cache[f] = (int(res["status"][j]), int(res["n_lines"][j]), lines[j])
        k = 0
        while k < n:
            if self.hold > 0:
                skip = min(self.hold, n - k)
                self.hold -= skip
                self.seen += skip
                self.finder.total_f_processed += skip
                k += skip
                continue
            if k not in cache:
                c = self.finder.total_f_processed
                if not plan_ahead:
                    if self._run == 0:                       # a window opens here: up to its probable end in one request
                        rounds = min(max(self.recent) + self.generosity, 4)
                        load(range(k, k + (-c) % 4 + 1 + 4 * rounds))
                    else:
                        load(range(k, k + chunk))
                else:
                    load(self.plan_request(n, self.refresh_frames, k, c, self._run, self._opened))
            status, n_lines, lines = cache[k]
            if self._run == 0:
                self._opened = self.finder.total_f_processed  # the count this window opens on (it may span two batches)
            self.step(dict(status=status, n_lines=n_lines, lines=lines))
            k, self._run = k + 1, self._run + 1
            if self.hold > 0:
                self.episode = self._run                     # frames it took from the end of the hold-off to this hit
                self.recent = (self.recent + [max(0, (self._run - ((-self._opened) % 4 + 1)) // 4)])[-4:]
                self.rounds_seen[self.recent[-1]] = self.rounds_seen.get(self.recent[-1], 0) + 1
                self._run = 0
        return self.mtx


class StonesFold:
    """Ordered replay of SfNeural on per-frame records through the SAME policy object the per-frame finder uses
    (capi.PolicyCore -> ck_policy_run), requests applied to the controller with StonesFinder's sink semantics.
    Python is only entered for frames that emit something."""

    def __init__(self, controller, bg_init_frames=50):
        from .stone.stonesfinder import StoneSink
        self.controller = controller
        self.sink = StoneSink(lambda: self.controller)
        self.policy = capi.PolicyCore(bg_init_frames)
        self.bg_init_frames = bg_init_frames
        self.frames_seen = 0                                 # = SfNeural.total_f_processed
        self.refused = []

    def run(self, region_label, region_conf, fgcount, records=None, order=None):
        """-> per-frame lists of (kind, [(colour, r, c), ...]) requests, in frame order.  `records`: the batch's REC array
        (frame f at records[order[f]] when `order` is given) -- the policy then reads the classifier's answers from the
        records where they lie (ck_policy_run_records)"""
        n = len(region_label) if records is None else (len(records) if order is None else len(order))
        out = [()] * n                     # frames that emit nothing share one empty tuple (no per-frame allocation)

        def apply(kind, moves, k):
            named = [(_SYMBOL[col], r, c) for col, r, c in moves]
            if not out[k]:
                out[k] = []
            out[k].append((kind, named))
            from .core.exceptions import DeletedError
            try:
                if kind == capi.PolicyCore.SUGGEST:
                    self.sink.suggest(*named[0], doprint=False)
                else:
                    self.sink.bulk_update(named)
            except DeletedError as locked:                   # no user, no deletion watch in a batch run: cannot happen
                self.refused.append(locked)
        if records is not None and not (records.dtype == REC and records.flags.c_contiguous):
            records = records if order is None else records[order]
            region_label, region_conf, records, order = records["region_label"], records["region_conf"], None, None
        self.policy.run(self.frames_seen, region_label, region_conf, fgcount, self.sink.board_codes, apply, records=records, order=order)
        self.frames_seen += n
        return out


def learning_rates(first, n, bg_init_frames):
    """the background model's rate for stones-frames first .. first + n - 1 (stonesfinder.py:171-176)"""
    return np.where(first + np.arange(n) < bg_init_frames, 0.01, 0.005)


class GpuCore:
    """The per-shard GPU work of one batch, spread over `lanes` = pairs of (board context, stones context).

    Every context is a HIP stream with its own host thread (a context is single-threaded by contract), so the board
    path (K1..K6) and the stones path (K8, K10..K12) of every lane are in flight together, like the reference's two
    finder threads.  The background model (K9) is ONE stream of frames in order: the lanes warp their frames into
    slices of one goban tensor, and a dedicated context runs the model over it batch after batch (world == 1); with
    world > 1 the goban tensor is handed back for the pixel-sharded exchange instead."""

    def __init__(self, lanes, bg_ctx=None, local_model=True, records_device=None):
        """records_device: where the shard's record buffer lives -- a torch device (HBM: the records go through RCCL from
        there) or None (host memory: one rank, or collectives on host buffers)"""
        self.records_device = records_device
        # A context belongs to ONE thread (the library refuses a second one: CK_ERR_STATE).  A lane given without a board
        # context, or a core given without a model context, gets one of its own when the stones context is a real
        # capi.Context; with stand-in contexts (tests) the orphan work shares the stones context AND its thread.
        self.lanes, self.pools = [], []
        self.device = dev = next((_device_of(c) for pair in lanes for c in pair if _device_of(c) is not None), None)
        ThreadPoolExecutor = lambda k: _pool(k, dev)         # noqa: E731  every worker thread starts on this GPU
        for b, s in lanes:
            ps = ThreadPoolExecutor(1)
            if b is None and isinstance(s, capi.Context):
                b = capi.Context(s.device)
            if b is None or b is s:
                self.lanes.append((s, s))
                self.pools.append((ps, ps))
            else:
                self.lanes.append((b, s))
                self.pools.append((ThreadPoolExecutor(1), ps))
        first = self.lanes[0][1]
        if bg_ctx is None and local_model and isinstance(first, capi.Context):
            bg_ctx = capi.Context(first.device)
        if bg_ctx is None:
            bg_ctx = first
        self.bg_ctx, self.bg_pool = bg_ctx, None
        for (b, s), (pb, ps) in zip(self.lanes, self.pools):          # the model's context is a lane's: use that lane's thread
            if bg_ctx is b:
                self.bg_pool = pb
            elif bg_ctx is s:
                self.bg_pool = ps
        if self.bg_pool is None:
            self.bg_pool = ThreadPoolExecutor(1)             # FIFO: batches go through the model in submission order
        self.local_model, self._handle = local_model, None
        import threading
        self._turn, self._issued, self._served = threading.Condition(), 0, 0

    def close(self, wait=True):
        """stop the lanes' threads (idempotent; contexts stay the caller's).  By default this returns only when every
        queued call has left its context: the caller may close the contexts right after."""
        pools = {id(p): p for pair in self.pools for p in pair}
        pools[id(self.bg_pool)] = self.bg_pool
        for p in pools.values():
            p.shutdown(wait=wait)

    def ticket(self):
        """call in batch order (FastFilePipeline.submit does): the background model sees the batches in that order even
        when two of them are inside __call__ at once"""
        with self._turn:
            self._issued += 1
            return self._issued - 1

    @staticmethod
    def _host(t):
        return t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)

    def _cuts(self, n):
        k = len(self.lanes)
        return [round(i * n / k) for i in range(k + 1)]

    def _in_order(self, seq, fn):
        if seq is None:
            return fn()
        with self._turn:
            while self._served != seq:
                self._turn.wait()
            try:
                return fn()
            finally:
                self._served += 1
                self._turn.notify_all()

    def __call__(self, frames, mtx, rates, seq=None, board=True):
        turn = {"open": seq is not None}

        def take_turn(fn):
            turn["open"] = False
            return self._in_order(seq, fn)
        try:
            return self._batch(frames, mtx, rates, take_turn, board)
        finally:
            if turn["open"]:                                 # failed (or had nothing for the model) before its turn:
                self._in_order(seq, lambda: None)            # the batches behind must not wait for it forever

    def _batch(self, frames, mtx, rates, take_turn, want_board=True):
        """-> (the shard's record buffer (shard_buffer: header row, n records written in place by the library, a spare row),
        foreground counts (n, 19, 19) or None, goban images or None)"""
        n = len(frames)
        recs = shard_buffer(n, self.records_device)
        if n == 0:
            return recs, None, None
        cuts = self._cuts(n)
        parts = [frames[cuts[i]:cuts[i + 1]] for i in range(len(self.lanes))]
        board_f = [pb.submit(board_into, cb, fr, recs[1 + cuts[i]:1 + cuts[i + 1]]) if len(fr) and want_board else None
                   for i, ((pb, _), (cb, _), fr) in enumerate(zip(self.pools, self.lanes, parts))]
        fg, gobans = None, None
        if mtx is not None:
            on_gpu = hasattr(frames, "is_cuda") and frames.is_cuda
            if on_gpu:
                import torch
                gobans = torch.empty((n, 380, 380, 3), dtype=torch.uint8, device=frames.device)
            else:
                gobans = np.empty((n, 380, 380, 3), np.uint8)

            def stones(cs, fr, lo, hi):
                view = gobans[lo:hi]
                cs.warp_perspective(fr, mtx, out=view)
                return view

            def classify(cs, view, lo, hi):
                regions_into(cs, view, recs[1 + lo:1 + hi])
            warp_f = [ps.submit(stones, cs, fr, cuts[i], cuts[i + 1]) if len(fr) else None
                      for i, ((_, ps), (_, cs), fr) in enumerate(zip(self.pools, self.lanes, parts))]
            views = [f.result() if f is not None else None for f in warp_f]
            cnn_f = [ps.submit(classify, cs, v, cuts[i], cuts[i + 1]) if v is not None else None
                     for i, ((_, ps), (_, cs), v) in enumerate(zip(self.pools, self.lanes, views))]
            if self.local_model:
                fg_f = take_turn(lambda: self.bg_pool.submit(self._model_run, gobans, rates))
            for f in cnn_f:
                if f is not None:
                    f.result()
            if self.local_model:
                fg, gobans = self._host(fg_f.result()), None
        for f in board_f:
            if f is not None:
                f.result()
        return recs, fg, gobans

    def _model_run(self, gobans, rates):
        if self._handle is None:
            self._handle = self.bg_ctx.mog2_create(380, 380)
        return self.bg_ctx.mog2_band_run(self._handle, gobans, rates, last_band=True)


class _BandExchangeBroken(RuntimeError):
    """this rank could not take part in the band all-to-all at all (not even with blank bands)"""


class Gathered:
    """the records of a whole batch on rank 0 as the gather left them: `rows` (a REC array: rank after rank, each rank's
    header row first) and `order` (int32: frame f of the batch is rows[order[f]]).  The folds read them in place."""
    __slots__ = ("rows", "order")

    def __init__(self, rows, order):
        self.rows, self.order = rows, np.ascontiguousarray(order, np.int32)

    def __len__(self):
        return len(self.order)

    def in_frame_order(self):
        return self.rows[self.order]


class _Ticket:
    """one batch on its way through the stages: GPU core -> exchange (records, transform, bands, counts) -> stones fold"""
    __slots__ = ("core", "exchange", "mtx", "rates", "n_total", "frames", "have_mtx", "lazy_fold", "keep", "landing")

    def __init__(self, core, mtx, rates, n_total, frames):
        self.core, self.exchange, self.lazy_fold = core, None, None
        self.keep = []                                        # tensors queued work still reads (index tensors of _take): until finish()
        self.landing = []                                     # pinned host buffers the batch's gathers landed in: back to the pool at finish()
        self.mtx, self.rates, self.n_total, self.frames = mtx, rates, n_total, frames
        self.have_mtx = mtx is not None


class FastFilePipeline:
    """process_batch(my_frames, n_total): this rank's shard of a batch -> (on rank 0) the requests the fold emitted.

    A batch goes through three stages, each on its own thread, so that batch k's stones fold overlaps batch k + 1's
    exchange, which overlaps batch k + 2's GPU core (submit() / finish() keep two batches in flight):

      1. GPU core (`compute`, default GpuCore over the given contexts): board path + stones path of this rank's frames.
      2. exchange, ONE thread per rank that issues every collective, batch after batch in submission order, so all
         ranks issue the same collectives in the same order: records packed -> all-gather -> (rank 0) the ordered board
         fold, which only needs the records -> the transform broadcast -> all-to-all of goban bands -> this rank's band
         of the background model through the whole batch in frame order, on the model's OWN context (`ctx_bg`, never
         a lane's) -> all-gather of the foreground counts.
      3. stones fold on rank 0 (finish(), the caller's thread): the library's ordered policy over records + counts.

    The transform a batch is warped with is the one published by the last finish() before its submit(), whatever the
    threads' timing: same results at any world size and any overlap.

    `compute(frames, mtx, learning_rates)` returns (board, region_label, region_conf, fgcount or None, gobans or None);
    `gobans` (n x 380 x 380 x 3) is only handed back when world > 1, for the pixel-sharded background model."""

    def __init__(self, h, w, controller, ctx=None, rank=0, world=1, device=None, compute=None, ctx_board=None,
                 bg_init_frames=50, band_model=None, lanes=None, ctx_bg=None, board_lazy=False, force_exchange=False,
                 records_in_hbm=None):
        self.h, self.w = h, w
        self.rank, self.world = rank, world
        self.group = _Group(rank, world, device)
        self.ctx = ctx if ctx is not None else (lanes[0][1] if lanes else None)      # frame source helper (process_y4m)
        self.ctx_bg = ctx_bg
        # force_exchange: run the whole exchange stage (record gather, broadcast, band all-to-all, counts gather) even with
        # one rank -- a process group of one must exist -- so that the collective code path can be exercised on a box with
        # a single GPU (tests/test_gpu_multirank.py does, over RCCL)
        self.exchange = world > 1 or bool(force_exchange)
        self._owns_compute = compute is None
        # the records go through RCCL from HBM: the library writes them there (GpuCore) and nothing is staged on the host;
        # with one rank, or collectives on host buffers (gloo), they are written in host memory and used from there
        # records_in_hbm: a torch device to force the HBM form whatever the collectives' buffers are (rehearsal on one GPU:
        # two processes over gloo still have the library write their records in HBM; they cross to the host for the wire)
        self.records_device = None
        if records_in_hbm is not None and records_in_hbm is not False:
            self.records_device = records_in_hbm
        elif self.exchange and device is not None and str(device).startswith("cuda"):
            self.records_device = device
        if compute is None:
            compute = GpuCore(lanes or [(ctx_board, ctx)], bg_ctx=ctx_bg, local_model=not self.exchange,
                              records_device=self.records_device)
        self.compute = compute
        gpu = getattr(compute, "device", None)
        if gpu is None:
            gpu = next((_device_of(c) for c in (ctx, ctx_board, ctx_bg) if _device_of(c) is not None), None)
        self.gpu = gpu
        self._runner = _pool(2, gpu)                          # two batches may be inside the GPU core at once
        self._comm = _pool(1, gpu)                            # stage 2: every collective of this rank, in batch order
        # The exchange thread's torch work (band slicing, the scatter into `full`, waiting for the collectives) goes on a
        # stream of ITS OWN: the lanes' threads synchronise torch's DEFAULT stream before they hand a tensor to the library
        # (capi._in), so anything this thread queued there -- i.e. a wait for an all-to-all that completes when the SLOWEST
        # rank has joined -- would gate every board / warp / classifier call of the next batch.
        self._xstream = None
        self._board_thread = None                             # hold-off-aware mode: the board folds of the batches, one after the other
        self.board = BoardFold(h, w)
        self.stones = StonesFold(controller, bg_init_frames)
        self.bg_init_frames, self.stone_frames = bg_init_frames, 0      # frames the stones path has been given (every rank counts)
        self.mtx = None                                       # what every rank warps with (rank 0's fold, broadcast)
        self.frames_done = 0
        self.band = band_rows(world)[rank]
        self.band_model = band_model                          # callable(gobans_band (n, rows, 380, 3), rates) -> counts (n, band, 19)
        self.errors = []
        self.host_seconds = dict(pack=0.0, collectives=0.0, band_model=0.0, fold=0.0, fold_board=0.0, fold_stones=0.0,
                                 gather=0.0, bcast=0.0, band_exchange=0.0, counts_gather=0.0, flags=0.0, unpack=0.0)
        # hold-off-aware mode: the GPU core leaves the board path out and the board fold computes, through the lanes' board
        # contexts (on their own threads), only the records it looks at.  One rank: the fold runs on a thread of its own and
        # asks as it goes (_lazy_fold).  Frames dealt across ranks: every rank works out the batch's first request from the
        # fold's state, which travels with the transform broadcast, and computes the planned frames it owns; the records are
        # gathered, rank 0 folds, and every deviation costs one more (broadcast, gather) round (_lazy_board_exchange).
        self._board_state = (0, 0, 0, 0)                      # BoardFold.state() at the start of the next batch, on every rank
        if board_lazy and not hasattr(self.compute, "lanes"):
            raise ValueError("board_lazy needs a GPU core with lanes (their board contexts compute the requested records)")
        self.board_lazy = bool(board_lazy)

    # ---- pixel-sharded background model (world > 1) -------------------------------------------------------
    def _band_counts(self, gobans, n_total, rates):
        """all-to-all of goban bands, this rank's band through the whole batch in frame order -> its counts
        (n_total, band rows, 19) int32, left where the model put them (HBM with real contexts: they are gathered from there).
        Stream-ordered: the scatter into frame order is queued behind the collective on this thread's torch stream and the
        model's context is ordered behind that stream when it takes the tensor (capi.Context._in) -- no host wait."""
        import time
        import torch
        t0 = time.perf_counter()
        bands = band_rows(self.world)
        px = [(20 * a, min(20 * b, 380)) for a, b in bands]
        if gobans is None:
            gobans = torch.zeros((0, 380, 380, 3), dtype=torch.uint8, device=self.group.device or "cpu")
        if not hasattr(gobans, "numel"):
            gobans = torch.from_numpy(np.ascontiguousarray(gobans))
        if self.group.device is not None and gobans.device != torch.device(self.group.device):
            gobans = gobans.to(self.group.device)                 # gloo rehearsal: collectives on host buffers
        lo, hi = px[self.rank]
        expect = [len(shard_indices(n_total, src, self.world)) * (hi - lo) * 380 * 3 for src in range(self.world)]
        n_mine = len(shard_indices(n_total, self.rank, self.world))
        late = None
        try:
            if len(gobans) != n_mine:
                raise RuntimeError("the GPU core handed back %d goban images for a shard of %d frames" % (len(gobans), n_mine))
            if gobans.is_cuda and torch.cuda.current_stream(gobans.device) != torch.cuda.default_stream(gobans.device):
                gobans.record_stream(torch.cuda.current_stream(gobans.device))
            send = [gobans[:, a:b] for a, b in px]                 # strided views: copied once, into the send buffer
        except Exception as why:                                  # the peers are about to wait in the all-to-all: join it
            late = why                                            # with blank bands of the right sizes, fail afterwards
            try:
                send = [torch.zeros((n_mine, b - a, 380, 3), dtype=torch.uint8, device=self.group.device or "cpu") for a, b in px]
            except Exception as worse:
                raise _BandExchangeBroken("%s (and no memory for blank bands: %s)" % (why, worse))
        parts = self.group.all_to_all_bands(send, expect, bool(os.environ.get("CK_BAND_SELF_THROUGH_COLLECTIVE")))
        if late is not None:
            raise late
        if self.world == 1:
            full = parts[0].reshape(n_total, hi - lo, 380, 3)     # one sender: already in frame order
        else:
            full = torch.empty((n_total, hi - lo, 380, 3), dtype=torch.uint8, device=parts[0].device)
            for src, part in enumerate(parts):                    # frame f of the batch came from rank f mod world
                full[src::self.world] = part.reshape(-1, hi - lo, 380, 3)
        t1 = time.perf_counter()
        counts = self._band_model()(full, rates)
        t2 = time.perf_counter()
        self.host_seconds["band_exchange"] += t1 - t0
        self.host_seconds["band_model"] += t2 - t1
        rows = bands[self.rank][1] - bands[self.rank][0]
        if hasattr(counts, "reshape") and hasattr(counts, "is_cuda"):
            return counts.reshape(n_total, rows, gsize).to(torch.int32)
        return np.asarray(counts).astype(np.int32).reshape(n_total, rows, gsize)

    def _band_model(self):
        """this rank's band of the background model on ITS OWN context: a lane's context belongs to that lane's thread,
        which is inside the next batch while this one is being exchanged"""
        if self.band_model is None:
            if self.ctx_bg is None:
                dev = getattr(self.ctx, "device", None)
                if dev is None:
                    raise RuntimeError("the pixel-sharded background model needs a context of its own (ctx_bg=...)")
                self.ctx_bg = capi.Context(dev, priority=1)       # part of the exchange stage's chain: see _exchange
            a, b = self.band
            bg = self.ctx_bg
            handle = bg.mog2_create(min(20 * b, 380) - 20 * a, 380)
            last = b == gsize

            def run(band_gobans, rates):
                return bg.mog2_band_run(handle, band_gobans, rates, last_band=last)
            self.band_model = run
        return self.band_model

    # ---- one batch --------------------------------------------------------------------------------------
    def _as_shard(self, out, n_mine):
        """what a GPU core returns -> (record buffer, counts, gobans).  The built-in core hands over the buffer the library
        wrote; a `compute` of the old form (board, region_label, region_conf, counts, gobans) -- the stand-ins of the CPU
        tests -- is packed here"""
        if len(out) == 3:
            return out
        import time
        t0 = time.perf_counter()
        board, rl, rc, fg, gobans = out
        recs = shard_buffer(n_mine)
        packed = pack_records(board, rl, rc)
        if len(packed) != n_mine:
            raise RuntimeError("the GPU core handed back %d records for a shard of %d frames" % (len(packed), n_mine))
        recs[1:1 + n_mine] = packed
        self.host_seconds["pack"] += time.perf_counter() - t0
        return recs, fg, gobans

    def _guarded(self, frames, mtx, rates, n_mine, seq):
        try:
            if self.board_lazy:
                return self._as_shard(self.compute(frames, mtx, rates, seq, board=False), n_mine), None
            if seq is not None:
                return self._as_shard(self.compute(frames, mtx, rates, seq), n_mine), None
            return self._as_shard(self.compute(frames, mtx, rates), n_mine), None
        except Exception as why:                              # never leave the other ranks alone in a collective
            return (shard_buffer(n_mine), None, None), why    # (host memory: the GPU may be what failed)

    def submit(self, my_frames, n_total):
        """start a batch (my_frames: frames rank, rank + world, ... of it) -> ticket for finish()"""
        mtx = self.mtx
        rates = learning_rates(self.stone_frames, n_total, self.bg_init_frames) if mtx is not None else np.zeros(n_total)
        if mtx is not None:
            self.stone_frames += n_total
        mine = shard_indices(n_total, self.rank, self.world)
        rates_for_core = rates if not self.exchange else rates[mine]
        seq = self.compute.ticket() if hasattr(self.compute, "ticket") else None
        t = _Ticket(self._runner.submit(self._guarded, my_frames, mtx, rates_for_core, len(mine), seq), mtx, rates, n_total, my_frames)
        if self.board_lazy and self.rank == 0 and not self.exchange:
            # hold-off-aware mode: the board fold needs nothing of the GPU core's results -- it asks the lanes' BOARD contexts
            # (idle in this mode) for the few records it looks at, a chain of small round trips that depends only on the
            # fold before it.  It runs on a thread of its own, batch after batch, as far ahead of the cores as the batches
            # in flight allow.
            if self._board_thread is None:
                self._board_thread = _pool(1, self.gpu)
            t.lazy_fold = self._board_thread.submit(self._lazy_fold, t)
        t.exchange = self._comm.submit(self._exchange, t)
        return t

    def _lazy_fold(self, t):
        import time
        t0 = time.perf_counter()
        try:
            self._fold_board(np.zeros(t.n_total, REC), t.frames, t.keep)
            return self.board.mtx, None
        except Exception as why:
            return None, why
        finally:
            self.host_seconds["fold_board"] += time.perf_counter() - t0

    def _exchange(self, t):
        """stage 2 on this rank's exchange thread, with that thread's own torch stream current (see __init__)"""
        import os
        if self.gpu is None or not self.exchange or os.environ.get("CK_EXCHANGE_DEFAULT_STREAM"):   # (developer A/B knob)
            return self._exchange_on_stream(t)
        import torch
        if not torch.cuda.is_available():
            return self._exchange_on_stream(t)
        if self._xstream is None:
            # high priority: the stage's small copies and the collectives' kernels are a dependent chain of short
            # pieces between the lanes' millisecond launches -- on a normal stream each piece can sit behind one
            # (streams share hardware queues), and the chain, not the lanes, ends up setting the step time
            prio = 0 if os.environ.get("CK_EXCHANGE_PRIORITY") == "0" else -1       # (developer A/B knob)
            self._xstream = torch.cuda.Stream(device=self.gpu, priority=prio)
        with torch.cuda.stream(self._xstream):
            try:
                return self._exchange_on_stream(t)
            finally:
                self._xstream.synchronize()                   # nothing of this batch is left queued behind the thread

    def _wire_rows(self, recs, rows, failed):
        """the first `rows` rows of a shard's record buffer = this rank's contribution to the gather, its header row flagged
        when this rank's GPU core failed"""
        if failed:
            if isinstance(recs, np.ndarray):
                recs[0]["flags"] = FLAG_FAILED
            else:
                recs[0, FLAGS_AT] = FLAG_FAILED               # (little-endian int32, the flag fits its first byte)
        return recs[:rows]

    def _fold_gap(self, n):
        """rank 0, a batch whose records never arrived: the board fold advances over its frames as frames without a contour
        (no detection can come of them), so the running count stays the film's"""
        blank = np.zeros(n, REC)
        blank["status"] = capi.CK_BOARD_NO_CONTOUR
        try:
            self.board.run(blank)
        except Exception as why:
            self.errors.append(why)

    def _exchange_on_stream(self, t):
        """-> (records of the whole batch in frame order -- rank 0 only, None elsewhere --, foreground counts (rank 0) or None,
        transform after this batch, failure seen by any rank).

        Collectives of a batch, the same on every rank in the same order: gather of the record buffers to rank 0 ->
        broadcast of the wire (outcome, transform, fold state: 16 doubles) -> all-to-all of goban bands -> one-word
        all-reduce (did a band of the background model fail?) -> gather of the counts to rank 0.  Ranks != 0 bring the
        wire and the flag word to the host, nothing else."""
        import time
        hs = self.host_seconds
        lazy_x = None
        if self.board_lazy and self.exchange:
            # hold-off-aware board path across ranks: its rounds and the transform broadcast come FIRST -- they need nothing
            # of this batch's GPU core (the board contexts are idle in this mode), so they run under the stones path
            lazy_x = self._lazy_board_exchange(t)
        (recs, fg, gobans), failure = t.core.result()
        lazy_fold = t.lazy_fold.result() if t.lazy_fold is not None else None
        if failure is not None:
            self.errors.append(failure)
        if lazy_x is not None and lazy_x[1]:                   # the board path failed somewhere: every rank leaves here,
            return None, None, self.mtx, True                  # before the records gather
        n_total, W = t.n_total, self.world
        if not self.exchange:
            if failure is not None:
                if lazy_fold is None:
                    self._fold_gap(n_total)
                return None, None, self.board.mtx, True
            full = records_view(recs)[1:1 + n_total]
            t3 = time.perf_counter()
            if lazy_fold is not None:
                new, fold_error = lazy_fold
                if fold_error is not None:
                    raise fold_error
            else:
                self._fold_board(full, None)
                new = self.board.mtx
            hs["fold_board"] += time.perf_counter() - t3
            return full, fg, new, False
        # ---- records to rank 0
        t1 = time.perf_counter()
        per = (n_total + W - 1) // W
        got = self.group.gather_rows(self._wire_rows(recs, per + 1, failure is not None), t.landing)
        t2 = time.perf_counter()
        hs["gather"] += t2 - t1
        wire, full, fold_error = np.zeros(self.WIRE), None, None
        if self.rank == 0:
            bad = [int(r) for r in np.nonzero(got[:, 0, FLAGS_AT] & FLAG_FAILED)[0]]
            if bad:
                wire[0] = -2.0
                if failure is None:
                    self.errors.append(RuntimeError("the GPU core failed on rank(s) %s" % bad))
                if lazy_x is None:
                    self._fold_gap(n_total)
            else:
                # frame f of the batch is row 1 + f // W of rank f mod W: the folds read the records where the gather left them
                f_all = np.arange(n_total, dtype=np.int32)
                full = Gathered(got.view(REC).reshape(W * (per + 1)), (f_all % W) * (per + 1) + 1 + f_all // W)
                t3 = time.perf_counter()
                hs["unpack"] += t3 - t2
                # ordered board fold: it needs the records only, so the transform is known -- and on its way to the other
                # ranks -- before the background model's exchange starts
                new = None
                if lazy_x is not None:
                    new = lazy_x[0]                            # folded (and broadcast) already
                else:
                    try:
                        self._fold_board(full, None)
                        new = self.board.mtx
                    except Exception as why:                   # e.g. the IndexError the reference raises on a 3-vertex hull
                        fold_error = why                       # the other ranks are about to wait in the broadcast: tell them
                        self.errors.append(why)
                hs["fold_board"] += time.perf_counter() - t3
                if fold_error is not None:
                    wire[0] = -1.0
                elif new is not None:
                    wire[0], wire[1:10] = 1.0, np.asarray(new, np.float64).reshape(9)
            wire[10:14] = self.board.state()
        t4 = time.perf_counter()
        wire = self.group.broadcast_array(wire, 0)
        hs["bcast"] += time.perf_counter() - t4
        if wire[0] < 0:                                        # every rank leaves the batch here, before the band exchange
            if self.rank != 0:                                 # whose sizes a failed rank could not honour
                self.errors.append(RuntimeError("the GPU core of a rank failed" if wire[0] == -2.0 else "the board fold failed on rank 0"))
            return full, None, self.mtx, True
        new = wire[1:10].reshape(3, 3).copy() if wire[0] == 1.0 else None
        counts = None
        if t.have_mtx:
            # A rank whose band model fails (a library error, out of memory for the band tensor ...) must not leave the
            # others waiting: it has joined the all-to-all (with blank bands if need be), the flag word says so to
            # everyone, and every rank leaves the batch together, before the counts gather.
            band_error, mine_counts = None, None
            try:
                mine_counts = self._band_counts(gobans, n_total, t.rates)                # (n_total, rows, 19) int32
            except _BandExchangeBroken:
                raise                                         # could not even join the all-to-all: nothing left to keep in step
            except Exception as why:
                band_error = why
            # the flag word and the counts are queued one behind the other and read with ONE host wait (rank 0: the counts'
            # copy to the host; the others: the flag); a rank whose band failed contributes zeros nobody will read
            import torch
            t5 = time.perf_counter()
            flag = self.group.max_flag_start(self.rank + 1 if band_error is not None else 0)
            t6 = time.perf_counter()
            hs["flags"] += t6 - t5
            bands = band_rows(W)
            widest = max(b - a for a, b in bands)
            if mine_counts is None:
                mine_counts = np.zeros((n_total, bands[self.rank][1] - bands[self.rank][0], gsize), np.int32)
            src = self.group.on_wire(mine_counts).reshape(-1)
            slab = src
            if src.numel() != n_total * widest * gsize:        # a narrower band: padded to the widest (equal contributions)
                slab = torch.zeros(n_total * widest * gsize, dtype=torch.int32, device=src.device)
                slab[:src.numel()] = src
            got = self.group.gather_rows(slab.reshape(1, -1), t.landing)
            t7 = time.perf_counter()
            hs["counts_gather"] += t7 - t6
            bad = self.group.max_flag_read(flag)
            hs["flags"] += time.perf_counter() - t7
            if bad:
                self.errors.append(band_error if band_error is not None
                                   else RuntimeError("the background model's band failed on rank %d" % (bad - 1)))
                return full, None, self.mtx, True
            if self.rank == 0:
                got = got.view(np.int32).reshape(W, -1)
                counts = np.concatenate([got[r, :n_total * (b - a) * gsize].reshape(n_total, b - a, gsize)
                                         for r, (a, b) in enumerate(bands)], 1)
        return full, counts, new, False

    # ---- hold-off-aware board path with the frames dealt across ranks ------------------------------------------------
    WIRE = 16                                                 # doubles in front of a request's frame list: kind, 9 transform / count, 4 state

    def _detect_frames(self, frames, idx, rows, keep=None):
        """K1-K6 of frames[idx] on the lanes' board contexts, each driven from its own lane thread (a context is
        single-threaded) -> the board half of the record rows `rows` (len(idx) of them, host memory or HBM), in idx order"""
        import time
        lanes, pools = self.compute.lanes, self.compute.pools

        def timed_detect(ctx, part, producer, out):
            t0 = time.perf_counter()
            if producer is not None and hasattr(ctx, "wait_stream"):
                ctx.wait_stream(producer)                     # the gather of `part` was queued on another thread's stream
            board_into(ctx, part, out)
            return time.perf_counter() - t0
        t0 = time.perf_counter()
        idx = list(idx)
        k = len(lanes) if len(idx) >= 8 * len(lanes) else 1
        cuts = [round(i * len(idx) / k) for i in range(k + 1)]
        futs = [pools[i][0].submit(timed_detect, lanes[i][0], *_take(frames, idx[cuts[i]:cuts[i + 1]], keep), rows[cuts[i]:cuts[i + 1]])
                for i in range(k)]
        got = [f.result() for f in futs]
        hs = self.host_seconds                               # diagnostics: the requests' wall time and the library calls inside
        hs["lazy_fetch"] = hs.get("lazy_fetch", 0.0) + time.perf_counter() - t0
        hs["lazy_detect"] = hs.get("lazy_detect", 0.0) + max(got)
        return rows

    def _board_round(self, t, idx):
        """One round of the hold-off-aware board path across ranks: frame f of the batch lives on rank f mod world (at
        index f // world of its shard); this rank computes K1-K6 of the requested frames it owns straight into a record
        buffer, ONE gather brings the buffers to rank 0 -> there (rows in idx order, their lines), None elsewhere.  A rank
        whose board path raises still joins the gather, with a header row that says so: rank 0 then raises, i.e. ends the
        fold and tells everyone."""
        W, r = self.world, self.rank
        per = max(sum(1 for f in idx if f % W == q) for q in range(W))
        mine = [f // W for f in idx if f % W == r]
        buf, failed = record_buffer(per + 1, self.records_device), False
        try:
            if mine:
                self._detect_frames(t.frames, mine, buf[1:1 + len(mine)], t.keep)
        except Exception as why:                               # the peers are about to wait in the gather: join it
            failed = True
            self.errors.append(why)
            buf = record_buffer(per + 1)                       # (host memory: the GPU may be what failed)
        lease = []
        got = self.group.gather_rows(self._wire_rows(buf, per + 1, failed), lease)
        if r != 0:
            return None
        try:
            got = got.view(REC).reshape(W, per + 1)
            bad = [int(q) for q in np.nonzero(got[:, 0]["flags"] & FLAG_FAILED)[0]]
            if bad:
                raise RuntimeError("the board path (K1-K6) failed on rank(s) %s: %s" % (bad, self.errors[-1:] if 0 in bad else "see their logs"))
            out, at = np.zeros(len(idx), REC), [1] * W
            for j, f in enumerate(idx):
                out[j] = got[f % W, at[f % W]]
                at[f % W] += 1
            return out, out["lines"]
        finally:
            self.group.release(lease)                          # (copied out: the landing buffer is free again)

    def _lazy_board_exchange(self, t):
        """The board fold of one batch in hold-off-aware mode with an exchange stage (reference: no K1-K6 during the hold-off
        after a hit, board/bf_auto.py:43-49).  Every rank knows the fold's state at the start of the batch (it came with
        the previous batch's transform) and therefore the batch's FIRST request -- the frames the fold looks at if every
        window hits on its first opportunity (BoardFold.first_request): each rank computes the ones it owns, one all-gather.
        Rank 0 folds; whenever the fold needs a frame outside what it has, it broadcasts the next request (a hypothesis for
        the rest of the batch again) and another round follows.  The closing broadcast carries the transform and the state
        the next batch starts from.  Every rank issues the same collectives in the same order: round 0 (unless the whole
        batch lies in the hold-off), then (broadcast, gather) until a broadcast is not a request.
        -> (transform after this batch or None, failed)"""
        import time
        t0 = time.perf_counter()
        n, H, WIRE = t.n_total, self.board.refresh_frames, self.WIRE
        first = BoardFold.first_request(self._board_state, n, H)
        wire = np.zeros(WIRE + n)
        if self.rank == 0:
            rounds, err, new = [0], None, None

            def fetch(idx):
                idx = [int(f) for f in idx]
                if rounds[0] == 0:
                    if idx != first:                           # (the same pure function on the same state: cannot happen)
                        raise RuntimeError("the fold's first request differs from the plan every rank made from its state")
                else:
                    req = np.zeros(WIRE + n)
                    req[0], req[1], req[WIRE:WIRE + len(idx)] = 2.0, len(idx), idx
                    self.group.broadcast_array(req, 0)
                rounds[0] += 1
                return self._board_round(t, idx)
            count0 = self.board.finder.total_f_processed
            try:
                if self.board.state() != self._board_state:
                    raise RuntimeError("the board fold's state %s is not the one this batch was planned from %s" % (self.board.state(), self._board_state))
                self.board.run_lazy(n, fetch, plan_ahead=True)
                new = self.board.mtx
            except Exception as why:
                err = why
                # the fold stopped at some frame k < n: it advances over the rest of the batch as frames without a contour, so
                # that the state broadcast below -- what every rank plans the next batch from -- is that of a film with a
                # gap, not of a film that is n - k frames short
                rest = max(0, n - (self.board.finder.total_f_processed - count0))

                def blank(idx):
                    rows = np.zeros(len(idx), REC)
                    rows["status"] = capi.CK_BOARD_NO_CONTOUR
                    return rows, rows["lines"]
                try:
                    self.board.run_lazy(rest, blank, plan_ahead=True)
                except Exception as again:
                    self.errors.append(again)
            if first and rounds[0] == 0:                       # the fold ended before its first request: the other ranks are
                try:                                           # in round 0's gather already
                    self._board_round(t, first)
                except Exception:
                    pass
            if err is not None:
                self.errors.append(err)
                wire[0] = -1.0
            else:
                wire[0] = 1.0 if new is not None else 0.0
                if new is not None:
                    wire[1:10] = np.asarray(new, np.float64).reshape(9)
            wire[10:14] = self.board.state()                   # also after a failure: the fold then stands at the end of the batch
            wire = self.group.broadcast_array(wire, 0)
        else:
            if first:
                self._board_round(t, first)
            while True:
                wire = self.group.broadcast_array(np.zeros(WIRE + n), 0)
                if wire[0] != 2.0:
                    break
                self._board_round(t, [int(f) for f in wire[WIRE:WIRE + int(wire[1])]])
        self.host_seconds["fold_board"] += time.perf_counter() - t0
        self._board_state = tuple(int(v) for v in wire[10:14])
        if wire[0] < 0:
            if self.rank != 0:
                self.errors.append(RuntimeError("the board fold failed (rank 0, or the board path of a rank it asked)"))
            return None, True
        return (wire[1:10].reshape(3, 3).copy() if wire[0] == 1.0 else None), False

    def finish(self, ticket):
        """wait for the batch's exchange, publish the transform, stones fold (rank 0) -> the fold's per-frame request
        lists on rank 0, None elsewhere"""
        import time
        try:
            try:
                full, counts, new, failed = ticket.exchange.result()
            finally:
                ticket.keep.clear()                            # everything queued for this batch has run
            if failed:
                raise RuntimeError("a rank failed in this batch (GPU core, the board fold on rank 0, or a band of the background "
                                   "model): %s" % (self.errors[-1:] or "see its log"))
            self.mtx = new
            t0 = time.perf_counter()
            emitted = None
            if self.rank == 0:
                if not ticket.have_mtx:
                    emitted = [()] * len(full)
                elif isinstance(full, Gathered):
                    emitted = self.stones.run(None, None, counts, records=full.rows, order=full.order)
                else:
                    emitted = self.stones.run(None, None, counts, records=full)
        finally:
            self.group.release(ticket.landing)                 # nothing reads the gathered records after the folds
        hs = self.host_seconds
        hs["fold_stones"] += time.perf_counter() - t0
        hs["fold"] = hs["fold_board"] + hs["fold_stones"]
        hs["collectives"] = hs["gather"] + hs["bcast"] + hs["band_exchange"] + hs["flags"] + hs["counts_gather"]
        self.frames_done += ticket.n_total
        return emitted

    def process_batch(self, my_frames, n_total):
        return self.finish(self.submit(my_frames, n_total))

    def close(self, wait=True):
        """stop this pipeline's stage threads (and the GPU core's, when it was built here); idempotent.  By default it
        returns only when the batches in flight have left the contexts, so that the caller may close those next."""
        self._runner.shutdown(wait=wait)
        self._comm.shutdown(wait=wait)
        if self._board_thread is not None:
            self._board_thread.shutdown(wait=wait)
        if self._owns_compute and hasattr(self.compute, "close"):
            self.compute.close(wait)

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        # leaving on an exception: do not wait for batches in flight (with a dead peer that is the collective timeout)
        self.close(wait=exc_type is None)

    def _fold_board(self, full, frames=None, keep=None):
        """ordered replay of the board finder on the gathered records of one batch (rank 0); with `frames` (hold-off-aware
        mode) the board records are computed on demand, the wanted frames split over the lanes' board contexts, each
        driven from its own lane thread (a context is single-threaded)"""
        if frames is None:
            return self.board.run(full.rows, full.order) if isinstance(full, Gathered) else self.board.run(full)
        def fetch(idx):
            rows = self._detect_frames(frames, idx, record_buffer(len(idx)), keep)
            return rows, rows["lines"]
        return self.board.run_lazy(len(full), fetch, plan_ahead=os.environ.get("CK_LAZY_PLAN") != "0")      # (developer A/B knob)

    def fold(self, full, counts, have_mtx=True, frames=None):
        """both ordered folds of one batch, one after the other (what finish() does across its stages)"""
        self._fold_board(full, frames)
        if not have_mtx:
            return [()] * len(full)
        return self.stones.run(None, None, counts, records=full)

    def process_y4m(self, capture, batch=256, file_fps=None, torch_device=None):
        """Fast processing of a video file (README "Fast video file processing"; frame selection as the reference's
        file reader, core/vmanager.py:510-525).  The frames to analyse are dealt to the ranks batch by batch; each rank
        uploads ITS frames as I420 (1.5 B/px through a reused pinned buffer), converts them in HBM (ck_i420_to_bgr)
        and the batch goes through process_batch.  Returns the concatenated request lists (rank 0)."""
        import torch
        from .core.capture import file_frame_indices
        idx = file_frame_indices(len(capture), capture.fps, file_fps)
        dev = torch_device if torch_device is not None else torch.device("cuda", getattr(self.ctx, "device", 0))
        pinned, emitted = None, []
        for b0 in range(0, len(idx), batch):
            chunk = idx[b0:b0 + batch]
            mine = [chunk[k] for k in shard_indices(len(chunk), self.rank, self.world)]
            if pinned is None:
                pinned = torch.empty((len(shard_indices(batch, 0, self.world)), capture.fsize), dtype=torch.uint8).pin_memory()
            raw = capture.read_raw_batch(mine, out=pinned.numpy())
            if len(mine):
                frames = self.ctx.i420_to_bgr(raw, capture.h, capture.w, to_device=dev)
            else:
                frames = torch.empty((0, capture.h, capture.w, 3), dtype=torch.uint8, device=dev)
            out = self.process_batch(frames, len(chunk))
            if out is not None:
                emitted.extend(out)
        return emitted
