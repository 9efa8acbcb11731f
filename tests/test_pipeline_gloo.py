"""N > 1 path on CPU: processes on the `gloo` backend.  Frames are dealt f -> rank f mod world, each rank computes
its shard's per-frame records, ONE all_gather moves them, the background model runs sharded by PIXEL (bands of
goban rows exchanged in one all_to_all, each rank's band through the whole batch in frame order), rank 0 folds
and broadcasts the transform.  What rank 0 ends up with must be exactly what a single process gives on the
whole batch -- including batches smaller than the world and frames with more Hough lines than a record holds."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from camkifu_amd import pipeline
from camkifu_amd.controller import ControllerHeadless

H, W = 480, 640
BATCHES = (13, 13, 1, 9, 22)      # odd sizes on purpose; the third is smaller than the world
BG = 4                            # background frames of the stones policy in this test


def _goban(g):
    """goban image of global stones-frame g: a flat board, sensor noise, and a dark block that comes and goes"""
    rng = np.random.default_rng(7000 + g)
    img = np.full((380, 380, 3), (70, 110, 140), np.uint8) + rng.integers(0, 3, (380, 380, 3), dtype=np.uint8)
    k = g % 12
    if 5 <= k < 9:
        img[60:130, 200 + 4 * k:290 + 4 * k] = 20
    if g >= 14:
        img[330:, 330:] = 230                      # something appears on the last row / column for good
    return img


def _fake_compute(state, use_band_model):
    """deterministic stand-in for the GPU core: records depend only on the global frame index"""
    from .stub_ctx import OracleCtx
    ctx = OracleCtx()
    full_model = ctx.mog2_create(380, 380)

    def compute(frames, mtx, rates):
        ids = state["ids"]
        board, rl, rc, fg, gob = [], [], [], [], []
        for k, f in enumerate(ids):
            rng = np.random.default_rng(1000 + int(f))
            # the four sides of a slanted board (so that the fold does find corners) plus a frame-dependent number
            # of near-duplicates, as Hough peaks come in practice; frame 3 carries more lines than a record holds
            sides = np.array([[100, 0.05], [540, 0.08], [60, 1.55], [420, 1.62]], np.float32)
            extra = 70 if f == 3 else int(rng.integers(0, 4))
            dup = sides[rng.integers(0, 4, extra)] + np.stack([rng.integers(-1, 2, extra), np.zeros(extra)], 1).astype(np.float32)
            lines = np.concatenate([sides, dup])
            board.append(dict(status=0, n_contours=1 + int(f), n_lines=len(lines), biggest_area=2e5 + f, lines=lines))
            if mtx is not None:
                g = state["stones_seen"] + int(np.flatnonzero(state["batch_ids"] == f)[0])
                img = _goban(g)
                gob.append(img)
                lab = np.zeros((10, 10), np.uint8)
                if g >= 14:
                    lab[9, 9] = 2 * 27                       # white on (18, 18) once it has appeared
                rl.append(lab)
                rc.append(rng.uniform(0.7, 1.0, (10, 10)))
                if not use_band_model:
                    fg.append(ctx.zone_counts(ctx.mog2_apply(full_model, img, float(rates[k]))))
        n = len(ids)
        if mtx is None:
            return board, np.zeros((n, 10, 10), np.uint8), np.zeros((n, 10, 10)), None, None
        gobs = np.stack(gob) if gob else np.zeros((0, 380, 380, 3), np.uint8)
        return (board, np.stack(rl) if rl else np.zeros((0, 10, 10), np.uint8), np.stack(rc) if rc else np.zeros((0, 10, 10)),
                (np.stack(fg) if fg else None) if not use_band_model else None, gobs if use_band_model else None)
    return compute, ctx


def _drive(rank, world):
    ctrl = ControllerHeadless()
    state = dict(ids=None, batch_ids=None, stones_seen=0)
    compute, octx = _fake_compute(state, use_band_model=world > 1)
    pipe = pipeline.FastFilePipeline(H, W, ctrl, rank=rank, world=world, compute=compute, bg_init_frames=BG)
    if world > 1:
        a, b = pipe.band
        handle = octx.mog2_create(min(20 * b, 380) - 20 * a, 380)
        pipe.band_model = lambda band, rates: octx.mog2_band_run(handle, band.numpy(), rates, last_band=(b == 19))
    pipe.board.refresh_frames = 3
    out, first = [], 0
    for n in BATCHES:
        state["batch_ids"] = first + np.arange(n)
        state["ids"] = state["batch_ids"][pipeline.shard_indices(n, rank, world)]
        had_mtx = pipe.mtx is not None
        emitted = pipe.process_batch(None, n)
        if had_mtx:
            state["stones_seen"] += n
        out.append(emitted)
        first += n
    mtx = None if pipe.mtx is None else pipe.mtx.tolist()
    return out, ctrl.kifu.to_sgf(), mtx, pipe.stones.policy.state()["targets"].tolist()


def _run(rank, world, port, q, env=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.update(env or {})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank,) + _drive(rank, world))
    finally:
        dist.destroy_process_group()


def test_record_roundtrip_and_line_cap():
    rng = np.random.default_rng(1)
    board = [dict(status=0, n_contours=4, n_lines=3, biggest_area=1e5, lines=rng.random((3, 2)).astype(np.float32)),
             dict(status=2, n_contours=1, n_lines=0, biggest_area=7.5, lines=np.zeros((0, 2), np.float32)),
             dict(status=0, n_contours=9, n_lines=90, biggest_area=3e5, lines=rng.random((90, 2)).astype(np.float32))]
    rl = rng.integers(0, 81, (3, 10, 10)).astype(np.uint8)
    rc = rng.random((3, 10, 10))
    rec = pipeline.pack_records(board, rl, rc)
    assert rec.dtype.itemsize == pipeline.REC_BYTES and rec.shape == (3,)
    assert rec["n_lines"].tolist() == [3, 0, 90] and rec["flags"].tolist() == [0, 0, pipeline.FLAG_LINES_CUT]
    assert np.array_equal(rec["lines"][0, :3], board[0]["lines"]) and not rec["lines"][0, 3:].any()
    assert np.array_equal(rec["lines"][2], board[2]["lines"][:pipeline.LMAX])           # the strongest LMAX lines survive
    assert np.array_equal(rec["region_label"], rl) and np.array_equal(rec["region_conf"], rc)
    assert rec["biggest_area"].tolist() == [1e5, 7.5, 3e5]
    # the raw form Context.board_detect(raw=True) hands over packs to the same bytes, stale scratch does not leak
    from camkifu_amd import capi
    res = np.zeros(3, capi.BOARD_DTYPE)
    lines = np.full((3, 128, 2), 7.0, np.float32)
    for f, b in enumerate(board):
        res[f] = (b["status"], b["n_contours"], b["n_lines"], 0, b["biggest_area"])
        lines[f, :b["n_lines"]] = b["lines"]
    assert pipeline.pack_records((res, lines), rl, rc).tobytes() == rec.tobytes()
    assert list(pipeline.shard_indices(7, 1, 3)) == [1, 4]
    assert pipeline.band_rows(8) == [(0, 2), (2, 5), (5, 7), (7, 10), (10, 12), (12, 14), (14, 17), (17, 19)]
    assert pipeline.band_rows(1) == [(0, 19)]


def test_grid_of_regions_matches_the_oracle(ora):
    rng = np.random.default_rng(3)
    y = rng.random((100, 81)).astype(np.float32)
    lab, conf = ora.decode_regions(y)
    grid, cgrid = pipeline.grid_of(lab.reshape(1, 10, 10), conf.reshape(1, 10, 10))
    want_l, want_c = ora.decode_all(y)
    assert np.array_equal(grid[0], want_l) and np.array_equal(cgrid[0], want_c)


@pytest.mark.parametrize("world", [2, 3, 8, -3])
def test_multi_rank_fold_equals_single_process(world):
    """(world -3: three ranks with every rank's own goban band sent through the all-to-all as well, the form a one-GPU box
    uses to exercise the RCCL call -- CK_BAND_SELF_THROUGH_COLLECTIVE; by default the own band is a local copy)"""
    env = {"CK_BAND_SELF_THROUGH_COLLECTIVE": "1"} if world < 0 else None
    world = abs(world)
    ref = _drive(0, 1)
    assert ref[2] is not None                                  # the fold did find the board
    assert any(req for batch in ref[0] for req in batch)       # and the policy emitted something afterwards
    assert "W[ss]" in ref[1] or "W[" in ref[1]

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run, args=(r, world, port, q, env)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    rank0 = res[0]
    assert rank0[1] == ref[0] and rank0[2] == ref[1]           # same requests, same game record
    assert rank0[4] == ref[3]                                  # same policy state (targets) at the end
    for r in res:
        assert np.allclose(np.array(r[3]), np.array(ref[2]))   # every rank holds the broadcast transform
    for r in res[1:]:
        assert all(e is None for e in r[1])                    # only rank 0 folds


# ---------------------------------------------------------------- host cost against the world size
def _timed_rank(rank, world, port, q, per_rank, steps, no_bands=False, film=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import time
        n_total = per_rank * world
        mine = pipeline.shard_indices(n_total, rank, world)
        from camkifu_amd import capi
        if film:
            # records as a filmed game leaves them: Hough lists of a dozen lines around a slanted board (every side in every
            # frame), and a hand over the board 12 frames in every 32 (foreground counts above both agitation thresholds
            # on a 5 x 5 patch of intersections, the new stone still foreground for a few frames afterwards)
            res, lines = _board_table(33, [n_total], 1080, 1920, steady=True)
            res, lines = res[mine], lines[mine]
            fg1 = np.zeros((n_total, 19, 19), np.int32)
            rng = np.random.default_rng(9)
            for f0 in range(52, n_total, 32):
                r, c = rng.integers(2, 17, 2)
                fg1[f0:f0 + 12, r - 2:r + 3, c - 2:c + 3] = rng.integers(150, 400, (min(12, n_total - f0), 5, 5))
                fg1[f0 + 12:f0 + 18, r, c] = 250
        else:
            sides = np.array([[100, 0.05], [540, 0.08], [60, 1.55], [420, 1.62]], np.float32)
            lines = np.zeros((len(mine), pipeline.LMAX, 2), np.float32)
            lines[:, :4] = sides
            res = np.zeros(len(mine), capi.BOARD_DTYPE)
            res["n_lines"] = 4
            fg1 = np.zeros((n_total, 19, 19), np.int32)
        gob = np.zeros((len(mine), 380, 380, 3), np.uint8)
        rl, rc = np.zeros((len(mine), 10, 10), np.uint8), np.full((len(mine), 10, 10), 0.9)
        a, b = pipeline.band_rows(world)[rank]

        def compute(frames, mtx, rates):                    # no GPU, no oracle: the host side is what is being timed
            if mtx is None:
                return (res, lines), rl, rc, None, None
            return (res, lines), rl, rc, (fg1 if world == 1 else None), (gob if world > 1 else None)
        h, w = (1080, 1920) if film else (H, W)
        pipe = pipeline.FastFilePipeline(h, w, ControllerHeadless(), rank=rank, world=world, compute=compute)
        pipe.band_model = lambda band, rates: np.zeros((len(band), b - a, 19), np.int32)
        if no_bands:           # the host's own work is what is timed: 2 048 goban images through gloo on loopback are not
            pipe._band_counts = lambda gobans, n, rates: np.ascontiguousarray(fg1[:, a:b])
        for _ in range(2):
            pipe.process_batch(None, n_total)                # board found, one stones batch
        per_step, walls = [], []
        for _ in range(steps):
            for k in pipe.host_seconds:
                pipe.host_seconds[k] = 0.0
            t0 = time.perf_counter()
            try:
                pipe.process_batch(None, n_total)
            except RuntimeError:                             # (a film may run into the reference's own IndexError: the time still counts)
                pass
            walls.append(1e3 * (time.perf_counter() - t0))
            per_step.append({k: 1e3 * v for k, v in pipe.host_seconds.items()})
        med = {k: float(np.median([p[k] for p in per_step])) for k in per_step[0]}      # medians: a GC pause is not the fold
        q.put((rank, med, float(np.median(walls)), n_total, dict(pipe.group.host_bytes)))
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_rank0_host_cost_per_frame_does_not_grow_with_world(capsys):
    """weak scaling rehearsal on CPU (gloo, fake GPU core, 32 frames per rank and step): rank 0 folds world x 32
    records per step; what it spends per RECORD in pack + fold must not grow with the world (the fold is O(records) in
    C++, Python is only entered for frames that emit), and the fold of a whole step stays in the low milliseconds"""
    per_rank, steps, got = 32, 6, {}
    for world in (1, 2, 8):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_timed_rank, args=(r, world, port, q, per_rank, steps)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=240) for _ in range(world))
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        got[world] = res[0]
    with capsys.disabled():
        for world, (_, phases, wall, n_total, _hb) in got.items():
            print("\n  world %d: %4d records/step  pack %.3f ms  collectives %.3f ms  fold %.3f ms  (step %.2f ms, gloo on CPU)"
                  % (world, n_total, phases["pack"], phases["collectives"], phases["fold"], wall), end="")
    per_rec = {w: (got[w][1]["unpack"] + got[w][1]["fold"]) / got[w][3] for w in got}
    assert per_rec[8] <= 2.0 * per_rec[1] + 0.01, per_rec            # ms per record: flat (generous slack for a busy box)
    assert got[8][1]["fold"] < 25.0                                    # 256 records folded in a few ms


@pytest.mark.parametrize("film", [False, True])
def test_rank0_host_threads_stay_under_the_gpu_step_at_world_8(capsys, film):
    """256 frames per rank at world 8 = 2 048 records per step.  Rank 0's two host stages run on their
    own threads (exchange thread: the gathered records into frame order + the board fold, between the collectives;
    caller's thread: the stones fold) and overlap the GPU core of the following batches, so the BUSIEST of them must stay
    under the step of the fastest classifier mode (bf16: 8 ms).  Plain records (four lines per frame, nothing moves, what
    rounds 2-5 timed): 2.5 ms each.  Records as a filmed game leaves them (a dozen lines per frame, a hand over the board
    12 frames in 32): under the 8 ms step, with room.  gloo's loopback collectives are not RCCL's and are left out; the goban
    bands (887 MB per step) are not moved in this rehearsal.  Ranks != 0 bring nothing to the host but the wire and the
    flag words."""
    world, per_rank, steps = 8, 256, 4
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_timed_rank, args=(r, world, port, q, per_rank, steps, True, film)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=400) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    _, ph, wall, n_total, host_bytes = res[0]
    exchange_thread = ph["unpack"] + ph["fold_board"]
    with capsys.disabled():
        print("\n  world 8, %d %s records/step: exchange thread %.2f ms (records into frame order %.2f + board fold %.2f), stones fold "
              "%.2f ms; gloo collectives %.1f ms (loopback, not RCCL)" % (n_total, "filmed-game" if film else "plain", exchange_thread,
                                                                       ph["unpack"], ph["fold_board"], ph["fold_stones"], ph["collectives"]), end="")
    assert n_total == 2048
    # (the eight ranks of this rehearsal share this box's eight cores with each other: generous against the 1.1 / 0.7 ms
    # (plain) and 4.5 / 1.5 ms (film) an idle box gives)
    limit = 8.0 if film else 2.5
    assert max(exchange_thread, ph["fold_stones"]) < limit, ph
    batches = 2 + steps
    assert host_bytes["gather"] >= batches * n_total * pipeline.REC_BYTES       # rank 0: the records (and the counts) came to its host
    for r in res[1:]:
        hb = r[4]
        assert hb["gather"] == 0, hb                                            # ranks != 0: never
        assert hb["bcast"] <= batches * 8 * pipeline.FastFilePipeline.WIRE and hb["flag"] <= batches * 4, hb


def _failing_rank(rank, world, port, q, bad_rank, n_total, fold_fails=False, band_fails=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from camkifu_amd import capi
        mine = pipeline.shard_indices(n_total, rank, world)
        lines = np.zeros((len(mine), pipeline.LMAX, 2), np.float32)
        lines[:, :4] = np.array([[100, 0.05], [540, 0.08], [60, 1.55], [420, 1.62]], np.float32)
        res = np.zeros(len(mine), capi.BOARD_DTYPE)
        res["n_lines"] = 4
        calls = [0]

        def compute(frames, mtx, rates):
            calls[0] += 1
            if rank == bad_rank and calls[0] == 3 and band_fails is None:
                raise capi.CkError("libck_hip error 2: injected")
            gob = None if mtx is None else np.zeros((len(mine), 380, 380, 3), np.uint8)
            if rank == bad_rank and calls[0] == 3 and band_fails == "before the all-to-all":
                gob = gob[:-1]                                # a shard short of one goban image: the send sizes are wrong
            return (res, lines), np.zeros((len(mine), 10, 10), np.uint8), np.full((len(mine), 10, 10), 0.9), None, gob
        pipe = pipeline.FastFilePipeline(H, W, ControllerHeadless(), rank=rank, world=world, compute=compute)
        a, b = pipe.band
        band_calls = [0]

        def band_model(band, rates):
            band_calls[0] += 1
            if rank == bad_rank and band_calls[0] == 2 and band_fails == "in the model":      # batch 3 = the 2nd with a transform
                raise capi.CkError("libck_hip error 2: injected into the band model")
            return np.zeros((len(band), b - a, 19), np.int32)
        pipe.band_model = band_model
        if fold_fails and rank == 0:                        # the reference's own failure mode: an exception out of _detect
            fold, batches = pipe._fold_board, [0]

            def failing(full, frames=None):
                batches[0] += 1
                if batches[0] == 3:
                    raise IndexError("corner hull has fewer than 4 vertices")
                return fold(full, frames)
            pipe._fold_board = failing
        outcome = []
        for k in range(4):
            try:
                pipe.process_batch(None, n_total)
                outcome.append("ok")
            except RuntimeError as why:
                outcome.append("raised: " + str(why)[:40])
        q.put((rank, outcome, pipe.mtx is not None, int(pipe.board.finder.total_f_processed)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bad_rank, n_total", [(1, 9), (2, 2)])
def test_a_failing_rank_makes_every_rank_raise_instead_of_hanging(bad_rank, n_total):
    """a GPU-core failure on one rank is carried as a flag through the record gather and EVERY rank raises
    after it, before the band exchange whose sizes the failed rank could not honour -- also when the failing rank's
    shard is empty (2 frames dealt to 3 ranks: rank 2 holds none, its header row still says so).  The batches before
    and after go through."""
    world = 3
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_rank, args=(r, world, port, q, bad_rank, n_total)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, outcome, has_mtx, counted in res:
        assert outcome[0] == outcome[1] == outcome[3] == "ok", (rank, outcome)
        assert outcome[2].startswith("raised: a rank failed"), (rank, outcome)
        assert has_mtx
    assert res[0][3] == 4 * n_total                            # the failed batch is a gap in the film, not a shorter film


@pytest.mark.parametrize("where", ["in the model", "before the all-to-all"])
def test_a_band_model_failure_on_one_rank_reaches_every_rank(where):
    """the pixel-sharded background model runs AFTER the record gather's failure flags, inside the exchange
    stage: a rank whose band fails (a library error in ck_mog2_band_run, no memory for the band tensor, a shard of the
    wrong size) still joins the band all-to-all (with blank bands if need be) and the counts gather, whose header row
    says so: every rank raises from finish(), nobody waits for a collective, the next batch goes through."""
    world = 3
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_rank, args=(r, world, port, q, 1, 9, False, where)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, outcome, has_mtx, counted in res:
        assert outcome[0] == outcome[1] == outcome[3] == "ok", (rank, outcome)
        assert outcome[2].startswith("raised: a rank failed"), (rank, outcome)
        assert has_mtx


def test_a_board_fold_exception_on_rank_0_reaches_every_rank():
    """rank 0 folds the board records between the record gather and the transform broadcast, while the other ranks
    already wait in that broadcast: an exception there (the reference raises IndexError on a 3-vertex corner hull,
    bf_auto.py:206) travels in the broadcast and every rank raises from finish() -- nobody is left in a collective"""
    world = 3
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_rank, args=(r, world, port, q, -1, 9, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, outcome, has_mtx, counted in res:
        assert outcome[0] == outcome[1] == outcome[3] == "ok", (rank, outcome)
        assert outcome[2].startswith("raised: a rank failed"), (rank, outcome)


@pytest.mark.parametrize("plan_ahead", [True, False])
def test_lazy_board_fold_equals_the_eager_one(plan_ahead):
    """BoardFold.run_lazy (records computed only for the frames the fold looks at: the reference skips K1..K6 during its
    hold-off) against BoardFold.run over the full records: same corners, same transform, same counters, batch after
    batch -- with steady inputs, with hits that come later than predicted, with frames that show no board.  Both request
    strategies: a hypothesis for the rest of the batch per request (plan_ahead) and window by window."""
    from camkifu_amd import capi
    from camkifu_amd.pipeline import BoardFold, LMAX
    from tests.test_fold_cpu import _hough_like, _sides
    rng = np.random.default_rng(8)
    h, w = 1080, 1920
    sides = _sides(rng, h, w)
    eager, lazy = BoardFold(h, w), BoardFold(h, w)
    asked = []
    for batch in range(8):
        n = int(rng.integers(40, 200))
        res = np.zeros(n, capi.BOARD_DTYPE)
        lines = np.zeros((n, LMAX, 2), np.float32)
        for f in range(n):
            if batch == 3 and f == 10:
                sides = _sides(rng, h, w)                            # the camera is bumped
            blind = batch == 5 and f < 120                           # nothing to see for a while: detection takes long
            ls = _hough_like(rng, sides, h, w, 1)[:LMAX]
            res["status"][f] = 1 if blind else int(rng.choice([0, 0, 0, 0, 2]))
            res["n_lines"][f] = 0 if blind else len(ls)
            lines[f, :len(ls)] = 0 if blind else ls
        recs = [dict(status=int(res["status"][f]), n_lines=int(res["n_lines"][f]), lines=lines[f]) for f in range(n)]
        eager.run(recs)

        def fetch(idx):
            asked.append(len(idx))
            assert len(set(idx)) == len(idx) and min(idx) >= 0 and max(idx) < n
            return res[list(idx)], lines[list(idx)]
        lazy.run_lazy(n, fetch, plan_ahead=plan_ahead)
        a, b = eager.finder, lazy.finder
        assert (eager.mtx is None) == (lazy.mtx is None) and (eager.mtx is None or np.array_equal(eager.mtx, lazy.mtx))
        assert a.corners.hull == b.corners.hull and a.total_f_processed == b.total_f_processed
        assert (eager.hold, eager.seen, eager.looked) == (lazy.hold, lazy.seen, lazy.looked)
    # what is computed stays close to what is looked at (the blind stretch -- 120 frames, all looked at -- included)
    assert eager.mtx is not None and lazy.looked <= lazy.fetched < lazy.looked + 0.15 * lazy.seen
    assert len(asked) < 60        # a request or two per window, plus the blind stretch in chunks of 8



class _TableCtx:
    """stand-in for a capi.Context whose "frames" are their own indices into a table of board records"""

    def __init__(self, res, lines, log):
        self.res, self.lines, self.log = res, lines, log

    def board_detect(self, frames, thresh, cap, raw):
        idx = np.asarray(frames, np.int64)
        self.log.append(len(idx))
        return self.res[idx], self.lines[idx]

    def warp_perspective(self, frames, mtx, out=None):
        return out

    def cnn_regions(self, view):
        return np.zeros((len(view), 10, 10), np.uint8), np.full((len(view), 10, 10), 0.9)

    def mog2_create(self, h, w):
        return 0

    def mog2_band_run(self, handle, gobans, rates, last_band=True):
        return np.zeros((len(gobans), 19, 19), np.int32)


def _board_table(seed, sizes, h, w, bump_at=300, steady=False):
    """board records of a filmed game, one per global frame: a slanted board seen through noisy Hough lists, a camera bump.
    `steady`: every side of the board shows in every frame, as on real footage of a fixed camera (most detections then
    come on a window's first opportunity: the bench film's distribution); otherwise sides drop out and one frame in five
    has no board-sized contour, so that detections take several grouping rounds"""
    from camkifu_amd import capi
    from camkifu_amd.pipeline import LMAX
    from tests.test_fold_cpu import _hough_like, _sides
    rng = np.random.default_rng(seed)
    sides = _sides(rng, h, w)
    total = sum(sizes)
    res = np.zeros(total, capi.BOARD_DTYPE)
    lines = np.zeros((total, LMAX, 2), np.float32)
    for f in range(total):
        if f == bump_at:
            sides = _sides(rng, h, w)                                # the camera is bumped
        ls = _hough_like(rng, sides, h, w, 1)[:LMAX]
        if steady:
            ls = np.concatenate([np.asarray(sides, np.float32).reshape(-1, 2), ls])[:LMAX]
        res["status"][f] = 0 if steady else int(rng.choice([0, 0, 0, 0, 2]))
        res["n_lines"][f] = len(ls)
        lines[f, :len(ls)] = ls
    return res, lines


LAZY_SIZES = [96, 128, 64, 128, 128, 96, 128, 128, 5, 128]            # a batch smaller than the world is in there


def _drive_lazy(rank, world, lazy, fail=None, steady=False):
    """the pipeline over a table of board records, frames dealt to `world` ranks; the GPU core is a real GpuCore over
    stand-in contexts whose "frames" are indices into the table.  -> (transforms after every batch, fold counters, frames
    this rank ran K1-K6 on, what finish() raised per batch)"""
    from camkifu_amd.pipeline import FastFilePipeline, GpuCore, shard_indices
    h, w = 1080, 1920
    res, lines = _board_table(21, LAZY_SIZES, h, w, steady=steady)
    log = []

    class Ctx(_TableCtx):
        def board_detect(self, frames, thresh, cap, raw):
            if fail is not None and fail[0] == rank and len(log) >= fail[1]:
                fail[1] = 1 << 30                                       # once
                raise RuntimeError("board path down on rank %d" % rank)
            return _TableCtx.board_detect(self, frames, thresh, cap, raw)
    lanes = [(Ctx(res, lines, log), Ctx(res, lines, log)) for _ in range(2)]
    core = GpuCore(lanes, bg_ctx=Ctx(res, lines, log), local_model=world == 1)
    pipe = FastFilePipeline(h, w, ControllerHeadless(), rank=rank, world=world, compute=core, board_lazy=lazy)
    assert pipe.board_lazy == lazy
    a, b = pipe.band
    # (the goban bands are not what is under test: 128 blank goban images per batch through gloo on loopback)
    pipe._band_counts = lambda gobans, n, rates: np.zeros((n, b - a, 19), np.int32)
    tickets, mtxs, raised, first = [], [], [], 0

    def finish(tk):
        try:
            pipe.finish(tk)
            raised.append(None)
        except RuntimeError as why:
            raised.append(str(why))
        mtxs.append(None if pipe.mtx is None else pipe.mtx.tolist())
    for n in LAZY_SIZES:
        tickets.append(pipe.submit(first + shard_indices(n, rank, world), n))
        first += n
        if len(tickets) == 2:                                            # two batches in flight
            finish(tickets.pop(0))
    while tickets:
        finish(tickets.pop(0))
    bd = pipe.board
    out = dict(mtxs=mtxs, looked=bd.looked, seen=bd.seen, hold=bd.hold, hull=bd.finder.corners.hull, fetched=bd.fetched, calls=bd.calls,
               asked=int(sum(log)), raised=raised, state=pipe._board_state)
    pipe.close()
    core.close()
    return out


def _run_lazy(rank, world, port, q, lazy, fail, steady):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, _drive_lazy(rank, world, lazy, fail, steady)))
    finally:
        dist.destroy_process_group()


def _spawn_lazy(world, lazy, fail=None, steady=False):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run_lazy, args=(r, world, port, q, lazy, fail, steady)) for r in range(world)]
    for p in procs:
        p.start()
    res = [r[1] for r in sorted((q.get(timeout=240) for _ in range(world)), key=lambda r: r[0])]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world,steady", [(2, False), (8, True), (8, False)])
def test_hold_off_aware_board_path_across_ranks_equals_the_eager_one(world, steady):
    """the hold-off-aware board path (reference: no K1-K6 during the hold-off after a hit,
    bf_auto.py:43-49) with the frames dealt across ranks.  Every rank plans the batch's first request from the fold's
    state, which travels with the transform broadcast, and runs K1-K6 only on the planned frames it owns; rank 0 folds
    the gathered records and asks for more, round by round, when a detection comes later than planned.  Ten batches (one
    smaller than the world), two in flight, a camera bump: on EVERY rank the transform after every batch equals the
    one-process eager pipeline's, rank 0's fold counters too, and the ranks together computed a fraction of the records:
    at most a quarter on footage where the board shows in every frame, looked-at + 15 % of the film where it does not."""
    total = sum(LAZY_SIZES)
    ref = _drive_lazy(0, 1, False, steady=steady)
    assert ref["asked"] == total and any(m is not None for m in ref["mtxs"])
    res = _spawn_lazy(world, True, steady=steady)
    for r in res:
        assert r["mtxs"] == ref["mtxs"] and r["raised"] == [None] * len(LAZY_SIZES)
        assert r["state"] == res[0]["state"]                            # every rank ends with the fold's state
    r0 = res[0]
    assert (r0["looked"], r0["seen"], r0["hold"], r0["hull"]) == (ref["looked"], ref["seen"], ref["hold"], ref["hull"])
    computed = sum(r["asked"] for r in res)
    assert computed == r0["fetched"] and r0["looked"] <= computed <= r0["looked"] + 0.15 * total
    pct = 100.0 * computed / total
    print("world %d, %s film: board_records_computed_pct %.1f (looked at %.1f %%), %d requests for %d batches"
          % (world, "steady" if steady else "hard", pct, 100.0 * r0["looked"] / total, r0["calls"], len(LAZY_SIZES)))
    if steady:
        assert pct <= 25.0
    assert all(r["asked"] > 0 for r in res)                              # and every rank did a share of it
    assert r0["calls"] <= 4 * len(LAZY_SIZES)


def test_a_board_path_failure_on_one_rank_reaches_every_rank():
    """a rank whose K1-K6 raises inside a round of the hold-off-aware board path still joins the round's gather (with a header
    row that says so): rank 0 ends the fold, the closing broadcast tells everyone, every rank raises from finish() for that
    batch -- nobody is left in a collective -- and the batches after it go through on the state rank 0 broadcast"""
    res = _spawn_lazy(2, True, fail=[1, 3])
    bad = [i for i, x in enumerate(res[0]["raised"]) if x is not None]
    assert len(bad) == 1
    for r in res:
        assert [i for i, x in enumerate(r["raised"]) if x is not None] == bad
        assert r["state"] == res[0]["state"] and r["mtxs"] == res[0]["mtxs"]
    assert any(m is not None for m in res[0]["mtxs"][bad[0] + 1:])
    # the failed batch leaves a GAP, not a shorter film -- the fold advanced over the frames whose records never
    # came as frames without a contour, so the running count every rank plans the next batches from is the film's
    assert res[0]["state"][0] == sum(LAZY_SIZES) and res[0]["seen"] == sum(LAZY_SIZES)


def test_hold_off_aware_pipeline_equals_the_eager_one_and_computes_a_fraction_of_the_records():
    """the reference does not run K1..K6 during the hold-off after a hit (bf_auto.py:43-49).  In the
    hold-off-aware mode the GPU core leaves the board path out; the board fold of a batch runs on the exchange thread
    BEFORE that thread waits for the core (i.e. under the stones path of the same batch) and asks the lanes' board
    contexts, window after window, for the records it looks at.  Two batches in flight, a camera bump: the transform
    after every batch and the fold's counters equal the eager pipeline's, and what is computed stays close to what is
    looked at."""
    from camkifu_amd import capi
    from camkifu_amd.pipeline import LMAX, FastFilePipeline, GpuCore
    from tests.test_fold_cpu import _hough_like, _sides
    rng = np.random.default_rng(21)
    h, w = 1080, 1920
    sides = _sides(rng, h, w)
    sizes = [96, 128, 64, 128, 128, 96, 128, 128]
    total = sum(sizes)
    res = np.zeros(total, capi.BOARD_DTYPE)
    lines = np.zeros((total, LMAX, 2), np.float32)
    for f in range(total):
        if f == 300:
            sides = _sides(rng, h, w)                                # the camera is bumped
        ls = _hough_like(rng, sides, h, w, 1)[:LMAX]
        res["status"][f] = int(rng.choice([0, 0, 0, 0, 2]))
        res["n_lines"][f] = len(ls)
        lines[f, :len(ls)] = ls

    def drive(lazy):
        log = []
        lanes = [(_TableCtx(res, lines, log), _TableCtx(res, lines, log)) for _ in range(2)]
        core = GpuCore(lanes, bg_ctx=_TableCtx(res, lines, log))
        pipe = FastFilePipeline(h, w, ControllerHeadless(), compute=core, board_lazy=lazy)
        assert pipe.board_lazy == lazy
        tickets, mtxs, first = [], [], 0
        for n in sizes:
            tickets.append(pipe.submit(np.arange(first, first + n), n))
            first += n
            if len(tickets) == 2:                                    # two batches in flight
                pipe.finish(tickets.pop(0))
                mtxs.append(None if pipe.mtx is None else pipe.mtx.tolist())
        while tickets:
            pipe.finish(tickets.pop(0))
            mtxs.append(None if pipe.mtx is None else pipe.mtx.tolist())
        b = pipe.board
        out = dict(mtxs=mtxs, looked=b.looked, seen=b.seen, hold=b.hold, hull=b.finder.corners.hull, fetched=b.fetched, calls=b.calls,
                   asked=list(log))
        pipe.close()
        core.close()
        return out
    eager, lazy = drive(False), drive(True)
    assert eager["mtxs"] == lazy["mtxs"] and any(m is not None for m in eager["mtxs"])
    assert (eager["looked"], eager["seen"], eager["hold"], eager["hull"]) == (lazy["looked"], lazy["seen"], lazy["hold"], lazy["hull"])
    assert sum(eager["asked"]) == total                              # the eager pipeline computes every record
    # computed = looked at + the overshoot of the last request of each window; well under half of the records here
    # (these synthetic line bundles need several grouping rounds per detection: 22 % of the frames are looked at)
    assert sum(lazy["asked"]) == lazy["fetched"], (sum(lazy["asked"]), lazy["fetched"])
    assert lazy["looked"] <= lazy["fetched"] <= lazy["looked"] + 0.15 * total < 0.4 * total, (lazy["looked"], lazy["fetched"], total)
    assert lazy["calls"] <= 2 * (total // 50 + 2)                    # a request or two per window


def test_every_worker_thread_of_rank_r_starts_on_gpu_r(monkeypatch):
    """nothing has ever run with local_rank > 0 (one-GPU boxes).  torch's current device is per-thread state
    and a new thread starts on device 0, so every pool the pipeline creates has an initializer that selects the rank's
    GPU: checked here with torch.cuda stubbed out -- a pool for device 3 must call set_device(3) on ITS thread before the
    first task, and a pipeline built over contexts of device 5 must hand 5 to all of its pools."""
    import threading
    import torch
    calls = []
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: calls.append((threading.get_ident(), d)))
    pool = pipeline._pool(1, 3)
    worker = pool.submit(threading.get_ident).result()
    pool.shutdown()
    assert (worker, 3) in calls and worker != threading.get_ident()
    assert pipeline._on_device(None) is None                   # stand-in contexts (CPU tests): nothing to select

    from camkifu_amd import capi

    class _Ctx(capi.Context):                                   # a real Context type without a library behind it
        def __init__(self, device):
            self.device = device

        def __del__(self):
            pass
    made, real_pool = [], pipeline._pool
    monkeypatch.setattr(pipeline, "_pool", lambda k, dev: (made.append(dev), real_pool(k, None))[1])
    lanes = [(_Ctx(5), _Ctx(5)), (_Ctx(5), _Ctx(5))]
    core = pipeline.GpuCore(lanes, bg_ctx=_Ctx(5))
    pipe = pipeline.FastFilePipeline(H, W, ControllerHeadless(), compute=core)
    assert core.device == 5 and pipe.gpu == 5 and made and set(made) == {5}
    pipe.close()
    core.close()
