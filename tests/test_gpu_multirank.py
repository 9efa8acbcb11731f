"""The N > 1 path with the REAL kernels: two processes on the `gloo` backend, both on cuda:0, the
real GpuCore (two lanes of board + stones contexts, the band of the background model on its own context), two batches
in flight through submit() / finish().  What rank 0 ends up with -- requests, transform after every batch, game
record, policy state -- must equal a one-process run over the same film with the same batch schedule.

Reference side: the two finder threads this replaces (core/vmanager.py:408-424), the stones finder reading the board
finder's transform (stone/stonesfinder.py:136-138) and the background model in frame order (:171-176)."""
import os
import socket

import numpy as np
import pytest

H, W = 480, 640
FILM = 120
BATCH = 24                       # 5 batches; the camera is bumped at frame 60 (inside batch 2): the transform changes twice
BUMP = 60
BG = 6


def _film(select):
    """two takes of one game: frames [0, BUMP) from one camera position, [BUMP, FILM) from another (other seed: other
    corners).  After the bump the stones path warps with a stale transform until the board fold catches up -- garbage
    in, but the SAME garbage at every world size."""
    import torch
    from camkifu_amd import synth
    # (rendered in HBM: every process of a comparison renders its frames on the same device from the same per-frame seeds, and
    # the renderer on the CPU took a third of these tests' time)
    a = synth.film(FILM, H, W, seed=11, quiet=8, move_every=30, hand_frames=12, select=[g for g in select if g < BUMP], device="cuda")[0]
    b = synth.film(FILM, H, W, seed=12, quiet=8, move_every=30, hand_frames=12, select=[g for g in select if g >= BUMP], device="cuda")[0]
    return torch.cat([a, b])


def _drive(rank, world, depth=2, force_nccl=False, lazy=False):
    import torch
    import torch.distributed as dist  # noqa: F401
    from camkifu_amd import capi, pipeline
    from camkifu_amd.controller import ControllerHeadless
    from camkifu_amd.stone.nn_manager import NNManager
    torch.cuda.set_device(0)
    lanes = [(capi.Context(0, priority=1 if lazy else 0), capi.Context(0)) for _ in range(2)]
    weights = NNManager.init_net()
    for _, c in lanes:
        c.cnn_set_weights(weights)
    ctrl = ControllerHeadless()
    if force_nccl:           # one rank, but every collective of the exchange stage issued for real, on device buffers over RCCL
        pipe = pipeline.FastFilePipeline(H, W, ctrl, rank=0, world=1, device=torch.device("cuda", 0), lanes=lanes,
                                         ctx_bg=capi.Context(0, priority=1), bg_init_frames=BG, force_exchange=True, board_lazy=lazy)
    else:
        # (two RCCL ranks cannot share a device, so with two processes the collectives run on host buffers over gloo -- but the
        # records are still written by the library in HBM, as on a multi-GPU node: records_in_hbm)
        pipe = pipeline.FastFilePipeline(H, W, ctrl, rank=rank, world=world, device=torch.device("cpu") if world > 1 else None,
                                         lanes=lanes, ctx_bg=capi.Context(0), bg_init_frames=BG, board_lazy=lazy,
                                         records_in_hbm=torch.device("cuda", 0) if world > 1 else None)
    pipe.board.refresh_frames = 5            # keep looking: the bump must be noticed within a batch or two
    mine, batches = [], []
    for b0 in range(0, FILM, BATCH):
        idx = b0 + pipeline.shard_indices(BATCH, rank, world)
        batches.append((len(mine), len(mine) + len(idx)))
        mine.extend(int(g) for g in idx)
    frames = _film(mine)
    tickets, emitted, mtxs = [], [], []
    for k, (lo, hi) in enumerate(batches):
        tickets.append(pipe.submit(frames[lo:hi], BATCH))
        if len(tickets) == depth:                               # two batches in flight
            emitted.append(pipe.finish(tickets.pop(0)))
            mtxs.append(None if pipe.mtx is None else pipe.mtx.tolist())
    while tickets:
        emitted.append(pipe.finish(tickets.pop(0)))
        mtxs.append(None if pipe.mtx is None else pipe.mtx.tolist())
    st = pipe.stones.policy.state()
    xs = pipe._xstream
    return dict(emitted=emitted, mtxs=mtxs, sgf=ctrl.kifu.to_sgf(), targets=st["targets"].tolist(),
                looked=pipe.board.looked, fetched=pipe.board.fetched, calls=pipe.board.calls, host=dict(pipe.host_seconds),
                host_bytes=dict(pipe.group.host_bytes), records_in_hbm=pipe.records_device is not None,
                xstream=None if xs is None else dict(priority=int(xs.priority), is_default=bool(xs == torch.cuda.default_stream(xs.device))))


def _run(rank, world, port, q, force_nccl=False, lazy=False, env=None):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.update(env or {})
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if force_nccl:
        import torch
        torch.cuda.set_device(0)
        from camkifu_amd.pipeline import rccl_group_options
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), pg_options=rccl_group_options())
    elif world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, _drive(rank, world, force_nccl=force_nccl, lazy=lazy)))
    except BaseException as why:                                 # the parent must not wait for a rank that died
        import traceback
        q.put((rank, "FAILED: %s\n%s" % (why, traceback.format_exc())))
        raise
    finally:
        if world > 1 or force_nccl:
            dist.destroy_process_group()


def _spawn(world, force_nccl=False, lazy=False, env=None):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run, args=(r, world, port, q, force_nccl, lazy, env)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=420) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, out in res:
        assert not isinstance(out, str), out
    return [out for _, out in res]


_ONE_RANK = []


def _one_rank():
    """the plain one-process run, made once for the tests that compare against it (a spawn costs ~30 s of imports)"""
    if not _ONE_RANK:
        _ONE_RANK.append(_spawn(1)[0])
    return _ONE_RANK[0]


_RCCL_ONE = []


def _rccl_one_rank():
    """one rank with the whole exchange stage over RCCL, made once"""
    if not _RCCL_ONE:
        _RCCL_ONE.append(_spawn(1, force_nccl=True)[0])
    return _RCCL_ONE[0]


@pytest.mark.gpu
def test_two_ranks_with_real_kernels_equal_one_rank():
    one = _one_rank()
    changes = [k for k in range(1, len(one["mtxs"])) if one["mtxs"][k] != one["mtxs"][k - 1]]
    assert one["mtxs"][0] is not None and len(changes) >= 1, one["mtxs"]          # found, then moved by the bump
    assert any(req for batch in one["emitted"] for req in batch)                  # the policy did record stones
    two = _spawn(2)
    r0, r1 = two
    assert r0["emitted"] == one["emitted"]
    assert r0["sgf"] == one["sgf"] and r0["targets"] == one["targets"] and r0["looked"] == one["looked"]
    assert r0["mtxs"] == one["mtxs"] and r1["mtxs"] == one["mtxs"]                # every rank warps with the same transform
    assert all(e is None for e in r1["emitted"])                                  # only rank 0 folds
    assert r0["host"]["band_model"] > 0 and r1["host"]["band_model"] > 0          # the band model did run on both
    # round 6: what is gathered is gathered to rank 0; rank 1 brought the wire (16 doubles) and a flag word per batch to its host
    from camkifu_amd import pipeline
    n_batches = FILM // BATCH
    assert r0["host_bytes"]["gather"] >= n_batches * BATCH * pipeline.REC_BYTES
    assert r1["host_bytes"]["gather"] == 0 and r1["host_bytes"]["bcast"] <= n_batches * 8 * 16 and r1["host_bytes"]["flag"] <= 4 * n_batches
    assert r0["records_in_hbm"] and r1["records_in_hbm"]                            # written in place in HBM on both ranks


@pytest.mark.gpu
def test_the_exchange_stage_over_rccl_with_one_rank():
    """What a one-GPU box can check of the RCCL path: a process group of ONE rank on the `nccl` backend and the pipeline
    told to run its whole exchange stage anyway -- all-gather of the records, transform broadcast, all-to-all of goban
    bands on device buffers, band model on its own context, counts gather, all issued from the exchange thread.  Same
    requests, transforms and game record as the plain one-rank run (which has no exchange stage at all)."""
    plain = _one_rank()
    rccl = _rccl_one_rank()
    assert rccl["emitted"] == plain["emitted"] and rccl["mtxs"] == plain["mtxs"]
    assert rccl["sgf"] == plain["sgf"] and rccl["targets"] == plain["targets"]
    assert rccl["host"]["band_exchange"] > 0 and rccl["host"]["band_model"] > 0 and plain["host"]["band_exchange"] == 0
    assert rccl["records_in_hbm"] and not plain["records_in_hbm"]       # written by the library where RCCL takes them from


@pytest.mark.gpu
def test_the_band_all_to_all_itself_over_rccl():
    """a rank's own band does not go through the collective (a device copy), so with one rank the all-to-all is not issued at
    all; CK_BAND_SELF_THROUGH_COLLECTIVE=1 sends it through RCCL anyway -- the call every rank of a multi-GPU run makes, on
    device buffers, issued once on this box.  Same answers."""
    plain = _one_rank()
    rccl = _spawn(1, force_nccl=True, env={"CK_BAND_SELF_THROUGH_COLLECTIVE": "1"})[0]
    assert rccl["emitted"] == plain["emitted"] and rccl["mtxs"] == plain["mtxs"] and rccl["sgf"] == plain["sgf"]


@pytest.mark.gpu
def test_a_second_thread_on_a_context_is_refused():
    """the one-thread-per-context contract is enforced by the library: while a thread is inside a call, another
    thread's call on the same ck_ctx returns CK_ERR_STATE (and says why) instead of racing on the stream"""
    import threading
    import torch
    from camkifu_amd import capi
    ctx = capi.Context(0)
    frames = torch.zeros((64, 1080, 1920, 3), dtype=torch.uint8, device="cuda:0")
    small = np.zeros((16, 16, 3), np.uint8)
    refused, stop = [], threading.Event()

    def hammer(img):
        while not stop.is_set():
            try:
                ctx.median15(img)
            except capi.CkError as why:
                refused.append(str(why))
                stop.set()
    threads = [threading.Thread(target=hammer, args=(img,)) for img in (frames, small)]
    for t in threads:
        t.start()
    stop.wait(60)
    stop.set()
    for t in threads:
        t.join()
    assert refused and "in use by another thread" in refused[0], refused
    assert np.array_equal(np.asarray(ctx.median15(small)), small)                 # and the context is fine afterwards
    ctx.close()


def _bench(*extra, timeout=900):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(extra), env=env, capture_output=True, text=True,
                       timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_from_the_plain_command_with_two_ranks():
    """`python bench.py --gpus 2 ...` with no external launcher starts its two ranks itself (before
    anything touches the GPU), each with its own RANK / LOCAL_RANK, and prints rank 0's ONE line.  On a one-GPU box the
    ranks share cuda:0 and talk over gloo (two RCCL ranks cannot share a device); frames are dealt 0, 2, 4 ... / 1, 3, 5 ...
    and the fold's game record must be the true one."""
    out = _bench("--gpus", "2", "--dist-backend", "gloo", "--single-device", "--frames", "128", "--steps", "2", "--warmup", "1",
                 "--no-extras", "--no-cpu-baseline")
    assert out["n_gpus"] == 2 and out["config"]["frames_per_gpu"] == 128
    assert out["value"] > 0 and out["board_found_by_fold"] is True
    assert out["move_sequence_ratio"] == 1.0 and out["stone_grid_match_pct"] == 100.0
    assert out["host_ms_per_step"]["band_exchange"] > 0          # the pixel-sharded background model did exchange bands


@pytest.mark.gpu
def test_the_exchange_stage_does_not_gate_the_lanes():
    """The mechanism, not a wall-clock ratio (the ratio's failure case sat within one sigma of its threshold; the
    measured ratios stay in tools/attic/exchange_runs.sh and in the bench line's `rccl_exchange_one_rank`).  The exchange thread's
    torch work and its waits for the collectives sit on that thread's OWN stream -- never torch's default stream, which every
    lane thread orders its calls behind (capi._in) -- and that stream is a HIGH-PRIORITY one, so that the stage's chain of
    short dependent pieces does not queue behind the lanes' millisecond launches; with the whole stage issued over RCCL (one
    rank) every collective did run and the results are the plain run's."""
    plain = _one_rank()
    rccl = _rccl_one_rank()
    xs = rccl["xstream"]
    assert xs is not None and not xs["is_default"]
    assert xs["priority"] < 0, xs                                          # torch: negative = above the default priority
    assert plain["xstream"] is None                                         # no exchange stage, no stream
    for k in ("gather", "bcast", "band_exchange", "flags", "counts_gather", "band_model"):
        assert rccl["host"][k] > 0, k
    assert rccl["emitted"] == plain["emitted"] and rccl["mtxs"] == plain["mtxs"]


@pytest.mark.gpu
def test_the_hold_off_aware_board_path_across_ranks_with_real_kernels():
    """VERDICT r4 item 4 on the GPU: board_lazy together with the exchange stage.  (a) One rank over RCCL: every round of the
    board path (planned first request, gather, follow-up requests by broadcast) issued for real on device buffers.  (b) Two
    processes on one device over gloo, frames dealt 0, 2, 4 ... / 1, 3, 5 ...: each rank runs K1-K6 only on the planned
    frames it owns.  Requests, transforms after every batch, game record and policy state equal the eager one-rank run's;
    fewer board records computed than frames filmed."""
    plain = _one_rank()
    one = _spawn(1, force_nccl=True, lazy=True)[0]
    assert one["emitted"] == plain["emitted"] and one["mtxs"] == plain["mtxs"] and one["sgf"] == plain["sgf"]
    assert one["looked"] == plain["looked"] and plain["looked"] <= one["fetched"] < FILM and one["calls"] > 0
    r0, r1 = _spawn(2, lazy=True)
    assert r0["emitted"] == plain["emitted"] and r0["sgf"] == plain["sgf"] and r0["targets"] == plain["targets"]
    assert r0["mtxs"] == plain["mtxs"] and r1["mtxs"] == plain["mtxs"]
    assert r0["looked"] == plain["looked"] and r0["fetched"] == one["fetched"]
    assert r0["host"]["lazy_detect"] > 0 and r1["host"]["lazy_detect"] > 0          # both ranks computed board records


@pytest.mark.gpu
def test_hold_off_aware_pipeline_equals_the_eager_one_on_the_gpu():
    """FastFilePipeline(board_lazy=True) with the real kernels: the board fold asks the lanes' board contexts for the frames
    of its hypotheses (non-consecutive frame sets: gathered on the device, see pipeline._take), the stones path runs on
    every frame.  Requests, transform after every batch, game record and counters equal the eager pipeline's on the same
    film, two batches in flight, a short hold-off so that several windows fall into every batch."""
    import torch
    from camkifu_amd import capi, pipeline
    from camkifu_amd.controller import ControllerHeadless
    from camkifu_amd.stone.nn_manager import NNManager
    torch.cuda.set_device(0)
    frames = _film(list(range(FILM)))
    weights = NNManager.init_net()

    def drive(lazy):
        lanes = [(capi.Context(0, priority=1 if lazy else 0), capi.Context(0)) for _ in range(2)]
        for _, c in lanes:
            c.cnn_set_weights(weights)
        ctrl = ControllerHeadless()
        pipe = pipeline.FastFilePipeline(H, W, ctrl, lanes=lanes, ctx_bg=capi.Context(0), bg_init_frames=BG, board_lazy=lazy)
        pipe.board.refresh_frames = 9                    # several windows per batch, and windows that straddle batches
        tickets, emitted, mtxs = [], [], []
        for b0 in range(0, FILM, BATCH):
            tickets.append(pipe.submit(frames[b0:b0 + BATCH], BATCH))
            if len(tickets) == 2:
                emitted.append(pipe.finish(tickets.pop(0)))
                mtxs.append(None if pipe.mtx is None else pipe.mtx.tolist())
        while tickets:
            emitted.append(pipe.finish(tickets.pop(0)))
            mtxs.append(None if pipe.mtx is None else pipe.mtx.tolist())
        b = pipe.board
        out = dict(emitted=emitted, mtxs=mtxs, sgf=ctrl.kifu.to_sgf(), looked=b.looked, seen=b.seen, hold=b.hold, fetched=b.fetched, calls=b.calls)
        pipe.close()
        for pair in lanes:
            for c in pair:
                c.close()
        return out
    eager, lazy = drive(False), drive(True)
    assert eager["mtxs"] == lazy["mtxs"] and any(m is not None for m in eager["mtxs"])
    assert eager["emitted"] == lazy["emitted"] and eager["sgf"] == lazy["sgf"]
    assert (eager["looked"], eager["seen"], eager["hold"]) == (lazy["looked"], lazy["seen"], lazy["hold"])
    assert lazy["looked"] <= lazy["fetched"] < FILM and lazy["calls"] > 0, lazy


@pytest.mark.gpu
def test_closing_the_pipeline_then_the_contexts_with_a_batch_in_flight():
    """close() waits for the batches in flight, and ck_ctx_destroy waits for a thread still inside a call --
    the with-statement pattern (close the pipeline, drop the contexts) cannot free a stream under a lane's thread"""
    import torch
    from camkifu_amd import capi, pipeline, synth
    from camkifu_amd.controller import ControllerHeadless
    frames = synth.film(32, H, W, seed=5, quiet=8, move_every=30, hand_frames=12)[0].cuda()
    lanes = [(capi.Context(0), capi.Context(0)) for _ in range(2)]
    bg = capi.Context(0)
    with pipeline.FastFilePipeline(H, W, ControllerHeadless(), lanes=lanes, ctx_bg=bg) as pipe:
        pipe.submit(frames, 32)                       # never finished by the caller
        pipe.submit(frames, 32)
    for pair in lanes:
        for c in pair:
            c.close()
    bg.close()
    torch.cuda.synchronize()
    c = capi.Context(0)
    assert np.array_equal(np.asarray(c.median15(np.zeros((16, 16, 3), np.uint8))), np.zeros((16, 16, 3), np.uint8))
    c.close()
