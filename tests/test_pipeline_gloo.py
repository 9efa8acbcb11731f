"""N > 1 path on CPU: two processes, `gloo` backend.  Frames are dealt f -> rank f mod 2, each
rank computes its shard's per-frame records, ONE all_gather moves them, and the ordered fold on
every rank must give exactly what a single process gives on the whole batch."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from camkifu_amd import pipeline
from camkifu_amd.controller import ControllerHeadless

H, W, NF = 480, 640, 13          # odd on purpose: ranks hold 7 and 6 frames


def _fake_compute(frame_ids):
    """deterministic stand-in for the GPU core: records depend only on the global frame index"""
    def compute(frames, mtx):
        board, labels, conf = [], [], []
        for f in frame_ids:
            rng = np.random.default_rng(1000 + int(f))
            # the four sides of a slanted board (so that the fold does find corners) plus a
            # frame-dependent number of near-duplicates, as Hough peaks come in practice
            sides = np.array([[100, 0.05], [540, 0.08], [60, 1.55], [420, 1.62]], np.float32)
            k = int(rng.integers(0, 4))
            dup = sides[rng.integers(0, 4, k)] + np.stack([rng.integers(-1, 2, k), np.zeros(k)], 1).astype(np.float32)
            lines = np.concatenate([sides, dup])
            board.append(dict(status=0, n_contours=1 + int(f), n_lines=len(lines), biggest_area=2e5 + f, lines=lines))
            labels.append(rng.integers(0, 3, (19, 19)).astype(np.uint8))
            conf.append(rng.uniform(0.3, 1.0, (19, 19)))
        return board, np.stack(labels), np.stack(conf)
    return compute


def _run(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ctrl = ControllerHeadless()
    idx = pipeline.shard_indices(NF, rank, world)
    pipe = pipeline.FastFilePipeline(H, W, ctrl, rank=rank, world=world, compute=_fake_compute(idx))
    pipe.board.refresh_frames = 3
    out = []
    for batch in range(2):
        emitted = pipe.process_batch(None, NF)
        out.append([[(m.color, m.x, m.y) for m in mv] for mv in emitted])
    q.put((rank, out, ctrl.kifu.to_sgf(), None if pipe.board.mtx is None else pipe.board.mtx.tolist()))
    dist.destroy_process_group()


def _single():
    ctrl = ControllerHeadless()
    pipe = pipeline.FastFilePipeline(H, W, ctrl, compute=_fake_compute(np.arange(NF)))
    pipe.board.refresh_frames = 3
    out = []
    for batch in range(2):
        emitted = pipe.process_batch(None, NF)
        out.append([[(m.color, m.x, m.y) for m in mv] for mv in emitted])
    return out, ctrl.kifu.to_sgf(), None if pipe.board.mtx is None else pipe.board.mtx.tolist()


def test_record_roundtrip():
    board, labels, conf = _fake_compute(np.arange(5))(None, None)
    rec = pipeline.pack_records(board, labels, conf)
    assert rec.shape == (5, pipeline.REC_BYTES)
    for f in range(5):
        r = pipeline.unpack_record(rec[f])
        assert r["n_contours"] == board[f]["n_contours"] and r["biggest_area"] == board[f]["biggest_area"]
        assert np.array_equal(r["lines"], board[f]["lines"]) and np.array_equal(r["labels"], labels[f])
        assert np.array_equal(r["conf"], conf[f])
    assert list(pipeline.shard_indices(7, 1, 3)) == [1, 4]


def test_two_rank_fold_equals_single_process():
    # K7 (ck_get_perspective_transform) is host-only code of the C-ABI: it runs without a GPU
    ref = _single()
    assert ref[2] is not None                      # the fold did find the board
    assert sum(len(m) for m in ref[0][1]) > 0      # second batch emits moves (transform known)

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_run, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    for rank, out, sgf, mtx in res:
        assert out == ref[0] and sgf == ref[1]
        assert np.allclose(np.array(mtx), np.array(ref[2]))


def test_raw_pack_equals_dict_pack():
    from camkifu_amd import capi
    board, labels, conf = _fake_compute(np.arange(9))(None, None)
    res = np.zeros(9, capi.BOARD_DTYPE)
    lines = np.zeros((9, pipeline.LMAX, 2), np.float32)
    for f, b in enumerate(board):
        res[f] = (b["status"], b["n_contours"], b["n_lines"], 0, b["biggest_area"])
        lines[f, :b["n_lines"]] = b["lines"]
        lines[f, b["n_lines"]:] = 7.0                      # stale scratch beyond n_lines must not leak
    assert np.array_equal(pipeline.pack_records_raw(res, lines, labels, conf), pipeline.pack_records(board, labels, conf))
