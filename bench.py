#!/usr/bin/env python3
"""bench.py -- frames/sec of the per-frame vision hot path on synthetic 1080p video.

One "step" = one pass of the hot path over one batch of frames that is already resident in
HBM: board detect (K1..K6: median -> Canny -> contours -> Hough lines, lines back on the
host) and stones detect (K8, K10..K12: warp -> 100 patches -> CNN -> 19x19 labels) for every
frame of the batch, then (N > 1) ONE RCCL all-gather of the fixed-size per-frame records and the
ordered host fold (line accumulation -> corners; label acceptance -> moves) on every rank.
Workload (BASELINE.json configs[2]): 1080p, 256-frame batches on one MI355X; with N GPUs
every rank processes its own 256-frame shard of the video (weak scaling, no data-path
collective except the label gather).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by
torch.distributed.run, one rank per GPU.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
MFMA_F32_PEAK_TF = 157.3         # v_mfma_f32_32x32x2_f32 dense peak
MFMA_BF16_PEAK_TF = 2500.0

# MACs per frame (100 patches), SURVEY.md 8(a)
MACS = dict(cnn_conv1=311.04e6, cnn_conv2=2621.44e6, cnn_conv3=508.03e6, cnn_conv4=1049.76e6)


def cpu_baseline(h, w, frames, corners, weights, nframes):
    """Time the CPU oracle (our C restatement, OpenMP) on a bounded sample of the same workload."""
    import numpy as np
    from oracle import oracle as ora
    dst = np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32)
    M = ora.get_perspective_transform(corners, dst)
    sample = [frames[i].cpu().numpy() for i in range(nframes)]
    t0 = time.perf_counter()
    for fr in sample:
        ora.board_lines(ora.canny(ora.median(fr, 15), 25, 75))
        ora.decode_all(ora.cnn_predict_regions(weights, ora.warp_perspective(fr, M)))
    dt = time.perf_counter() - t0
    return dict(value=round(nframes / dt, 4), unit="frames/s", cores=ora.num_threads(), kind="port",
                sample="%d frames of the same %dx%d batch, board path + stones path, oracle/*.c with OpenMP"
                       % (nframes, w, h))


LANES_DEFAULT = 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=256, help="frames per batch per GPU")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--cnn", choices=["fp32", "bf16", "f16x2"], default="f16x2",
                    help="f16x2 (default): f32-accurate split-fp16 operands on the fp16 matrix pipe; fp32: k-ordered f32 "
                         "MFMA chain; bf16: bf16 operands (BASELINE config 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) | gloo (rehearsal on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--cpu-frames", type=int, default=8)
    ap.add_argument("--streams", action="store_true",
                    help="BASELINE config 5: every GPU processes its OWN video stream (seed + rank), no record gather; "
                         "the default is ONE video whose frames are dealt to the ranks and gathered (configs 3 / 4)")
    ap.add_argument("--lanes", type=int, default=LANES_DEFAULT,
                    help="pairs of (board, stones) contexts per GPU; the batch is split between them so more "
                         "kernels are in flight and drain / host gaps of one lane are filled by the others")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")      # where collective buffers live
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

    from camkifu_amd import capi, synth
    from concurrent.futures import ThreadPoolExecutor
    # like the reference, the board finder and the stones finder are two threads; each owns a
    # context (= HIP stream + scratch), so the host-side gaps of one path are filled by the other
    ctx_b = capi.Context(local_rank)
    ctx = capi.Context(local_rank)
    lanes = [(ctx_b, ctx)] + [(capi.Context(local_rank), capi.Context(local_rank)) for _ in range(args.lanes - 1)]
    H, W, F = args.height, args.width, args.frames

    # ---- synthetic video shard of this rank, rendered straight into HBM --------------------
    # ONE game filmed by a fixed camera, world*F frames long: a random mid-game position, then one
    # new stone every 5 frames.  Global frame g lives on rank g mod world (the
    # pipeline's sharding), so every rank builds the same move list and renders only its frames;
    # frames that show an unchanged position are the last render under fresh sensor noise.
    dworld, drank = (1, 0) if args.streams else (world, rank)   # how the VIDEO is laid out over the ranks
    rng = np.random.default_rng(synth.SEED + (rank if args.streams else 0))     # one game: identical on every rank
    corners = synth.random_corners(H, W, rng)
    frames = torch.empty((F, H, W, 3), dtype=torch.uint8, device=dev)
    stones0 = synth.random_stones(rng, density=0.3)
    true_moves = [("EBW"[stones0[r, c]], r, c) for r in range(19) for c in range(19) if stones0[r, c]]
    n_init = len(true_moves)
    positions, st, color = [stones0.copy()], stones0.copy(), 1
    for k in range((dworld * F - 1) // 5):
        while True:
            r, c = rng.integers(1, 18, 2)
            if st[r, c] == 0:
                break
        st[r, c] = color
        true_moves.append(("EBW"[color], int(r), int(c)))
        color = 3 - color
        positions.append(st.copy())
    truth = np.zeros((F, 19, 19), np.uint8)
    last_pos, last = -1, None
    for i in range(F):
        g_idx = i * dworld + drank
        pos = g_idx // 5
        if pos != last_pos:
            frames[i] = synth.render(H, W, positions[pos], corners, seed=synth.SEED + g_idx, device=dev)
            last_pos, last = pos, i
        else:
            g = torch.Generator(device=dev)
            g.manual_seed(synth.SEED + 7 * g_idx)
            noise = torch.randint(-2, 3, frames[last].shape, generator=g, device=dev, dtype=torch.int16)
            frames[i] = (frames[last].to(torch.int16) + noise).clamp_(0, 255).to(torch.uint8)
        truth[i] = positions[pos]
    from camkifu_amd.stone.nn_manager import NNManager, GOLDEN_WEIGHTS
    weights = NNManager.init_net()               # trained fixture when present, else seeded He-normal
    torch.cuda.synchronize()
    for _, c in lanes:
        c.cnn_set_weights({k: torch.from_numpy(v).to(dev) for k, v in weights.items()})
        c.cnn_set_mode({"fp32": capi.CK_CNN_FP32, "bf16": capi.CK_CNN_BF16, "f16x2": capi.CK_CNN_F16X2}[args.cnn])
    dst = np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32)
    M = capi.get_perspective_transform(corners, dst)
    from camkifu_amd import pipeline
    from camkifu_amd.controller import ControllerHeadless
    pipe = pipeline.FastFilePipeline(H, W, ControllerHeadless(), ctx=ctx, ctx_board=ctx_b, rank=drank, world=dworld,
                                     device=cdev)
    # one host thread per context (a context is single-threaded by contract), each with its own stream
    pools = [(ThreadPoolExecutor(1), ThreadPoolExecutor(1)) for _ in lanes]
    cuts = [round(i * F / len(lanes)) for i in range(len(lanes) + 1)]
    slices = [frames[cuts[i]:cuts[i + 1]] for i in range(len(lanes))]

    class _Both:
        """the per-lane futures of one step, joined: board = (records, lines), stones = (labels, conf)"""
        def __init__(self, futs):
            self.futs = futs

        def board(self):
            parts = [fb.result() for fb, _ in self.futs]
            return (np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]))

        def stones(self):
            parts = [fs.result() for _, fs in self.futs]
            return torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])

    def launch():
        """GPU part of one step: per lane, the board path and the stones path on two host threads / HIP streams"""
        return _Both([(pb.submit(cb.board_detect, fr, -1, pipeline.LMAX, True),        # K1..K6, lines on the host
                       ps.submit(cs.stones_detect, fr, M))                             # K8, K10..K12, labels in HBM
                      for (pb, ps), (cb, cs), fr in zip(pools, lanes, slices)])

    host_s = [0.0, 0.0, 0.0]        # pack, gather, fold: host seconds spent per phase (diagnostic)

    def finish_host(board, labels, conf):
        """pack the fixed-size per-frame records, one all-gather (RCCL over xGMI), ordered fold"""
        t_a = time.perf_counter()
        rec = pipeline.pack_records_raw(board[0], board[1], labels.cpu().numpy(), conf.cpu().numpy())
        t_b = time.perf_counter()
        full = pipeline.gather_records(rec, world * F, rank, world, cdev) if dworld > 1 else rec
        t_c = time.perf_counter()
        pipe.stones = pipeline.StonesFold(ControllerHeadless())     # every step replays the same game from scratch
        pipe.fold(full)
        t_d = time.perf_counter()
        host_s[0] += t_b - t_a; host_s[1] += t_c - t_b; host_s[2] += t_d - t_c

    def run_steps(k):
        """k steps, two batches in flight: the host part of batch i (contour pruning inside board_detect,
        records, gather, fold) overlaps the GPU work of batches i+1 and i+2"""
        DEPTH = 2
        inflight = [launch() for _ in range(min(DEPTH, k))]
        for i in range(k):
            futs = inflight.pop(0)
            board = futs.board()
            labels, conf = futs.stones()
            if i + DEPTH < k:
                inflight.append(launch())
            finish_host(board, labels, conf)
        return board, labels

    def step_serial():
        board = ctx_b.board_detect(frames, cap=pipeline.LMAX)
        labels, conf = ctx.stones_detect(frames, M)
        return board, labels

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup:
        run_steps(args.warmup)
    sync()
    host_s[:] = [0.0, 0.0, 0.0]
    t0 = time.perf_counter()
    board, labels = run_steps(args.steps)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # per-kernel durations: HIP events on each context's own stream, taken over a short SERIAL
    # pass right after the timed region (with the two paths overlapped, an event pair on one
    # stream would also count the time the other stream's kernels hold the CUs)
    prof_steps = 2
    if rank == 0:
        for c in (ctx, ctx_b):
            c.timing_enable(True)
            c.timing_reset()
        for _ in range(prof_steps):
            step_serial()
        torch.cuda.synchronize()
    if rank == 0:
        stage_names = ["median", "canny_nms", "canny_hyst", "ccl", "contour_gather", "ghost", "hough_vote",
                       "hough_peaks", "warp", "cnn_conv1", "cnn_conv2", "cnn_conv3", "cnn_conv4", "cnn_tail"]
        stages = {}
        for nme in stage_names:
            ms, cnt = ctx.timing_get(nme)
            ms2, cnt2 = ctx_b.timing_get(nme)
            ms, cnt = ms + ms2, cnt + cnt2
            if cnt:
                stages[nme] = dict(ms_total=round(ms, 3), launches=cnt, us_per_frame=round(1e3 * ms / (prof_steps * F), 3))
        ctx.timing_enable(False)
        ctx_b.timing_enable(False)
        pmc = {}
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.isfile(pmc_path) and (H, W) == (1080, 1920):
            pmc = json.load(open(pmc_path))

        def traffic_of(stage, frames_per_launch):
            t = pmc.get(stage)
            return None if not t else int(t["hbm_bytes_per_frame"] * frames_per_launch)
        def valu_frac(stage):
            """VALU issue-slot utilisation: wave-level VALU instructions (PMC SQ_INSTS_VALU, profiles/) x 4 cycles per
            wave64 instruction on a SIMD16 / (1024 SIMDs x 2.4 GHz x kernel time)"""
            t = pmc.get(stage)
            if not t or "valu_wave_insts_per_frame" not in t:
                return None
            return round(t["valu_wave_insts_per_frame"] * 4.0 / (1024 * 2.4e9 * stages[stage]["us_per_frame"] * 1e-6), 4)
        def mfma_busy_frac(stage, cycles_per_inst):
            """share of the kernel time the matrix pipe is busy: MFMA instructions (PMC SQ_INSTS_MFMA) x their pipe cycles"""
            t = pmc.get(stage)
            if not t or not t.get("mfma_wave_insts_per_frame"):
                return None
            return round(t["mfma_wave_insts_per_frame"] * cycles_per_inst / (1024 * 2.4e9 * stages[stage]["us_per_frame"] * 1e-6), 4)

        # split-precision mode: conv1 is fused into conv2 (its stage slot only times an empty scope)
        fused_conv1 = args.cnn == "f16x2" and "cnn_conv1" in stages and stages["cnn_conv1"]["us_per_frame"] < 0.5
        if fused_conv1:
            del stages["cnn_conv1"]
        fused_conv34 = args.cnn == "f16x2" and "cnn_conv4" in stages and "cnn_conv3" not in stages

        def roof_of(stage):
            per_launch_frames = prof_steps * F / stages[stage]["launches"]
            avg_s = stages[stage]["ms_total"] / stages[stage]["launches"] * 1e-3
            if stage in MACS:
                # f16x2 executes three fp16 MFMAs per f32-equivalent MAC block: priced against the fp16 peak
                f32_kernel = args.cnn == "fp32" or (args.cnn == "bf16" and stage == "cnn_conv1")     # bf16 mode keeps conv1 in f32
                peak = MFMA_F32_PEAK_TF if f32_kernel else MFMA_BF16_PEAK_TF
                mults = 1.0
                if args.cnn == "f16x2":
                    mults = 2.0 if stage == "cnn_conv1" else 3.0       # fp16 MFMAs executed per f32-equivalent product
                macs = mults * MACS[stage]
                r_note = None
                if fused_conv1 and stage == "cnn_conv2":
                    macs += 2.0 * MACS["cnn_conv1"]          # conv1 runs inside this kernel (two fp16 MFMAs per product)
                    r_note = "conv1 is computed inside this kernel's staging; its algorithmic flops are included, the halo rows recomputed per block are not"
                if fused_conv34 and stage == "cnn_conv4":
                    macs += mults * MACS["cnn_conv3"]        # conv3 + conv4 of a patch run in one workgroup
                    r_note = "conv3 and conv4 in one kernel (conv3's output stays in LDS); flops of both"
                ach = 2.0 * macs * per_launch_frames / avg_s / 1e12
                r = dict(kernel=stage, bound="mfma", achieved=round(ach, 3), peak=peak, unit="TFLOP/s",
                         frac=round(ach / peak, 5), traffic=traffic_of(stage, per_launch_frames))
                if r_note:
                    r["note"] = r_note
                return r
            bytes_per_frame = {"median": 2 * 3 * W * H, "canny_nms": 4 * W * H, "warp": 433200 + 3 * W * H,
                               "ccl": 6 * W * H, "canny_hyst": 2 * W * H}.get(stage, 4 * W * H)
            ach = bytes_per_frame * per_launch_frames / avg_s / 1e9
            r = dict(kernel=stage, bound="hbm", achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                     frac=round(ach / HBM_PEAK_GBS, 5), traffic=traffic_of(stage, per_launch_frames))
            vf = valu_frac(stage)
            if vf is not None:
                r["valu_issue_frac"] = vf
                mf = mfma_busy_frac(stage, 16.0)        # v_mfma_i32_16x16x64_i8: 4 passes of 4 cycles
                if mf is not None:
                    r["mfma_busy_frac"] = mf
                    r["note"] = ("nominally HBM-bound (SURVEY 8d), in fact bound by instruction issue: the box sums run on the i8 "
                                 "matrix cores (busy mfma_busy_frac of the kernel time), the rest on the vector ALUs (valu_issue_frac of "
                                 "their issue slots, at the nominal 2.4 GHz); the two add up, a SIMD overlaps them only marginally")
                else:
                    r["note"] = ("nominally HBM-bound (SURVEY 8d), in fact bound by VALU instruction issue: "
                                 "valu_issue_frac of the SIMDs' issue slots are busy")
            return r
        # roofline of the dominant kernel
        dom = max(stages, key=lambda k: stages[k]["ms_total"]) if stages else None
        roof = roof_of(dom) if dom is not None else None
        # the dominant matrix-core kernel and the filter pass (K1; north_star quotes the HBM roofline on it) are
        # always reported as well
        conv_stages = [k for k in stages if k in MACS]
        mfma_roof = roof_of(max(conv_stages, key=lambda k: stages[k]["ms_total"])) if conv_stages else None
        filt = roof_of("median") if "median" in stages else None
        out = {
            "metric": "frames/sec on 1920x1080 video + 19x19 stone-grid match % vs reference SGF",
            "value": round(world * F * args.steps / dt, 2),
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "u8+f32", "bf16": "u8+bf16", "f16x2": "u8+f16x2(f32 accumulate)"}[args.cnn], "data": "synthetic",
            "config": {"workload": "%dx%d synthetic video, %d-frame batch per GPU, board detect (K1-K6) + "
                                   "stones detect (K8,K10-K12), cnn %s" % (W, H, F, args.cnn),
                       "frames_per_gpu": F, "height": H, "width": W, "parallelism": ("%d independent streams" % world) if args.streams else ("frames sharded x%d" % world),
                       "lanes_per_gpu": len(lanes)},
            "roofline": roof,
            "mfma_kernel": mfma_roof,
            "filter_pass": filt,
            "stages": stages,
            "stage_timing": "HIP events per context stream over %d serial steps after the timed region; the timed "
                            "region overlaps the board and stones paths on two streams" % prof_steps,
            "host_ms_per_step": {"pack_records": round(1e3 * host_s[0] / args.steps, 3),
                                 "gather": round(1e3 * host_s[1] / args.steps, 3),
                                 "ordered_fold": round(1e3 * host_s[2] / args.steps, 3),
                                 "note": "rank 0; overlapped with the GPU work of the next batches"},
            "lines_found_frame0": int(board[0]["n_lines"][0]),
            "board_found_by_fold": pipe.board.mtx is not None,
            "moves_recorded_by_fold": len(pipe.stones.controller.kifu.moves),
            "move_sequence_ratio": round(__import__("difflib").SequenceMatcher(
                a=["%s%d,%d" % m for m in true_moves],
                b=["%s%d,%d" % (m.color, m.y, m.x) for m in pipe.stones.controller.kifu.moves[:len(true_moves)]]).ratio(), 4),
            "stone_grid_match_pct": round(100.0 * float((labels.cpu().numpy() == truth).mean()), 3),
            "cnn_weights": "trained on synthetic boards (camkifu_amd/data/keras.h5, Keras-1 HDF5 layout)" if os.path.isfile(GOLDEN_WEIGHTS)
                           else "seeded random (labels meaningless)",
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(H, W, frames, corners, weights, args.cpu_frames)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
