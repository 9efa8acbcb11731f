#!/usr/bin/env python3
"""Timing of SURVEY 8f rank 3 on goban images resident in HBM: ck_contour_stones (SfContours.find_stones) and
ck_find_intersections (StonesFinder.find_intersections) -- images/s at a few batch sizes, HIP-event time per stage,
the CPU oracle beside them, and (CK_PROFILE_HOST=1) the host laps of ck_contour_stones on stderr.
    python tools/stonefind_timing.py [--n 256] [--reps 5] [--cpu 2]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu", type=int, default=2, help="images timed on the CPU oracle (0: skip)")
    args = ap.parse_args()
    import torch
    from camkifu_amd import capi, synth
    from camkifu_amd.stone.stonesfinder import PosGrid
    ctx = capi.Context(0)
    n = args.n
    film, corners, truth, moves, hands = synth.film(n, 480, 640, seed=synth.SEED, quiet=50, move_every=12, hand_frames=6, device="cuda")
    M = capi.get_perspective_transform(corners, np.array([(0, 0), (380, 0), (380, 380), (0, 380)], np.float32))
    gobans = torch.empty((n, 380, 380, 3), dtype=torch.uint8, device="cuda")
    ctx.warp_perspective(film, M, out=gobans)
    h = ctx.mog2_create(380, 380)
    fgs = torch.stack([torch.as_tensor(ctx.mog2_apply(h, gobans[f], 0.01 if f < 50 else 0.005)) for f in range(n)]).cuda()
    rects = PosGrid(380).zones(1.0)
    out = {}
    for b in sorted({1, 16, 64, n}):
        if b > n:
            continue
        g, m = gobans[n - b:].contiguous(), fgs[n - b:].contiguous()
        ctx.contour_stones(g, m, rects)
        t = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            stones = ctx.contour_stones(g, m, rects)
            t.append(time.perf_counter() - t0)
        out["batch_%d" % b] = dict(images_per_s=round(b / float(np.median(t)), 1), ms_per_call=round(1e3 * float(np.median(t)), 3))
    ctx.timing_enable(True)
    ctx.timing_reset()
    ctx.contour_stones(gobans, fgs, rects)
    stages = {}
    for name in ("stonefind_open", "median", "canny_nms", "canny_hyst", "survey_ccl", "survey_trace", "survey_gather", "stonefind_zones"):
        try:
            ms, k = ctx.timing_get(name)
            stages[name] = dict(us_per_image=round(1e3 * ms / n, 3), launches=k)
        except Exception:
            pass
    ctx.timing_enable(False)
    out["stages_batch_%d" % n] = stages
    agree = float((stones == np.stack(truth[n - len(stones):])).mean()) if len(truth) >= n else None
    out["grid_agreement_with_truth"] = agree
    if args.cpu:
        from oracle import ora_stones
        gh, mh = gobans[n - args.cpu:].cpu().numpy(), fgs[n - args.cpu:].cpu().numpy()
        t0 = time.perf_counter()
        ref = [ora_stones.find_stones(a, b) for a, b in zip(gh, mh)]
        out["cpu_oracle"] = dict(images_per_s=round(args.cpu / (time.perf_counter() - t0), 2), kind="port", note="numpy + C restatement, one core")
        out["cpu_oracle"]["equal"] = bool(all(np.array_equal(r, s) for r, s in zip(ref, stones[len(stones) - args.cpu:])))
    # two contexts in flight (two host threads, ctypes drops the GIL): one call's host geometry overlaps the other's kernels
    import threading
    ctx2 = capi.Context(0)
    halves = [(gobans[:n // 2].contiguous(), fgs[:n // 2].contiguous()), (gobans[n // 2:].contiguous(), fgs[n // 2:].contiguous())]
    for c, (g, m) in zip((ctx, ctx2), halves):
        c.contour_stones(g, m, rects)
    rounds = 6

    def worker(c, g, m):
        for _ in range(rounds):
            c.contour_stones(g, m, rects)
    th = [threading.Thread(target=worker, args=(c, g, m)) for c, (g, m) in zip((ctx, ctx2), halves)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    out["two_contexts_in_flight"] = dict(images_per_s=round(rounds * n / (time.perf_counter() - t0), 1), batch_per_context=n // 2)
    ctx2.close()
    # ---- find_intersections ----
    mtx = PosGrid(380).mtx
    gi = {}
    for b in sorted({1, 64, n}):
        if b > n:
            continue
        g = gobans[n - b:].contiguous()
        ctx.find_intersections(g, mtx, rects)
        t = []
        for _ in range(args.reps):
            t0 = time.perf_counter()
            grid = ctx.find_intersections(g, mtx, rects)
            t.append(time.perf_counter() - t0)
        gi["batch_%d" % b] = dict(images_per_s=round(b / float(np.median(t)), 1), ms_per_call=round(1e3 * float(np.median(t)), 3))
    ctx.timing_enable(True)
    ctx.timing_reset()
    ctx.find_intersections(gobans, mtx, rects)
    stages = {}
    for name in ("grid_gray", "canny_nms", "canny_hyst", "grid_hough"):
        try:
            ms, k = ctx.timing_get(name)
            stages[name] = dict(us_per_image=round(1e3 * ms / n, 3), launches=k)
        except Exception:
            pass
    ctx.timing_enable(False)
    gi["stages_batch_%d" % n] = stages
    gi["zones_seen_empty_per_image"] = round(float((grid[..., 0] < 0).sum()) / len(grid), 1)
    if args.cpu:
        from oracle import ora_grid
        gh = gobans[n - args.cpu:].cpu().numpy()
        t0 = time.perf_counter()
        ref = [ora_grid.find_intersections(a, mtx, rects) for a in gh]
        gi["cpu_oracle"] = dict(images_per_s=round(args.cpu / (time.perf_counter() - t0), 2), kind="port", note="numpy restatement, one core",
                                equal=bool(all(np.array_equal(r, s) for r, s in zip(ref, grid[len(grid) - args.cpu:]))))
    print(json.dumps({"contour_stones": out, "find_intersections": gi}))


if __name__ == "__main__":
    main()
