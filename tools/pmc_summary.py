#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, optionally SQ_INSTS_VALU + SQ_INSTS_MFMA; separate runs, --kernel-trace only)
into per-stage HBM traffic per frame, corrected as MI355X_MICROARCH.md prescribes for gfx950:
FETCH_SIZE counts 128-B read requests at 64 B, so wide coalesced reads are doubled; WRITE_SIZE
is exact; both are in KiB.  SQ_INSTS_VALU = wave-level VALU instructions (each costs one 4-cycle issue slot per SIMD).
usage: pmc_summary.py <fetch_dir> <write_dir> <frames_per_batch> <out.json> [<valu_dir>]"""
import collections
import csv
import glob
import json
import sys

STAGE = [("median15", "median", None), ("canny_nms", "canny_nms", None),
         ("prep_rows", "ccl_prep_rows", None), ("hough_vote", "hough_vote", None),
         ("warp_kernel", "warp", None),
         ("conv1_", "cnn_conv1", 128), ("_kernel<36, 36, 32", "cnn_conv2", 128), ("conv34_h2_kernel", "cnn_conv4", 128),
         ("_kernel<16, 16, 32", "cnn_conv3", 128), ("_kernel<14, 14, 9", "cnn_conv4", 128)]


def load(d, counter):
    f = glob.glob(d + "/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            for key, stage, _ in STAGE:
                if key in r["Kernel_Name"]:
                    agg[stage][0] += 1
                    agg[stage][1] += float(r["Counter_Value"])
    return agg


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    valu = load(sys.argv[5], "SQ_INSTS_VALU") if len(sys.argv) > 5 else {}
    mfma = load(sys.argv[5], "SQ_INSTS_MFMA") if len(sys.argv) > 5 else {}
    frames = int(sys.argv[3])
    out = {"_note": "KiB counters from rocprofv3 --pmc (separate passes); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                    "(gfx950 FETCH_SIZE halves wide coalesced reads); per frame of a %d-frame batch" % frames}
    for key, stage, chunk in STAGE:
        if stage not in fetch:
            continue
        per_dispatch_frames = min(chunk or frames, frames)
        fk = fetch[stage][1] / fetch[stage][0] / per_dispatch_frames
        wk = write[stage][1] / write[stage][0] / per_dispatch_frames
        out[stage] = dict(fetch_kib_per_frame=round(fk, 1), write_kib_per_frame=round(wk, 1),
                          hbm_bytes_per_frame=int((2 * fk + wk) * 1024), dispatches=fetch[stage][0])
        if stage in valu:
            # SQ_INSTS_VALU includes the MFMA instructions: report the two separately
            m = int(mfma[stage][1] / mfma[stage][0] / per_dispatch_frames) if stage in mfma else 0
            out[stage]["valu_wave_insts_per_frame"] = int(valu[stage][1] / valu[stage][0] / per_dispatch_frames) - m
            out[stage]["mfma_wave_insts_per_frame"] = m
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
