#!/usr/bin/env python3
"""Turn the rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, optionally SQ_INSTS_VALU + SQ_INSTS_MFMA; separate runs,
--kernel-trace only) into per-stage HBM traffic per frame.

Units and the gfx950 correction, as MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes: both counters are KiB;
WRITE_SIZE is exact; FETCH_SIZE reports exactly HALF of the bytes of a streaming read.  The guide documents the halving
for 16-byte-per-lane streams and calls other widths uncalibrated; round 4 calibrated them (tools/micro/fetch_calib.hip,
profiles/r04_fetch_calib.txt: a 512 MiB stream reads 262 150 KiB whether it is fetched as single bytes at a 3-byte
stride, dwords, 8 or 16 bytes per lane), so the x2 is applied to EVERY kernel.  Rounds 1-3 doubled only the 16-byte
streams and so under-counted the median and the NMS kernel (their "0.77 x / 0.92 x of the algorithmic bytes" were 1.26 x /
1.29 x).  Both the raw and the corrected figure are written, with the run they come from (frames per batch, lanes), so
that nobody has to take the correction on trust.
usage: pmc_summary.py <fetch_dir> <write_dir> <frames_per_batch> <out.json> [<valu_dir>]"""
import collections
import csv
import glob
import json
import sys

# (kernel-name substring, stage, frames per dispatch cap, loads are 16 B per lane)
# (last column: double FETCH_SIZE -- true for every load width on gfx950, see the module docstring)
STAGE = [("median_mfma_kernel<15>", "median", None, True), ("canny_nms", "canny_nms", None, True),
         ("prep_runs", "ccl_prep_runs", None, True), ("link_runs", "ccl_link_runs", None, True),
         ("border_runs", "ccl_border_runs", None, True),
         ("prep_rows", "ccl_prep_rows", None, True), ("border_list", "ccl_border_list", None, True),
         ("hough_vote", "hough_vote", None, True), ("warp_kernel", "warp", None, True),
         ("mog2_run_kernel", "mog2", None, True),
         ("conv_mfma16_h2_kernel", "cnn_conv2", 128, True), ("conv34_h2_kernel", "cnn_conv4", 128, True),
         ("conv12_bf16_kernel", "cnn_conv2_bf16", 128, True), ("conv34_bf16_kernel", "cnn_conv4_bf16", 128, True),
         ("fc1_h2_kernel", "cnn_fc1", None, True)]


# SURVEY.md 8(d) per 1080p frame: median 2 x 3WH, NMS 3WH + WH, warp 433 200 + the source quad (<= 3WH), background model 433 200 + 1 444
ALGORITHMIC = {"median": 12441600, "canny_nms": 8294400, "warp": 433200 + 6220800, "mog2": 433200 + 1444}


def load(d, counter):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            for key, stage, _, _ in STAGE:
                if key in r["Kernel_Name"]:
                    agg[stage][0] += 1
                    agg[stage][1] += float(r["Counter_Value"])
                    break
    return agg


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    valu = load(sys.argv[5], "SQ_INSTS_VALU") if len(sys.argv) > 5 else {}
    mfma = load(sys.argv[5], "SQ_INSTS_MFMA") if len(sys.argv) > 5 else {}
    frames = int(sys.argv[3])
    import datetime
    import os
    out = {"_head": os.environ.get("CK_HEAD", "unknown"), "_date": datetime.date.today().isoformat(),
           "_source": "rocprofv3 --pmc, one pass per counter group, the bench's own shape: `bench.py --timed-only --frames 256 --lanes 2 "
                      "--warmup 1 --steps 2` = %d frames per launch, two lanes cycling 1.59 GB of frames (inputs come from HBM, not "
                      "from the Infinity Cache), tools/collect_profiles.sh; KiB counters; per frame" % frames,
           "_correction": "hbm_bytes_raw = (FETCH_SIZE + WRITE_SIZE) * 1024; hbm_bytes_corrected = (2 FETCH_SIZE + WRITE_SIZE) * 1024 for "
                          "every kernel: FETCH_SIZE reports half the bytes of a stream at every load width (MI355X_MICROARCH.md for 16 B "
                          "per lane; profiles/r04_fetch_calib.txt for 1, 4 and 8 B)"}
    for key, stage, chunk, wide in STAGE:
        if stage not in fetch or stage not in write:
            continue
        per = min(chunk or frames, frames)
        fk = fetch[stage][1] / fetch[stage][0] / per
        wk = write[stage][1] / write[stage][0] / per
        out[stage] = dict(fetch_kib_per_frame=round(fk, 1), write_kib_per_frame=round(wk, 1), wide_loads=wide,
                          hbm_bytes_raw=int((fk + wk) * 1024), hbm_bytes_corrected=int(((2 if wide else 1) * fk + wk) * 1024),
                          dispatches=fetch[stage][0])
        if stage in ALGORITHMIC:
            out[stage]["algorithmic_bytes"] = ALGORITHMIC[stage]
            out[stage]["counter_over_algorithmic"] = round(out[stage]["hbm_bytes_corrected"] / ALGORITHMIC[stage], 3)
        if stage in valu:
            m = int(mfma[stage][1] / mfma[stage][0] / per) if stage in mfma else 0      # SQ_INSTS_VALU includes the MFMAs
            out[stage]["valu_wave_insts_per_frame"] = int(valu[stage][1] / valu[stage][0] / per) - m
            out[stage]["mfma_wave_insts_per_frame"] = m
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
