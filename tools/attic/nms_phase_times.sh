#!/bin/bash
# Wall time of the NMS kernel per phase: builds that leave the kernel after phase n (-DNMS_STOP=n: 1 staging, 2 + gradient,
# 3 + NMS, 4 + tile-local union-find; results are then wrong) under rocprofv3 --kernel-trace --stats, 64-frame 1080p launches.
#   tools/nms_phase_times.sh [libs...]     default: the matrix-pipe variants mstop1..4, the whole kernel, the packed kernel
export TMPDIR=/tmp
libs=${@:-"camkifu_amd/libck_hip_mstop1.so camkifu_amd/libck_hip_mstop2.so camkifu_amd/libck_hip_mstop3.so camkifu_amd/libck_hip_mstop4.so default camkifu_amd/libck_hip_nmspk.so"}
for lib in $libs; do
  O=gpurun_out/prof_nms
  rm -rf $O
  if [ "$lib" == "default" ]; then unset CK_HIP_LIB; else export CK_HIP_LIB=$lib; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/canny_only.py 64 3 > /dev/null 2> $O.err || { echo "FAILED $lib"; tail -3 $O.err; exit 1; }
  python3 - "$lib" <<'PY'
import csv, glob, sys
f = glob.glob('gpurun_out/prof_nms/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "canny_nms" in r["Name"]: print(sys.argv[1].ljust(44), r["Name"].split("(")[0][-28:], "calls", r["Calls"], "avg %.1f us per 64-frame launch = %.3f us per frame" % (float(r["AverageNs"]) / 1e3, float(r["AverageNs"]) / 64e3))
PY
  rm -rf $O
done
