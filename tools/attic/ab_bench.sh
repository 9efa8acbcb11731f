#!/bin/bash
# Same-box A/B of kernel variants: runs bench.py once per library and prints fps + the per-stage us/frame.
#   tools/ab_bench.sh "" camkifu_amd/libck_hip_w1.so ...     ("" = the default library)
for lib in "$@"; do
  CK_HIP_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline ${AB_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('${lib:-default}', d['value'], d['ms_per_step'], {k:round(v['us_per_frame'],2) for k,v in d['stages'].items() if k.startswith(('cnn','median','ccl','canny','warp','mog2','hough','contour'))})"
done
