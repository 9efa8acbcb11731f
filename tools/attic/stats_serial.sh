#!/bin/bash
# rocprofv3 --kernel-trace --stats of the serial board + stones pass; prints the per-kernel table (ns per launch)
export TMPDIR=/tmp
O=gpurun_out/prof_serial
rm -rf $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 tools/board_serial.py ${1:-128} ${2:-3} > /dev/null 2> $O.err
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_serial/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "anonymous namespace" in r["Name"] and "at::native" not in r["Name"]: print(r["Name"].replace("(anonymous namespace)::","")[:64].ljust(64), r['Calls'].rjust(4), ('%.1f' % (float(r['AverageNs']) / 1e3)).rjust(9), 'us', r['Percentage'])
PY
