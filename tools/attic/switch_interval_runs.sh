#!/bin/bash
# One box: the whole bench (eager headline + the hold-off-aware leg) with the interpreter's default thread switch interval
# (5 ms) and with shorter ones.  usage: tools/switch_interval_runs.sh out.txt
out=${1:-gpurun_out/switch_interval.txt}
: > $out
for round in 1 2; do
  for si in "" 0.001 0.0002; do
    CK_SWITCH_INTERVAL=$si python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read())
h = d['holdoff_aware']
print('interval ${si:-default}: eager %.0f frames/s  hold-off-aware %.0f (%.1f %% of the records, %.2f calls per batch, fold_board %.2f ms per step)' % (
    d['value'], h['value'], h['board_records_computed_pct'], h['board_fetch_calls_per_batch'], h['host_ms_per_step']['fold_board']))" >> $out || exit 1
  done
done
cat $out
