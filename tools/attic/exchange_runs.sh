#!/bin/bash
# Several runs of the whole job on ONE box, alternating: without the exchange stage; with it over RCCL (one rank) on
# high-priority streams (the default); the same on normal-priority streams; and with the background model's context
# alone left at normal priority (forced_p_xs).  Gating test's size; frames/s and the host milliseconds per step of every stage.
# usage: tools/exchange_runs.sh out.txt
out=${1:-gpurun_out/exchange_runs.txt}
A="--timed-only --frames 256 --steps 12 --warmup 2"
: > $out
for round in 1 2 3 4 5; do
  echo "plain       $(python bench.py $A 2>/dev/null)" >> $out || exit 1
  echo "forced_prio $(python bench.py $A --force-exchange 2>/dev/null)" >> $out || exit 1
  echo "forced_norm $(CK_EXCHANGE_PRIORITY=0 python bench.py $A --force-exchange 2>/dev/null)" >> $out || exit 1
  echo "forced_p_xs $(CK_BG_PRIORITY=0 python bench.py $A --force-exchange 2>/dev/null)" >> $out || exit 1
done
python - $out <<'PY'
import json, sys
for l in open(sys.argv[1]):
    tag, js = l.split(None, 1)
    d = json.loads(js)
    h = d["host_ms_per_step"]
    print("%-12s %8.0f frames/s  step %.2f ms   exchange thread: pack %.2f collectives %.2f (gather %.2f bcast %.2f) band_model %.2f fold %.2f" % (
        tag, d["value"], d["ms_per_step"], h["pack"], h["collectives"], h["gather"], h["bcast"], h["band_model"], h["fold"]))
import collections
v = collections.defaultdict(list)
for l in open(sys.argv[1]):
    tag, js = l.split(None, 1)
    v[tag].append(json.loads(js)["value"])
for tag, x in v.items():
    print("%-12s mean %.0f  min %.0f  max %.0f" % (tag, sum(x) / len(x), min(x), max(x)))
PY
