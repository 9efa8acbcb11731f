#!/bin/bash
# A/B on ONE box: the whole job without the exchange stage, with it on the exchange thread's own stream, and with it on
# torch's default stream (the round-3 behaviour); interleaved, two rounds.  usage: tools/ab_exchange.sh out.txt
out=${1:-gpurun_out/ab_exchange.txt}
: > $out
for round in 1 2; do
  echo "plain $(python bench.py --timed-only --steps 30 --warmup 3 2>/dev/null)" >> $out || exit 1
  echo "own_stream $(python bench.py --timed-only --steps 30 --warmup 3 --force-exchange 2>/dev/null)" >> $out || exit 1
  echo "default_stream $(CK_EXCHANGE_DEFAULT_STREAM=1 python bench.py --timed-only --steps 30 --warmup 3 --force-exchange 2>/dev/null)" >> $out || exit 1
done
cat $out
