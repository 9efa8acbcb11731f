"""Where a request of the hold-off-aware fold spends its time inside the library, in the REAL pipeline (the bench's
hold-off-aware leg, two stones lanes running): bench.py with CK_PROFILE_HOST=1, the lap lines between the leg's markers
averaged per k_board_lines call.  usage: python tools/lazy_laps.py"""
import collections
import json
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline"], env=dict(os.environ, CK_PROFILE_HOST="1"),
                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
inside, laps, calls = False, collections.OrderedDict(), 0
for line in r.stderr.splitlines():
    if line.startswith("[bench] hold-off-aware leg"):
        inside = line.endswith("start")
        continue
    m = re.match(r"\[board_lines\] (.+?)\s+([\d.]+) ms", line)
    if m and inside:
        laps[m.group(1)] = laps.get(m.group(1), 0.0) + float(m.group(2))
        calls += m.group(1) == "ccl kernels"
d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
h = d["holdoff_aware"]
print("eager %.0f frames/s, hold-off-aware %.0f; %d k_board_lines calls in the timed steps, %.2f ms inside per call:" % (
    d["value"], h["value"], calls, sum(laps.values()) / max(calls, 1)))
for k, v in laps.items():
    print("    %-20s %.3f ms" % (k, v / max(calls, 1)))
print("host ms per step:", h["host_ms_per_step"])
