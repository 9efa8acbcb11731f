"""print the headline fields of a bench.py JSON line read from stdin"""
import json
import sys
for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        print(d["n_gpus"], d["config"].get("lanes_per_gpu"), d["value"], d["ms_per_step"], d.get("host_ms_per_step"),
              d["move_sequence_ratio"], d["stone_grid_match_pct"], d["roofline"]["frac"])
