#!/bin/bash
# Driver-side helper: run tools/collect_profiles.sh on a GPU box, fold the PMC passes into profiles/r01_pmc_traffic.json,
# copy the rocprofv3 kernel stats, then re-run the three benches (they read the refreshed PMC file) and the 4K / VGA sizes.
set -e
cd "$(dirname "$0")/.."
rm -rf gpurun_out/prof_* gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_valu
/usr/local/graft/bin/gpurun --timeout 1100 -- 'timeout -k 10 1000 bash tools/collect_profiles.sh' > /tmp/collect.log 2>&1 || { tail -20 /tmp/collect.log; exit 1; }
python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write 64 profiles/r01_pmc_traffic.json gpurun_out/pmc_valu > /dev/null
for m in fp32 bf16 f16x2; do cp "$(ls gpurun_out/prof_$m/*/*kernel_stats.csv | head -1)" profiles/r01_bench_${m}_kernel_stats.csv; done
/usr/local/graft/bin/gpurun --timeout 900 -- 'for m in fp32 f16x2 bf16; do timeout -k 10 170 python bench.py --cnn $m > gpurun_out/bench_$m.json 2> gpurun_out/bench_$m.err && echo $m done; done; timeout -k 10 200 python bench.py --no-cpu-baseline --height 2160 --width 3840 --frames 64 > gpurun_out/bench_4k.json 2>/dev/null; timeout -k 10 200 python bench.py --no-cpu-baseline --height 480 --width 640 --frames 1024 > gpurun_out/bench_vga.json 2>/dev/null; echo sizes done' 2>&1 | tail -2
python - <<'PY'
import json
for m in ("fp32", "f16x2", "bf16", "4k", "vga"):
    l = [l for l in open("gpurun_out/bench_%s.json" % m) if l.startswith("{")][-1]
    d = json.loads(l)
    if m in ("fp32", "f16x2", "bf16"):
        open("profiles/r01_bench_%s.json" % m, "w").write(l)
    r = d["roofline"]
    print(m, d["value"], d["ms_per_step"], r["kernel"], r["frac"], r.get("valu_issue_frac"), r.get("mfma_busy_frac"),
          d["mfma_kernel"]["kernel"], d["mfma_kernel"]["frac"], (d.get("cpu_baseline") or {}).get("value"))
    print("   ", {k: round(v["us_per_frame"], 2) for k, v in d["stages"].items()})
PY
