#!/bin/bash
# Driver-side helper: run tools/collect_profiles.sh on a GPU box and copy the summaries that are to be judged into profiles/
# (named per round): rocprofv3 kernel stats of the bench command and of the serial pass, the PMC traffic summary, the bench line.
set -e
cd "$(dirname "$0")/.."
R=${ROUND:-r06}
rm -rf gpurun_out/prof_bench gpurun_out/prof_serial gpurun_out/prof_stonefind gpurun_out/prof_cnn_bf16 gpurun_out/prof_cnn_q8 gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_valu
HEAD=$(git rev-parse --short HEAD)$(git diff --quiet || echo "+dirty")
/usr/local/graft/bin/gpurun --timeout 1100 -- "CK_HEAD=$HEAD ROUND=$R timeout -k 10 1000 bash tools/collect_profiles.sh" > /tmp/collect.log 2>&1 || { tail -20 /tmp/collect.log; exit 1; }
cp gpurun_out/pmc_traffic.json profiles/${R}_pmc_traffic.json
if [ -f gpurun_out/fetch_calib.txt ]; then cp gpurun_out/fetch_calib.txt profiles/${R}_fetch_calib.txt; else echo "no FETCH_SIZE calibration from this run: profiles/${R}_fetch_calib.txt not refreshed"; fi
grep -v amdgpu.ids gpurun_out/k1_content.txt > profiles/${R}_k1_content.txt || true
cp gpurun_out/margin_probe.txt profiles/${R}_margin_probe.txt || true
grep -E "bf16 +us per frame" gpurun_out/cnn_bf16.log > profiles/${R}_cnn_bf16_stages.txt || true
grep -E "f16q8 +us per frame" gpurun_out/cnn_q8.log > profiles/${R}_cnn_q8_stages.txt || true
python - "$R" <<'PY'
import csv, glob, json, sys
R = sys.argv[1]
for tag, d in (("bench", "prof_bench"), ("serial", "prof_serial"), ("stonefind", "prof_stonefind"), ("cnn_bf16", "prof_cnn_bf16"), ("cnn_q8", "prof_cnn_q8")):
    found = glob.glob("gpurun_out/%s/*/*kernel_stats.csv" % d)
    if not found:
        print("no kernel stats for", tag)
        continue
    src = found[0]
    rows = [r for r in csv.DictReader(open(src))]
    keep = [r for r in rows if "at::native" not in r["Name"] and "rocprim" not in r["Name"] and "__amd_rocclr" not in r["Name"]]
    with open("profiles/%s_%s_kernel_stats.csv" % (R, tag), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(keep)          # this package's kernels only (the renderer's torch kernels are dropped; percentages are of the whole run)
open("profiles/%s_stonefind_timing.json" % R, "w").write([l for l in open("gpurun_out/stonefind_timing.json") if l.startswith("{")][-1])
line = [l for l in open("gpurun_out/bench_full.json") if l.startswith("{")][-1]
open("profiles/%s_bench.json" % R, "w").write(line)
d = json.loads(line)
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["filter_pass_fused"]["frac"],
      d.get("fp32_chain", {}).get("value"), d.get("pcie_inclusive", {}).get("value"), d["cpu_baseline"]["value"])
print({k: round(v[2] if isinstance(v, list) else v["us_per_frame"], 2) for k, v in d["stages"].items()})
PY
