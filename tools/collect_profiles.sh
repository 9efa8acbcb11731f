#!/bin/bash
# Runs on the GPU box (tools/refresh_profiles.sh sends it there): rocprofv3 kernel-trace stats of the bench command,
# the same for the serial board path alone, then the PMC passes -- counters collected on their own, one rocprofv3 run per
# counter group, --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section).
# Results land in gpurun_out/prof_*; tools/refresh_profiles.sh copies the summaries to be judged into profiles/.
set -e
export TMPDIR=/tmp
O=gpurun_out
R=${ROUND:-r06}
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 bench.py --no-cpu-baseline --no-extras > $O/bench_prof.json 2> $O/prof_bench.err
echo "bench stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -- python3 tools/board_serial.py > $O/board_serial.log 2> $O/prof_serial.err
echo "serial stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_stonefind -- python3 tools/stonefind_timing.py --reps 3 --cpu 0 > $O/stonefind_prof.json 2> $O/prof_stonefind.err
echo "stonefind stats done"
# the per-launch traces are large and not needed once the stats exist (gpurun merges back at most 64 MiB)
find $O/prof_bench $O/prof_serial $O/prof_stonefind -name "*kernel_trace.csv" -delete 2>/dev/null || true
find $O/prof_bench $O/prof_serial $O/prof_stonefind -name "*.db" -delete 2>/dev/null || true
PMCARGS="--timed-only --steps 2 --warmup 1 --frames 256 --lanes 2"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py $PMCARGS > /dev/null 2> $O/pmc_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py $PMCARGS > /dev/null 2> $O/pmc_write.err
echo "write done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc_valu -- python3 bench.py $PMCARGS > /dev/null 2> $O/pmc_valu.err
echo "valu done"
find $O/pmc_fetch $O/pmc_write $O/pmc_valu -name "*.db" -delete 2>/dev/null || true
# summarise here: at the bench's own scale the raw counter tables are too large to travel back (gpurun merges 64 MiB)
python3 tools/pmc_summary.py $O/pmc_fetch $O/pmc_write 128 $O/pmc_traffic.json $O/pmc_valu > /dev/null
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_valu
# what FETCH_SIZE counts per load width, on a stream of known size (the file the traffic figures are read against): the
# micro-benchmark is built here (its binary is not tracked), and a failure of the step is said, not swallowed
if /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/micro/fetch_calib.hip -o tools/micro/fetch_calib 2> $O/fetch_calib_build.err \
   && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_calib -- tools/micro/fetch_calib > $O/fetch_calib.log 2>&1 \
   && python3 tools/pmc_kernels.py $O/pmc_calib read_b8x3 "read_wide<unsigned int>" "read_wide<HIP_vector_type<unsigned int, 2" "read_wide<HIP_vector_type<unsigned int, 4" > $O/fetch_calib.txt 2>&1; then
  echo "fetch calibration done"
else
  echo "FETCH_SIZE CALIBRATION SKIPPED (build or run failed: $O/fetch_calib_build.err, $O/fetch_calib.log)"; rm -f $O/fetch_calib.txt
fi
rm -rf $O/pmc_calib
# the headline bench line FIRST among what is left: nothing below may keep it from being produced
python bench.py > $O/bench_full.json 2> $O/bench_full.err
echo "bench done"
python tools/stonefind_timing.py > $O/stonefind_timing.json 2> $O/stonefind_timing.err
echo "stonefind timing done"
# the bf16 classifier (BASELINE config 5) and the opt-in F16Q8 one: kernel stats of the classifier alone, 128 frames per call.
# Profiles of optional modes: a failure is said, not fatal
for mode in bf16 f16q8; do
  tag=cnn_${mode/f16q8/q8}
  if rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python3 tools/cnn_modes.py 128 $mode > $O/$tag.log 2> $O/prof_$tag.err; then
    echo "$mode classifier stats done"
  else
    echo "CLASSIFIER STATS OF MODE $mode FAILED (see $O/prof_$tag.err): profiles/${R}_${tag}_* not refreshed"
  fi
  find $O/prof_$tag -name "*kernel_trace.csv" -delete 2>/dev/null || true
  find $O/prof_$tag -name "*.db" -delete 2>/dev/null || true
done
# K1 across content classes (exactness against a sort-based median included) and the margin probe of the reduced-precision modes
python tools/k1_content.py 5 > $O/k1_content.txt 2>&1 || echo "K1 CONTENT RUN FAILED (see $O/k1_content.txt)"
python tools/margin_probe.py 16 1 > $O/margin_probe.txt 2>&1 || echo "MARGIN PROBE FAILED OR BEYOND ITS GATE (see $O/margin_probe.txt)"
du -sh $O | tail -1
