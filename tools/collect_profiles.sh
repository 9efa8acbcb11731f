#!/bin/bash
# Runs on the GPU box: kernel-trace stats for the fp32 and bf16 bench, then the two PMC passes
# (FETCH_SIZE, WRITE_SIZE; counters collected on their own, --kernel-trace only).
# Results land in gpurun_out/prof_*; copy what is to be judged into profiles/.
set -e
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fp32 -- python3 bench.py --no-cpu-baseline --cnn fp32 > $O/bench_fp32_prof.json 2> $O/prof_fp32.err
echo "fp32 stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16 -- python3 bench.py --no-cpu-baseline --cnn bf16 > $O/bench_bf16_prof.json 2> $O/prof_bf16.err
echo "bf16 stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f16x2 -- python3 bench.py --no-cpu-baseline --cnn f16x2 > $O/bench_f16x2_prof.json 2> $O/prof_f16x2.err
echo "f16x2 stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --frames 64 --lanes 1 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 1 --warmup 0 --frames 64 --lanes 1 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err
echo "write done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc_valu -- python3 bench.py --steps 1 --warmup 0 --frames 64 --lanes 1 --no-cpu-baseline > /dev/null 2> $O/pmc_valu.err
echo "valu done"
python bench.py --cnn fp32 > $O/bench_fp32.json 2> $O/bench_fp32.err
echo "bench fp32 done"
python bench.py --cnn f16x2 > $O/bench_f16x2.json 2> $O/bench_f16x2.err
echo "bench f16x2 done"
python bench.py --cnn bf16 > $O/bench_bf16.json 2> $O/bench_bf16.err
echo "bench bf16 done"
