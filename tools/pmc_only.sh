export TMPDIR=/tmp
O=gpurun_out
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --frames 64 --lanes 1 --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err
echo fetch done
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 1 --warmup 0 --frames 64 --lanes 1 --no-cpu-baseline > /dev/null 2> $O/pmc_write.err
echo write done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc_valu -- python3 bench.py --steps 1 --warmup 0 --frames 64 --lanes 1 --no-cpu-baseline > /dev/null 2> $O/pmc_valu.err
echo valu done
