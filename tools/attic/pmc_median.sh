export TMPDIR=/tmp
O=gpurun_out
python3 tools/median_only.py 32 3
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $O/pmc_med1 -- python3 tools/median_only.py 32 1 > /dev/null 2> $O/pmc_med1.err
echo pass1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_med2 -- python3 tools/median_only.py 32 1 > /dev/null 2> $O/pmc_med2.err
echo pass2
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_med3 -- python3 tools/median_only.py 32 1 > /dev/null 2> $O/pmc_med3.err
echo pass3
python3 tools/pmc_kernels.py $O/pmc_med1 median15
python3 tools/pmc_kernels.py $O/pmc_med2 median15
python3 tools/pmc_kernels.py $O/pmc_med3 median15
