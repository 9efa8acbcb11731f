# PMC passes over one bench batch, summed per kernel family:  tools/pmc_stage.sh <kernel-name-substring>
export TMPDIR=/tmp
O=gpurun_out
K=${1:-canny_nms}
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rm -rf $O/pmc_st$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_st$i -- python3 bench.py --steps 1 --warmup 0 --frames 64 --lanes 1 --no-cpu-baseline > /dev/null 2> $O/pmc_st$i.err
  python3 tools/pmc_kernels.py $O/pmc_st$i $K
done
