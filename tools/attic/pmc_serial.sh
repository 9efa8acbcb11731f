#!/bin/bash
# PMC passes over the SERIAL board + stones pass (tools/board_serial.py <frames> <reps>), per kernel family:
#   tools/pmc_serial.sh <frames> <reps> <kernel-name-substring> [more substrings ...]
# One rocprofv3 run per counter group, --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section).
export TMPDIR=/tmp
O=gpurun_out
F=${1:-64}; R=${2:-1}; shift 2
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rm -rf $O/pmc_se$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_se$i -- python3 tools/board_serial.py $F $R > /dev/null 2> $O/pmc_se$i.err
  python3 tools/pmc_kernels.py $O/pmc_se$i "$@"
  find $O/pmc_se$i -name "*.db" -delete 2>/dev/null || true
done
