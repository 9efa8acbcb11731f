#!/bin/bash
# PMC passes over the classifier alone (tools/cnn_modes.py <frames> <mode>), summed per kernel family:
#   tools/pmc_cnn.sh <mode> <frames> <kernel-name-substring>...
export TMPDIR=/tmp
O=gpurun_out
M=${1:-bf16}; F=${2:-64}; shift 2
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rm -rf $O/pmc_cnn$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_cnn$i -- python3 tools/cnn_modes.py $F $M > /dev/null 2> $O/pmc_cnn$i.err
  python3 tools/pmc_kernels.py $O/pmc_cnn$i "$@"
  rm -rf $O/pmc_cnn$i
done
