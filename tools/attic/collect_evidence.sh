#!/bin/bash
# Round-3 evidence files for profiles/ (run on the GPU box; the instrumented libraries are built beforehand with
# tools/ab_variant.sh: k_canny.hip stop3 / stop4 (-DNMS_STOP), k_cnn.hip time (-DH2_DBG_TIME=1)).
export TMPDIR=/tmp
O=gpurun_out
{
  echo "# canny_nms_packed_kernel: SQ_INSTS_VALU over one 64-frame 1080p launch, whole kernel and builds that leave after phase 3 (staging + gradient + NMS) / phase 4 (+ tile-local union-find)"
  for lib in "" camkifu_amd/libck_hip_stop3.so camkifu_amd/libck_hip_stop4.so; do
    rm -rf $O/pmc_ev
    CK_HIP_LIB=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc_ev -- python3 tools/board_serial.py 64 1 > /dev/null 2> $O/pmc_ev.err
    echo "library: ${lib:-default}"; python3 tools/pmc_kernels.py $O/pmc_ev canny_nms
  done
  rm -rf $O/pmc_ev
} > $O/r03_nms_phases.txt 2>&1
{
  echo "# classifier kernels, 128-frame launches: LDS and matrix-pipe counters (tools/pmc_one.sh)"
  tools/pmc_one.sh "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY" 128 conv_mfma16_h2 conv34_h2
  tools/pmc_one.sh "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" 128 conv_mfma16_h2 conv34_h2 median_mfma
  tools/pmc_one.sh "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" 128 conv_mfma16_h2 conv34_h2 median_mfma
  echo "# fused conv1 + conv2: wall-clock phases per workgroup (-DH2_DBG_TIME=1)"
  CK_HIP_LIB=camkifu_amd/libck_hip_time.so python3 tools/board_serial.py 128 2 2>&1 | grep "conv2 phases"
  echo "# v_mfma_f32_16x16x32_f16 sustained rate (tools/micro/mfma_f16.hip)"
  tools/micro/mfma_f16
} > $O/r03_classifier_counters.txt 2>&1
