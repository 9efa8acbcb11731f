export TMPDIR=/tmp
for k in 1 2 3 4; do
  rm -rf gpurun_out/pmc_stop$k
  CK_HIP_LIB=camkifu_amd/libck_hip_stop$k.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_stop$k -- python3 tools/board_serial.py 64 1 > /dev/null 2> gpurun_out/pmc_stop$k.err
  echo stop$k; python3 tools/pmc_kernels.py gpurun_out/pmc_stop$k canny_nms
  rm -rf gpurun_out/pmc_stop$k
done
