#!/bin/bash
# one PMC pass over the serial board + stones pass: tools/pmc_one.sh "<counters>" <frames> <kernel substrings...>
export TMPDIR=/tmp
C=$1; F=$2; shift 2
rm -rf gpurun_out/pmc_one
rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_one -- python3 tools/board_serial.py $F 1 > /dev/null 2> gpurun_out/pmc_one.err
python3 tools/pmc_kernels.py gpurun_out/pmc_one "$@"
rm -rf gpurun_out/pmc_one
