#!/bin/bash
# A/B builds of the F16Q8 classifier kernels on the GPU box: each argument is a set of -D flags for k_cnn_q8.hip
# ("" = the defaults); prints tools/cnn_modes.py's f16q8 line per build.  usage: tools/q8_variants.sh "" "-DQ8_DBG_TIME=1" ...
cd "$(dirname "$0")/../camkifu_amd/csrc" || exit 1
for flags in "$@"; do
    rm -f _build/k_cnn_q8.o
    make EXTRA="$flags" > /dev/null 2>&1 || { echo "build failed: $flags"; exit 1; }
    echo "== k_cnn_q8.hip flags: [$flags]"
    (cd ../.. && timeout -k 10 120 python tools/cnn_modes.py 128 ${Q8_MODES:-f16q8} 2>&1 | grep "us per\|residency")
done
rm -f _build/k_cnn_q8.o
make > /dev/null 2>&1
