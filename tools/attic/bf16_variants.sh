#!/bin/bash
# A/B builds of the bf16 classifier kernels on the GPU box: each argument is a set of -D flags for k_cnn_bf16.hip
# ("" = the defaults); prints tools/cnn_modes.py's bf16 line per build.  usage: tools/bf16_variants.sh "" "-DBF_C2_D=4" ...
cd "$(dirname "$0")/../camkifu_amd/csrc" || exit 1
for flags in "$@"; do
    rm -f _build/k_cnn_bf16.o
    make EXTRA="$flags" > /dev/null 2>&1 || { echo "build failed: $flags"; exit 1; }
    echo "== k_cnn_bf16.hip flags: [$flags]"
    (cd ../.. && timeout -k 10 120 python tools/cnn_modes.py 128 bf16 2>&1 | grep "us per")
done
rm -f _build/k_cnn_bf16.o
make > /dev/null 2>&1
