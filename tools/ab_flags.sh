#!/bin/bash
# A/B builds of the whole library on the GPU box: each argument is a set of -D flags for EVERY source ("" = the defaults;
# the object directory is removed before each build, so a flag that lives in any .hip / .cpp file takes effect);
# runs the command in $AB_CMD per build.  usage: AB_CMD="python tools/board_only_4k.py" tools/ab_flags.sh "" "-DCCL_LIST_BLOCKS=128"
cd "$(dirname "$0")/../camkifu_amd/csrc" || exit 1
for flags in "$@"; do
    rm -rf _build
    make -j8 EXTRA="$flags" > /dev/null 2>&1 || { echo "build failed: $flags"; exit 1; }
    echo "== flags: [$flags]"
    (cd ../.. && timeout -k 10 200 $AB_CMD 2>&1 | tail -2)
done
rm -rf _build
make -j8 > /dev/null 2>&1
