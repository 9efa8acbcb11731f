#!/bin/bash
# Build a variant of libck_hip.so with extra -D flags for ONE source file, for same-box A/B runs:
#   tools/ab_variant.sh k_cnn.hip varA -DC3_WM=1 -DC4_WM=1   ->  camkifu_amd/libck_hip_varA.so
#   CK_HIP_LIB=camkifu_amd/libck_hip_varA.so python bench.py ...
set -e
cd "$(dirname "$0")/../camkifu_amd/csrc"
src=$1; tag=$2; shift 2
make -s -j8
mkdir -p _build/$tag
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function "$@" -c $src -o _build/$tag/${src%.hip}.o
objs=""
for o in _build/*.o; do
  b=$(basename $o)
  if [ "$b" == "${src%.hip}.o" ]; then objs="$objs _build/$tag/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libck_hip_$tag.so $objs
echo built camkifu_amd/libck_hip_$tag.so
