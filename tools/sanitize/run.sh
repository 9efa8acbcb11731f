#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side native code: the product's host geometry (ck_host_geom.cpp, ck_stonegeom.cpp) and ordered
# halves (ck_fold.cpp) with fuzz harnesses, the oracle's C restatement through its own quick self-checks, and the worker pool of the
# host loops (ck_pool.h / ck_pool.cpp) under ThreadSanitizer as well (leaks are not checked there: the pool's threads live as long as the process).  GPU ASan is not available on
# the pool; this is the sanitizer coverage the repository has.
set -e
cd "$(dirname "$0")/../.."
mkdir -p tools/sanitize/_build
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off \
    -I include tools/sanitize/host_geom_fuzz.cpp camkifu_amd/csrc/ck_host_geom.cpp -o tools/sanitize/_build/host_geom_fuzz
ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=halt_on_error=1 tools/sanitize/_build/host_geom_fuzz
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off \
    -I include tools/sanitize/fold_fuzz.cpp camkifu_amd/csrc/ck_fold.cpp -o tools/sanitize/_build/fold_fuzz
ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=halt_on_error=1 tools/sanitize/_build/fold_fuzz
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off \
    -I include tools/sanitize/stonegeom_fuzz.cpp camkifu_amd/csrc/ck_stonegeom.cpp camkifu_amd/csrc/ck_host_geom.cpp -o tools/sanitize/_build/stonegeom_fuzz
ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=halt_on_error=1 tools/sanitize/_build/stonegeom_fuzz
gcc -O1 -g -std=gnu11 -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -fopenmp \
    -I oracle tools/sanitize/oracle_smoke.c oracle/ora_filter.c oracle/ora_contours.c oracle/ora_geom.c oracle/ora_mog2.c \
    oracle/ora_cnn.c oracle/ora_color.c -lm -o tools/sanitize/_build/oracle_smoke
ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=halt_on_error=1 OMP_NUM_THREADS=4 tools/sanitize/_build/oracle_smoke
for san in thread address,undefined; do
  g++ -O1 -g -std=c++17 -fsanitize=$san -fno-omit-frame-pointer tools/sanitize/pool_stress.cpp camkifu_amd/csrc/ck_pool.cpp \
      -lpthread -o tools/sanitize/_build/pool_stress_${san%%,*}
  TSAN_OPTIONS=halt_on_error=1 ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=halt_on_error=1 tools/sanitize/_build/pool_stress_${san%%,*}
done
