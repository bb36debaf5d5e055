#!/bin/bash
# CPU-side sanitizer run of the oracle (GPU ASan is not available on this pool): builds oracle/icp_oracle.c with
# -fsanitize=address,undefined and runs the oracle's own CPU tests against that build.
set -e
cd "$(dirname "$0")/.."
OUT=/tmp/o3d_oracle_san
mkdir -p $OUT
gcc -O1 -g -march=x86-64-v2 -ffp-contract=off -fno-fast-math -fopenmp -fPIC -std=gnu11 -fsanitize=address,undefined \
    -fno-omit-frame-pointer -shared -o $OUT/libicp_oracle.so oracle/icp_oracle.c -lm
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD=$(gcc -print-file-name=libasan.so) O3D_ORACLE_LIB=$OUT/libicp_oracle.so OMP_NUM_THREADS=4 \
    python3 -m pytest tests/test_oracle_golden.py tests/test_host_and_abi.py -q -x -m "not gpu"
