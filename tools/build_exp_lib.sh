#!/bin/bash
# Build a VARIANT of the library for a timing experiment (never the product: caliscope_amd/build.py builds that):
#   tools/build_exp_lib.sh NAME -DCBA_SCHUNK6=192 -DCBA_NCD6=4 -DCBA_NBUF6=3      -> tools/exp/libcba_NAME.so
# and point CALISCOPE_BA_LIB at it.  The macros live in tools/pair_kernel_experiments/cba_kernels_experiments.patch (apply it to a scratch copy first).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
mkdir -p "$R/tools/exp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -munsafe-fp-atomics -Wno-unused-function "$@" \
  "$R/caliscope_amd/csrc/cba_lib.hip" "$R/caliscope_amd/csrc/cba_solve.cpp" -o "$R/tools/exp/libcba_$name.so" -pthread -lrccl -lrocprofiler-sdk-roctx
echo "$R/tools/exp/libcba_$name.so"
