#!/bin/bash
# The profiling build of the library (NOT the product): -DCBA_PROFILING compiles the phase-clock variant of the pair kernel (CBA_SCHUR_CLOCK=1)
# and the stamps of the dense solve (CBA_CHOL_TRACE=1) in.  Use it through CALISCOPE_BA_LIB:
#     bash tools/build_profiling_lib.sh && CALISCOPE_BA_LIB=$PWD/caliscope_amd/libcaliscope_ba_prof.so CBA_SCHUR_CLOCK=1 python bench.py --no-cpu --also '' --steps 4 --warmup 1
cd "$(dirname "$0")/../caliscope_amd/csrc" || exit 1
exec /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -munsafe-fp-atomics -DCBA_PROFILING -Wall -Wno-unused-function \
  cba_lib.hip cba_solve.cpp -o ../libcaliscope_ba_prof.so -pthread -lrccl -lrocprofiler-sdk-roctx
