#!/bin/bash
# timing + check of the 2 x 2-pivot block factorisation (tools/chol_factor_bench/fac7.hip)
cd "$GRAFT_REPO_ROOT/tools/chol_factor_bench" || exit 1
[ -x fac7.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value fac7.hip -o fac7.bin
timeout 60 ./fac7.bin
