#!/bin/bash
# timing + check of the block factorisation variants (tools/chol_factor_bench): fac8 = two 16-wide stages (adopted), fac7 = 2 x 2 pivots
cd "$GRAFT_REPO_ROOT/tools/chol_factor_bench" || exit 1
for v in ${FAC_VARIANTS:-fac8}; do
  [ -x $v.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value $v.hip -o $v.bin
  echo "== $v"; timeout 60 ./$v.bin
done
