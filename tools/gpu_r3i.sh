#!/bin/bash
cd "$GRAFT_REPO_ROOT/tools/chol_factor_bench" || exit 1
echo "== one pivot at a time (production)"; timeout 60 ./fac7_old.bin
echo "== two pivots at a time"; timeout 60 ./fac7.bin
