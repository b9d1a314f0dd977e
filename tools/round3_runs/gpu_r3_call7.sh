#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/c7; mkdir -p $O
for th in 16 32 64 128 256; do
  echo "---- CBA_PLAN_THREADS=$th"; CBA_PLAN_THREADS=$th timeout 300 python tools/create_timing.py 2>&1 | grep -E "^==|dealt" | grep -v cfg2 | awk '{print}' | sed -n 1,40p > $O/create_$th.log; grep "^==" $O/create_$th.log
done
