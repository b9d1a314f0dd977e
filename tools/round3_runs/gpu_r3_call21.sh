#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/c21; mkdir -p $O
timeout 300 python tools/create_timing.py > $O/create_timing.log 2>&1; grep -E "^==|cba_create|plan:" $O/create_timing.log | sed -n 30,120p
