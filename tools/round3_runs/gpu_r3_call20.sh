#!/bin/bash
# set-up after the threaded sorted-input pass, the arrays without zero-fill and the plan's parallel prologue; quick parity subset on the same build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/c20; mkdir -p $O
timeout 300 python tools/create_timing.py > $O/create_timing.log 2>&1; grep -E "^==|waited|took|cba_create" $O/create_timing.log | head -120
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=300 -k "evaluation_parity or limits or heavy or unsorted or order" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
