#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r2i; mkdir -p $O
CBA_CHOL_TRACE=1 timeout 120 python tools/chol_trace.py > $O/chol_trace.log 2>&1
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o cfg4 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --also "" --steps 20 --warmup 4 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; tail -16 $O/chol_trace.log
