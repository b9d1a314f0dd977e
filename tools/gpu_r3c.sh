#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3c; mkdir -p $O
CBA_SCHUR_CLOCK=2 timeout 120 python tools/newton_probe.py cfg4 1 2> $O/clock.log
grep -c "^wg " $O/clock.log
