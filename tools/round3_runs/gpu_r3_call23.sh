#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/c23; mkdir -p $O
timeout 300 python tools/plan_swap_probe.py > $O/swap.log 2>&1; cat $O/swap.log | tail -12
CBA_PLAN=swap timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=200 -k "step_parity or converged or seam or handle" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -3 $O/tests.log
