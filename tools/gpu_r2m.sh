#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=gpurun_out/r2m; mkdir -p $O
CBA_SCHUR=reg2 timeout 100 python tools/schur_split.py cfg4 0 > $O/split.log 2>&1
timeout 400 python -m pytest tests/test_multi_device.py tests/test_gpu_parity.py tests/test_constraints.py -m gpu -x -q --timeout=150 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
CBA_SCHUR=lds timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=150 -k "step_parity or evaluation or converged" > $O/tests_lds.log 2>&1; echo "rc=$?" >> $O/tests_lds.log
cat $O/split.log; tail -4 $O/tests.log; tail -4 $O/tests_lds.log
