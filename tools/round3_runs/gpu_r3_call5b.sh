#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/c5; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=400 -k "C420 or C300 or C200" > $O/tests_b.log 2>&1; echo "rc=$?" >> $O/tests_b.log
tail -5 $O/tests_b.log
timeout 900 python -m pytest tests -m gpu -x -q --timeout=400 --deselect tests/test_gpu_parity.py > $O/tests_rest.log 2>&1; echo "rc=$?" >> $O/tests_rest.log
tail -5 $O/tests_rest.log
