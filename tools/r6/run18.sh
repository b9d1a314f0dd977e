#!/bin/bash
# round 6, GPU call 18: the GPU suite three times over on one box (no -x): does any tolerance sit at the edge of the run-to-run spread of the atomics' summation order?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=30
O=$GRAFT_REPO_ROOT/gpurun_out/r6_run18; mkdir -p $O
for i in 1 2 3; do
  timeout 1200 python -m pytest tests -m gpu -q --timeout=400 -p no:cacheprovider > $O/tests_$i.log 2>&1; echo "rc=$?" >> $O/tests_$i.log
  tail -4 $O/tests_$i.log; grep -E "^FAILED|^ERROR" $O/tests_$i.log
done
