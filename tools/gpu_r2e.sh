#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=gpurun_out/r2e; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_constraints.py -m gpu -x -q --timeout=120 > $O/tests_parity.log 2>&1; echo "rc=$?" >> $O/tests_parity.log
timeout 200 python tools/schur_split.py cfg4 0,16 > $O/split_cfg4.log 2>&1
CBA_PLAN_TIMING=1 timeout 200 python bench.py --no-cpu --also cfg3,cfg5 --steps 20 --warmup 4 > $O/bench.json 2> $O/bench.err
timeout 300 python -m pytest tests/test_multi_device.py -m gpu -x -v --timeout=60 > $O/tests_md.log 2>&1; echo "rc=$?" >> $O/tests_md.log
tail -4 $O/tests_parity.log; tail -12 $O/tests_md.log; grep -v "^k_schur" $O/split_cfg4.log; grep "^k_schur" $O/split_cfg4.log | tail -1
