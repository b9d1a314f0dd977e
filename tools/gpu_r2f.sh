#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=gpurun_out/r2f; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q --timeout=120 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
timeout 200 python tools/schur_split.py cfg4 0,16 > $O/split_cfg4.log 2>&1
CBA_PLAN_TIMING=1 timeout 300 python bench.py --no-cpu --also cfg3,cfg5 --steps 20 --warmup 4 > $O/bench.json 2> $O/bench.err
tail -8 $O/tests.log; grep -v "^k_schur" $O/split_cfg4.log; grep "^k_schur" $O/split_cfg4.log | tail -1; tail -c 1500 $O/bench.json
