#!/bin/bash
# the round's last validation: full GPU suite, smoke and the bench line as the driver runs them, on the final library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/final3; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=400 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
timeout 120 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -3 $O/bench.time; tail -c 300 $O/bench.json
