#!/bin/bash
# end of round 3: the full GPU suite, smoke and the bench line exactly as the driver runs them, on the final library; set-up timing and the plan-thread
# sweep behind profiles/r03_setup_timing.txt; the reference's own session through optimize()
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/final2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=400 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
timeout 120 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -3 $O/bench.time; tail -c 300 $O/bench.json
timeout 200 python tools/create_timing.py > $O/create_timing.log 2>&1; grep -E "^==" $O/create_timing.log
timeout 300 python tools/plan_threads_sweep.py 2>&1 | grep -v "^cba_create\|^  plan" > $O/plan_threads.log; cat $O/plan_threads.log
timeout 200 python tools/real_session_timing.py > $O/real_session.log 2>&1; tail -5 $O/real_session.log
