#!/bin/bash
# round 6, GPU call 7: staged small uploads in the set-up; full suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=30
O=$GRAFT_REPO_ROOT/gpurun_out/r6_run7; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --timeout=400 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -5 $O/tests.log
timeout 200 python tools/real_session_timing.py > $O/real_session.log 2>&1; cat $O/real_session.log
CBA_PLAN_TIMING=1 timeout 200 python tools/real_session_timing.py 2>&1 | grep "cba_create" | tail -30
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["config"]["workload"][:5], "ms_per_step", d["ms_per_step"], "setup_ms", d.get("setup_ms"), d.get("setup_ms_warm"), "plan_wait", d.get("plan_wait_ms"))'
for w in cfg4 cfg2; do
  timeout 300 python bench.py --no-cpu --no-first-call --workload $w --also "" --steps 20 --warmup 4 2> $O/bench_$w.err | python -c "$pick"
done
