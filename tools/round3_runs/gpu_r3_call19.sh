#!/bin/bash
# validation of the plan thread and of constraint components beyond the LDS copy; set-up timing; host time per primitive of cfg2 / cfg4 solves
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/c19; mkdir -p $O
timeout 600 python -m pytest tests/test_constraints.py tests/test_scenarios.py tests/test_stage_driver.py -m gpu -x -q --timeout=300 > $O/tests_a.log 2>&1; echo "rc=$?" >> $O/tests_a.log; tail -4 $O/tests_a.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=300 -k "step_parity or limits or lds_tile or variants or point_ordered" > $O/tests_b.log 2>&1; echo "rc=$?" >> $O/tests_b.log; tail -4 $O/tests_b.log
timeout 200 python tools/create_timing.py > $O/create_timing.log 2>&1; grep -E "^==|waited|took" $O/create_timing.log
CBA_SOLVE_TRACE=1 timeout 200 python bench.py --no-cpu --workload cfg2 --also "" --steps 20 --warmup 4 > $O/bench_cfg2.json 2> $O/bench_cfg2.err; grep -B2 -A12 "cba_solve trace" $O/bench_cfg2.err | tail -60
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c19/bench_cfg2.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["config"].get("solves_in_timed_region"), {x:(y["avg_us"], y["launches"]) for x,y in d["roofline"].get("kernels",{}).items()})
PY
