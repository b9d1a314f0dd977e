#!/bin/bash
# round 4, call 19: the driver's round-end sequence on HEAD (host-side Python changed after the final profile run; the library did not): GPU suite, smoke, bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r4c19; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --timeout=400 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log | tail -3
timeout 120 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 600 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -3 $O/bench.time
python - <<'PY'
import json, os
d = json.loads(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4c19/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["setup_ms"], d["setup_ms_warm"], d["plan_wait_ms"], d["library_source_sha256"][:16], d["roofline"]["frac"], d["cpu_baseline"]["value"])
print(d["parity"]["aligned_pos"], d["parity"]["oracle_polish"]["scipy_moves_within_1e-6_and_gains_within_1e-12"], d["parity"]["gpu"])
for k, v in d["also"].items(): print(k, v["ms_per_step"], v["setup_ms"], v["setup_ms_warm"], v["plan_wait_ms"])
PY
