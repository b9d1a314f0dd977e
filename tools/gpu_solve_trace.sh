#!/bin/bash
# host wall-clock per primitive of cba_solve (CBA_SOLVE_TRACE=1) beside the device timers: where a bounded (cfg5) iteration spends its time
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/solvetrace; mkdir -p $O
CBA_SOLVE_TRACE=1 timeout 300 python bench.py --no-cpu --also ${TRACE_ALSO:-cfg5} --steps ${TRACE_STEPS:-20} --warmup 4 > $O/bench.json 2> $O/bench.err
grep -A24 "cba_solve trace" $O/bench.err | tail -120
python - <<'PY'
import json
d=json.loads(open("gpurun_out/solvetrace/bench.json").read().strip().splitlines()[-1])
for n,v in d["also"].items(): print(n, v["ms_per_step"], {x:y["avg_us"] for x,y in v["roofline"].get("kernels",{}).items()})
PY
