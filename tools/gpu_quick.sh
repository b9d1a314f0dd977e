#!/bin/bash
# quick check after a kernel change: the parity files that exercise every kernel + one bench line with timers
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/quick; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_constraints.py -m gpu -x -q --timeout=150 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log
timeout 200 python bench.py --no-cpu --also cfg2,cfg3,cfg5 --steps 30 --warmup 6 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/quick/bench.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print(d["ms_per_step"], {x:k[x]["avg_us"] for x in k}, d["final_rms_px"])
for n,v in d["also"].items(): print(n, v["ms_per_step"], v["final_rms_px"], {x:y["avg_us"] for x,y in v["roofline"].get("kernels",{}).items()})
PY
