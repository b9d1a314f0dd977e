#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/c24; mkdir -p $O
timeout 300 python bench.py --no-cpu --also cfg2,cfg3 --steps 40 --warmup 8 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c24/bench.json").read().strip().splitlines()[-1])
print("cfg4", d["ms_per_step"], {x:y["avg_us"] for x,y in d["roofline"]["kernels"].items()})
for n,v in d["also"].items(): print(n, v["ms_per_step"], v.get("final_rms_px"))
PY
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=200 -k "step_parity" > $O/tests.log 2>&1; tail -2 $O/tests.log
