#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/c17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_multi_device.py tests/test_constraints.py tests/test_stage_driver.py -m gpu -x -q --timeout=400 -k "not C420 and not C300 and not C200 and not C128" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
timeout 300 python bench.py --no-cpu --also cfg2,cfg3 --steps 40 --warmup 8 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c17/bench.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print(d["ms_per_step"], {x:k[x]["avg_us"] for x in k}, d["final_rms_px"], d["solve"]["nfev"], d["solve"]["cost"])
for n, v in d["also"].items(): print("   ", n, v.get("ms_per_step"), v.get("nfev"), v.get("final_rms_px"), v.get("error"))
PY
