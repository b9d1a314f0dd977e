#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/c5; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=400 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -15 $O/tests.log
timeout 100 python bench.py --no-cpu --also "" --steps 30 --warmup 6 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c5/bench.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print(d["ms_per_step"], {x:k[x]["avg_us"] for x in k}, d["final_rms_px"], d["solve"]["nfev"])
PY
