#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/c12; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q --timeout=400 -k "not C420 and not C300 and not C200" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
timeout 300 python bench.py --no-cpu --also cfg2,cfg3 --steps 40 --warmup 8 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c12/bench.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print(d["ms_per_step"], d["final_rms_px"], d["solve"]["nfev"], {x:k[x]["avg_us"] for x in k}, {n: (v["ms_per_step"], v["nfev"], v["final_rms_px"]) for n, v in d["also"].items()})
PY
