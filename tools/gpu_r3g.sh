#!/bin/bash
# fused close kernel, handle cache, 112-byte records again: full GPU suite + bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/r3g; mkdir -p $O
timeout 700 python -m pytest tests -m gpu -x -q --timeout=150 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
timeout 200 python bench.py --no-cpu --also cfg2,cfg3,cfg5 --steps 30 --warmup 6 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3g/bench.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print(d["ms_per_step"], {x:k[x]["avg_us"] for x in k}, d["final_rms_px"])
print({n:(v["ms_per_step"], v["final_rms_px"], v["roofline"].get("avg_launch_us")) for n,v in d["also"].items()})
PY
