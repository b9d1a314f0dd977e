#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/c8; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_constraints.py tests/test_multi_device.py -m gpu -x -q --timeout=400 -k "not C420 and not C300 and not C200" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
for mode in inverse subst; do
CBA_CHOL_BACKWARD=$mode timeout 300 python bench.py --no-cpu --also cfg2,cfg5 --steps 20 --warmup 5 > $O/bench_$mode.json 2> $O/bench_$mode.err
python - $mode <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/c8/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print(sys.argv[1], d["ms_per_step"], "chol", k["cholesky_solve"]["avg_us"], d["final_rms_px"], d["solve"]["nfev"], d["solve"]["cost"])
for n, v in d["also"].items(): print("   ", n, v.get("ms_per_step"), v.get("nfev"), v.get("final_rms_px"), (v.get("roofline") or {}).get("kernels", {}).get("cholesky_solve"), v.get("error"))
PY
done
