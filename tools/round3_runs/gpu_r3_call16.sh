#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/c16; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=400 -k "(evaluation_parity or step_parity or point_ordered or camera_table or mixed_fisheye or ragged or limits or converged or cba_solve) and not C420 and not C300 and not C200 and not C128" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
run() {
  env $2 timeout 300 python bench.py --no-cpu --also cfg5 --steps 20 --warmup 5 > $O/$1.json 2> $O/$1.err
  python - $1 <<'PY'
import json, sys
try:
    d=json.loads(open(f"gpurun_out/c16/{sys.argv[1]}.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]; v=d["also"]["cfg5"]; k5=(v.get("roofline") or {}).get("kernels", {})
    print(sys.argv[1], d["ms_per_step"], {x:k[x]["avg_us"] for x in ("build","jv","backsub")}, d["solve"]["nfev"], d["solve"]["cost"], "| cfg5", v.get("ms_per_step"), {x:k5[x]["avg_us"] for x in ("build","jv","backsub") if x in k5}, v.get("final_rms_px"), v.get("error"))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(f"gpurun_out/c16/{sys.argv[1]}.err").read()[-800:])
PY
}
run both X=1
run no_jv CBA_JV_CS=0
run no_backsub CBA_BACKSUB_CS=0
