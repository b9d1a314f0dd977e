#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/c4; mkdir -p $O
run() {  # name, library, CBA_GRID_MULT, extra env
  local lib="X_UNUSED=1"; [ -n "$2" ] && lib="CALISCOPE_BA_LIB=$GRAFT_REPO_ROOT/tools/exp/$2"
  env $lib CBA_GRID_MULT=$3 $4 timeout 100 python bench.py --no-cpu --also "" --steps 30 --warmup 6 > $O/$1.json 2> $O/$1.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/c4/{sys.argv[1]}.json").read().strip().splitlines()[-1]); k = d["roofline"]["kernels"]
    print(sys.argv[1], d["ms_per_step"], "pairs", k["schur_pairs"]["avg_us"], "schur", k["schur"]["avg_us"], "nfev", d["solve"]["nfev"], "rms", d["final_rms_px"], "cost", d["solve"]["cost"])
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(f"gpurun_out/c4/{sys.argv[1]}.err").read()[-1200:])
PY
}
run default "" 2 X=1
run chunks192 libcba_exp2.so 2 CBA_PLAN_REGION=256
run producer_r256 libcba_producer.so 2 CBA_PLAN_REGION=256
run producer_r32 libcba_producer.so 2 X=1
run producer_r128 libcba_producer.so 2 CBA_PLAN_REGION=128
