#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/c14; mkdir -p $O
run() {
  local lib="X_UNUSED=1"; [ -n "$2" ] && lib="CALISCOPE_BA_LIB=$GRAFT_REPO_ROOT/tools/exp/$2"
  env $lib $3 timeout 200 python bench.py --no-cpu --also cfg5 --steps 16 --warmup 4 > $O/$1.json 2> $O/$1.err
  python - "$1" <<'PY'
import json, sys
try:
    d=json.loads(open(f"gpurun_out/c14/{sys.argv[1]}.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
    v=d["also"]["cfg5"]
    print(sys.argv[1], d["ms_per_step"], "build", k["build"]["avg_us"], "| cfg5", v.get("ms_per_step"), (v.get("roofline") or {}).get("kernels", {}).get("build", {}).get("avg_us"), v.get("final_rms_px"), v.get("error"))
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(f"gpurun_out/c14/{sys.argv[1]}.err").read()[-800:])
PY
}
run cs3072 "" X=1
run cs2048 libcba_cs2048.so X=1
run cs4096 libcba_cs4096.so X=1
run cs1536 libcba_cs1536.so X=1
run cs3072_grid3 "" CBA_GRID_MULT=3
