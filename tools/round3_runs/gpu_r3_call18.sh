#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/c18; mkdir -p $O
run() {  # name lib env...
  local name=$1 lib=$2; shift 2
  local L="X_UNUSED=1"; [ -n "$lib" ] && L="CALISCOPE_BA_LIB=$GRAFT_REPO_ROOT/tools/exp/$lib"
  env $L "$@" timeout 100 python bench.py --no-cpu --also "" --steps 24 --warmup 6 > $O/$name.json 2> $O/$name.err
  python - "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/c18/{sys.argv[1]}.json").read().strip().splitlines()[-1]); k = d["roofline"]["kernels"]
    print(sys.argv[1], d["ms_per_step"], "pairs", k["schur_pairs"]["avg_us"], "schur", k["schur"]["avg_us"], "nfev", d["solve"]["nfev"], "grid", d["engine"]["schur_grid"], "stream", d["engine"]["schur_stream_len"])
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(f"gpurun_out/c18/{sys.argv[1]}.err").read()[-600:])
PY
}
run default "" X=1
run occ3_g3 libcba_occ3.so CBA_GRID_MULT=3
run occ3_g3_r64 libcba_occ3.so CBA_GRID_MULT=3 CBA_PLAN_REGION=64
run occ4_g4 libcba_occ4.so CBA_GRID_MULT=4
run occ4_g4_r64 libcba_occ4.so CBA_GRID_MULT=4 CBA_PLAN_REGION=64
run occ2s libcba_occ2s.so X=1
