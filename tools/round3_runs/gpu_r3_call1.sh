#!/bin/bash
# round 3, first call: the producer-wave pair kernel (written at the end of round 2, never run) against the default and the three-buffer build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/exp; mkdir -p $O
run() {  # name, library ("" = the product's), CBA_GRID_MULT
  local lib="X_UNUSED=1"; [ -n "$2" ] && lib="CALISCOPE_BA_LIB=$GRAFT_REPO_ROOT/tools/exp/$2"
  env $lib CBA_GRID_MULT=$3 timeout 100 python bench.py --no-cpu --also "" --steps 30 --warmup 6 > $O/$1.json 2> $O/$1.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/exp/{sys.argv[1]}.json").read().strip().splitlines()[-1]); k = d["roofline"]["kernels"]
    print(sys.argv[1], d["ms_per_step"], "pairs", k["schur_pairs"]["avg_us"], "schur", k["schur"]["avg_us"], "build", k["build"]["avg_us"], "rms", d["final_rms_px"], {x: k[x]["avg_us"] for x in k})
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(f"gpurun_out/exp/{sys.argv[1]}.err").read()[-1500:])
PY
}
run default "" 2
export CBA_PLAN_REGION=256
run chunks192_3buffers libcba_depth3.so 2
run chunks192_producer_wave libcba_producer.so 2
run chunks192_producer_wave_g3 libcba_producer.so 3
