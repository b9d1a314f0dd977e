#!/bin/bash
# experiment: the six-parameter pair kernel with 192-slot chunks (49 KB of LDS) and three workgroups per CU (launch bounds for 170 registers) against the default
# (320 slots, two per CU), three chunk buffers, and the producer-wave build.  Build the variants first (on the machine that has hipcc; tools/exp/ travels):
#   tools/build_exp_lib.sh exp2     -DCBA_SCHUNK6=192 -DCBA_NCD6=4
#   tools/build_exp_lib.sh exp3     -DCBA_SCHUNK6=192 -DCBA_NCD6=4 -DCBA_MINW6=3 -DCBA_PER_CU6=3
#   tools/build_exp_lib.sh depth3   -DCBA_SCHUNK6=192 -DCBA_NCD6=4 -DCBA_NBUF6=3
#   tools/build_exp_lib.sh producer -DCBA_SCHUNK6=192 -DCBA_NCD6=4 -DCBA_NBUF6=3 -DCBA_NPROD6=1 -DCBA_MINW6=3      (written at the end of round 2, not run yet)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/exp; mkdir -p $O
run() {  # name, library ("" = the product's), CBA_GRID_MULT
  local lib="X_UNUSED=1"; [ -n "$2" ] && lib="CALISCOPE_BA_LIB=$GRAFT_REPO_ROOT/tools/exp/$2"
  env $lib CBA_GRID_MULT=$3 timeout 100 python bench.py --no-cpu --also "" --steps 30 --warmup 6 > $O/$1.json 2> $O/$1.err
  python - "$1" <<'PY'
import json, sys
d = json.loads(open(f"gpurun_out/exp/{sys.argv[1]}.json").read().strip().splitlines()[-1]); k = d["roofline"]["kernels"]
print(sys.argv[1], d["ms_per_step"], "pairs", k["schur_pairs"]["avg_us"], "schur", k["schur"]["avg_us"], "build", k["build"]["avg_us"], "rms", d["final_rms_px"])
PY
}
[ -n "$ONLY_DEPTH3" ] || run default "" 2
export CBA_PLAN_REGION=256
[ -n "$ONLY_DEPTH3" ] || run chunks192_2perCU libcba_exp2.so 2
[ -n "$ONLY_DEPTH3" ] || run chunks192_3perCU libcba_exp3.so 3
run chunks192_3buffers libcba_depth3.so 2
[ -f tools/exp/libcba_producer.so ] && run chunks192_producer_wave libcba_producer.so 2
