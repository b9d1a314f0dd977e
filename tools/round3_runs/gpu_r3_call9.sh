#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/c9; mkdir -p $O
run() {
  env CBA_GRID_MULT=$1 CBA_BACKSUB_WGS=$2 timeout 100 python bench.py --no-cpu --also "" --steps 30 --warmup 6 > $O/g$1_b$2.json 2> $O/g$1_b$2.err
  python - $1 $2 <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/c9/g{sys.argv[1]}_b{sys.argv[2]}.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print("grid_mult", sys.argv[1], "backsub_wgs", sys.argv[2], d["ms_per_step"], {x:k[x]["avg_us"] for x in ("build","jv","schur","backsub","schur_pairs")})
PY
}
run 2 2; run 3 2; run 4 2; run 2 3; run 2 4; run 2 6; run 3 4
