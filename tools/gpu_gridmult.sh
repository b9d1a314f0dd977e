#!/bin/bash
# persistent workgroups per CU of the per-observation kernels (CBA_GRID_MULT)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/gm; mkdir -p $O
for g in 2 3 4; do
  CBA_GRID_MULT=$g timeout 200 python bench.py --no-cpu --also "" --steps 30 --warmup 6 > $O/b_$g.json 2> $O/b_$g.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/gm/b_$g.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print("grid mult $g:", d["ms_per_step"], {x:k[x]["avg_us"] for x in ("build","jv","schur","schur_pairs","backsub","build_reduce")})
PY
done
