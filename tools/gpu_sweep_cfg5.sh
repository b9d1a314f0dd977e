#!/bin/bash
# plan parameters of the nine-parameter pair kernel on cfg5: cost of a chunk's gather (workgroup binding), region size
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/sweep5; mkdir -p $O
for cfg in "2.0 32" "1.0 32" "3.5 32" "2.0 16" "2.0 64" "2.0 128"; do
  set -- $cfg
  CBA_PLAN_COST_A=$1 CBA_PLAN_REGION=$2 timeout 200 python bench.py --no-cpu --workload cfg5 --also "" --steps 10 --warmup 3 > $O/b_$1_$2.json 2> $O/b_$1_$2.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/sweep5/b_$1_$2.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print("cost_a $1 region $2:", d["ms_per_step"], "pairs", k["schur_pairs"]["avg_us"], "schur", k["schur"]["avg_us"])
PY
done
