#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/c27; mkdir -p $O
run() {
  local label=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu --workload cfg5 --also "" --steps 8 --warmup 2 > $O/bench_$label.json 2> $O/bench_$label.err
  python - "$label" <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/c27/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
k=d["roofline"]["kernels"]
print(sys.argv[1], "cfg5", d["ms_per_step"], d["final_rms_px"], "chol", k["cholesky_solve"]["avg_us"], "reduce_finalize", k["schur_reduce_finalize"]["avg_us"])
PY
}
run graph A=1
run plain CBA_CHOL_GRAPH=0
run graph2 A=1
run plain2 CBA_CHOL_GRAPH=0
