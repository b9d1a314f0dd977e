#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/c26; mkdir -p $O
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu --also cfg2,cfg3 --steps 40 --warmup 8 > $O/bench_$label.json 2> $O/bench_$label.err
  python - "$label" <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/c26/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
k=d["roofline"]["kernels"]
print(sys.argv[1], "cfg4", d["ms_per_step"], d["final_rms_px"], "reduce_finalize", k["schur_reduce_finalize"]["avg_us"], "vector", k["vector_ops"]["avg_us"], "chol", k["cholesky_solve"]["avg_us"],
      "| " + " ".join(f"{n} {v['ms_per_step']} {v.get('final_rms_px')}" for n, v in d["also"].items()))
PY
}
run new A=1
run oldfold CBA_FOLD=kernel CBA_STEP_SMALL=0
run plainchol CBA_CHOL_GRAPH=0
timeout 100 python __graft_entry__.py --smoke 2>&1 | tail -1
