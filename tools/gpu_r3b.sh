#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3b; mkdir -p $O
for pr in 0 1; do
CBA_SCHUR_PRIO=$pr CBA_SCHUR_CLOCK=1 timeout 120 python tools/newton_probe.py cfg4 1 2> $O/clock_$pr.log
grep -A12 -h 'k_schur_reg3' $O/clock_$pr.log | tail -12 | cut -c1-200
  CBA_SCHUR_PRIO=$pr timeout 200 python bench.py --no-cpu --also cfg3,cfg5 --steps 30 --warmup 6 > $O/bench_$pr.json 2> $O/bench_$pr.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3b/bench_$pr.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print("prio $pr", d["ms_per_step"], k["schur_pairs"]["avg_us"], k["schur"]["avg_us"], {n:(v["ms_per_step"], v["roofline"].get("kernels",{}).get("schur_pairs",{}).get("avg_us")) for n,v in d["also"].items()})
PY
done
