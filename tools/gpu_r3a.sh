#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3a; mkdir -p $O
CBA_SCHUR_CLOCK=1 timeout 120 python tools/newton_probe.py cfg4 1 2> $O/clock.log
grep -A12 -h 'k_schur_reg3' $O/clock.log | tail -12
for a in -1 1.5 2.6 4; do
  CBA_PLAN_COST_A=$a timeout 200 python bench.py --no-cpu --also cfg3,cfg5 --steps 30 --warmup 6 > $O/bench_$a.json 2> $O/bench_$a.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3a/bench_$a.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print("cost_a $a", d["ms_per_step"], k["schur_pairs"]["avg_us"], k["schur"]["avg_us"], {n:(v["ms_per_step"], v["roofline"].get("kernels",{}).get("schur_pairs",{}).get("avg_us")) for n,v in d["also"].items()})
PY
done
