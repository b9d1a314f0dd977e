#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r2y; mkdir -p $O
for st in 0 16 32 48 64; do
  CBA_SCHUR_STAGGER=$st CBA_SCHUR_CLOCK=1 timeout 120 python tools/newton_probe.py cfg4 2 2> $O/clock_$st.log
  echo "stagger $st: $(grep -h 'k_schur_reg3' $O/clock_$st.log | tail -1)"
done
for st in 0 32; do
  CBA_SCHUR_STAGGER=$st timeout 200 python bench.py --no-cpu --also "" --steps 30 --warmup 6 > $O/bench_$st.json 2> $O/bench_$st.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2y/bench_$st.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print("stagger $st", d["ms_per_step"], k["schur_pairs"]["avg_us"], k["schur"]["avg_us"])
PY
done
