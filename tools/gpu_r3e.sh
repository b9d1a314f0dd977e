#!/bin/bash
# fast / slow halves: parity, clocks, timing for a few weights
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3e; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=150 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log
CBA_SCHUR_CLOCK=1 timeout 120 python tools/newton_probe.py cfg4 2 2> $O/clock.log
grep -A12 -h 'k_schur_reg3' $O/clock.log | tail -12 | cut -c1-210
for w in 1.0 0.9 0.82 0.75; do
  CBA_SCHUR_SLOW_WEIGHT=$w timeout 200 python bench.py --no-cpu --also cfg3,cfg5 --steps 30 --warmup 6 > $O/bench_$w.json 2> $O/bench_$w.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3e/bench_$w.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print("slow weight $w", d["ms_per_step"], k["schur_pairs"]["avg_us"], k["schur"]["avg_us"], d["final_rms_px"], {n:(v["ms_per_step"], v["final_rms_px"], v["roofline"].get("kernels",{}).get("schur_pairs",{}).get("avg_us")) for n,v in d["also"].items()})
PY
done
