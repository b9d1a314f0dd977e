#!/bin/bash
# work queues + interleaved workgroup ids: parity, clocks, timing
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3d; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_constraints.py -m gpu -x -q --timeout=150 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -3 $O/tests.log
for q in 1 0; do
CBA_SCHUR_QUEUE=$q CBA_SCHUR_CLOCK=1 timeout 120 python tools/newton_probe.py cfg4 2 2> $O/clock_$q.log
grep -A12 -h 'k_schur_reg3' $O/clock_$q.log | tail -12 | cut -c1-210
  CBA_SCHUR_QUEUE=$q timeout 200 python bench.py --no-cpu --also cfg2,cfg3,cfg5 --steps 30 --warmup 6 > $O/bench_$q.json 2> $O/bench_$q.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3d/bench_$q.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print("queue $q", d["ms_per_step"], k["schur_pairs"]["avg_us"], k["schur"]["avg_us"], d["final_rms_px"], {n:(v["ms_per_step"], v["final_rms_px"], v["roofline"].get("kernels",{}).get("schur_pairs",{}).get("avg_us")) for n,v in d["also"].items()})
PY
done
