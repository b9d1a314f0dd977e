#!/bin/bash
# round 4, call 2: three forms of the six-parameter pair kernel on cfg4 — one set (CBA_SCHUR_PP=0), two sets with record prefetch (1), two sets of
# eight waves (2; spills) — by bench line and phase clocks; the 1000-camera case; the tests the last call did not reach.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/r4c2; mkdir -p $O
P=$GRAFT_REPO_ROOT/caliscope_amd/libcaliscope_ba_prof.so
for pp in 0 1 2; do
  CBA_SCHUR_PP=$pp timeout 300 python bench.py --no-cpu --also "" --steps 40 --warmup 8 > $O/bench_pp$pp.json 2> $O/bench_pp$pp.err
  python - <<PY
import json
d=json.loads(open("$O/bench_pp$pp.json").read().strip().splitlines()[-1])
k=d["roofline"]["kernels"]
print("pp=$pp ms_per_step", d["ms_per_step"], "pairs", k.get("schur_pairs",{}).get("avg_us"), "schur", k.get("schur",{}).get("avg_us"), "rms", d["final_rms_px"], "nfev", d["solve"]["nfev"])
PY
  CALISCOPE_BA_LIB=$P CBA_SCHUR_PP=$pp CBA_SCHUR_CLOCK=1 timeout 200 python tools/newton_probe.py cfg4 1 2> $O/clock_pp$pp.log > /dev/null; grep "phases" $O/clock_pp$pp.log | tail -1 | cut -c1-330
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=400 -k "two_set or C1000 or point_ordered or test_step_parity" > $O/tests_a.log 2>&1; tail -5 $O/tests_a.log
