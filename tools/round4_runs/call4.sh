#!/bin/bash
# round 4, call 4: the loader / compute pair kernel (k_schur_lc, CBA_SCHUR_LC=1 / 0) on cfg4 by bench line and phase clocks; its parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/r4c4; mkdir -p $O
P=$GRAFT_REPO_ROOT/caliscope_amd/libcaliscope_ba_prof.so
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=200 -k "loader_compute" > $O/tests_a.log 2>&1; tail -3 $O/tests_a.log
for lc in 0 1; do
  CBA_SCHUR_LC=$lc timeout 300 python bench.py --no-cpu --also "" --steps 40 --warmup 8 > $O/bench_lc$lc.json 2> $O/bench_lc$lc.err
  python - <<PY
import json
d=json.loads(open("$O/bench_lc$lc.json").read().strip().splitlines()[-1])
k=d["roofline"]["kernels"]
print("cfg4 lc=$lc ms_per_step", d["ms_per_step"], "pairs", k.get("schur_pairs",{}).get("avg_us"), "schur", k.get("schur",{}).get("avg_us"), "rms", d["final_rms_px"], "nfev", d["solve"]["nfev"])
PY
  CALISCOPE_BA_LIB=$P CBA_SCHUR_LC=$lc CBA_SCHUR_CLOCK=1 timeout 200 python tools/newton_probe.py cfg4 1 2> $O/clock_lc$lc.log > /dev/null; grep "phases" $O/clock_lc$lc.log | tail -2 | cut -c1-330
done
