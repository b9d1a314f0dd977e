#!/bin/bash
# round 4, call 3: interleaved issue (CBA_SCHUR_ILV=1) of the one-set pair kernel on cfg4 and cfg5 by bench line and phase clocks; the 1000-camera test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15 CBA_SCHUR_PP=0
O=$GRAFT_REPO_ROOT/gpurun_out/r4c3; mkdir -p $O
P=$GRAFT_REPO_ROOT/caliscope_amd/libcaliscope_ba_prof.so
for ilv in 0 1; do
  CBA_SCHUR_ILV=$ilv timeout 300 python bench.py --no-cpu --also "" --steps 40 --warmup 8 > $O/bench_ilv$ilv.json 2> $O/bench_ilv$ilv.err
  python - <<PY
import json
d=json.loads(open("$O/bench_ilv$ilv.json").read().strip().splitlines()[-1])
k=d["roofline"]["kernels"]
print("cfg4 ilv=$ilv ms_per_step", d["ms_per_step"], "pairs", k.get("schur_pairs",{}).get("avg_us"), "schur", k.get("schur",{}).get("avg_us"), "rms", d["final_rms_px"], "nfev", d["solve"]["nfev"])
PY
  CALISCOPE_BA_LIB=$P CBA_SCHUR_ILV=$ilv CBA_SCHUR_CLOCK=1 timeout 200 python tools/newton_probe.py cfg4 1 2> $O/clock_ilv$ilv.log > /dev/null; grep "phases" $O/clock_ilv$ilv.log | tail -1 | cut -c1-330
done
for ilv in 0 1; do
  CBA_SCHUR_ILV=$ilv timeout 400 python bench.py --no-cpu --workload cfg5 --also "" --steps 10 --warmup 2 > $O/bench5_ilv$ilv.json 2> $O/bench5_ilv$ilv.err
  python - <<PY
import json
d=json.loads(open("$O/bench5_ilv$ilv.json").read().strip().splitlines()[-1])
k=d["roofline"]["kernels"]
print("cfg5 ilv=$ilv ms_per_step", d["ms_per_step"], "pairs", k.get("schur_pairs",{}).get("avg_us"), "schur", k.get("schur",{}).get("avg_us"), "rms", d["final_rms_px"], "nfev", d["solve"]["nfev"])
PY
done
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=300 -k "thousand" > $O/tests_a.log 2>&1; tail -4 $O/tests_a.log
