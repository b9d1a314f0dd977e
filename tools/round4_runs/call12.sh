#!/bin/bash
# round 4, call 12: the four-wave block factorisation (chol_factor_block_mw): step parity, phase stamps of the dense solve, bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r4c12; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "step_parity or thousand or small_solve or two_stage or converged" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log | tail -3
P=$GRAFT_REPO_ROOT/caliscope_amd/libcaliscope_ba_prof.so
CALISCOPE_BA_LIB=$P CBA_CHOL_TRACE=1 timeout 120 python tools/newton_probe.py cfg4 1 2> $O/chol_trace.log > /dev/null; head -8 $O/chol_trace.log
timeout 300 python bench.py --no-cpu --steps 40 --warmup 8 > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json, os
d = json.loads(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4c12/bench.json").read().strip().splitlines()[-1])
print("cfg4", d["ms_per_step"], {k: v["avg_us"] for k, v in d["roofline"]["kernels"].items() if k in ("cholesky_solve", "schur_pairs", "schur")})
for k, v in d.get("also", {}).items(): print(k, v["ms_per_step"], v.get("final_rms_px"), v.get("nfev"))
PY
