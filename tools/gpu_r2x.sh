#!/bin/bash
# software-pipelined pair loop (narrow NC = 6): phase clocks + timing + parity of the pair kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r2x; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=150 -k "wide or step_parity or lds_tile or evaluation or converged" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
CBA_SCHUR_CLOCK=1 timeout 120 python tools/newton_probe.py cfg4 2 2> $O/clock_narrow.log
timeout 200 python bench.py --no-cpu --also cfg2,cfg3,cfg5 --steps 30 --warmup 6 > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/tests.log
grep -h "k_schur_reg3" $O/clock_narrow.log | tail -1
python - <<'PY'
import json
def show(n):
    try:
        d=json.loads(open(f"gpurun_out/r2x/bench_{n}.json").read().strip().splitlines()[-1])
        k=d["roofline"]["kernels"]
        print(n, d["config"]["workload"][:12], d["ms_per_step"], {x:k[x]["avg_us"] for x in ("schur","schur_pairs","schur_reduce_finalize","cholesky_solve","build")}, d["final_rms_px"], d.get("engine",{}).get("schur_stream_len"), d.get("engine",{}).get("schur_wide"))
        for a,v in d.get("also",{}).items(): print("   ",a,v["ms_per_step"],v["final_rms_px"],v["roofline"]["avg_launch_us"], v["roofline"].get("kernels",{}).get("schur_pairs",{}).get("avg_us"))
    except Exception as e: print(n,"failed",e)
show("default")
PY
