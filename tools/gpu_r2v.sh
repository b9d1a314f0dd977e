#!/bin/bash
# 128-byte records: parity + phase clocks + timing
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r2v; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_constraints.py -m gpu -x -q --timeout=150 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
CBA_SCHUR_CLOCK=1 timeout 120 python tools/newton_probe.py cfg4 2 2> $O/clock_wide.log
CBA_SCHUR_CLOCK=1 CBA_SCHUR_WIDE=0 timeout 120 python tools/newton_probe.py cfg4 2 2> $O/clock_narrow.log
timeout 200 python bench.py --no-cpu --also cfg3,cfg5 --steps 30 --warmup 6 > $O/bench_default.json 2> $O/bench_default.err
CBA_SCHUR_WIDE=0 timeout 200 python bench.py --no-cpu --also "" --steps 30 --warmup 6 > $O/bench_narrow.json 2> $O/bench_narrow.err
tail -3 $O/tests.log
grep -h "k_schur_reg3" $O/clock_wide.log | tail -1; grep -h "k_schur_reg3" $O/clock_narrow.log | tail -1
python - <<'PY'
import json
def show(n):
    try:
        d=json.loads(open(f"gpurun_out/r2v/bench_{n}.json").read().strip().splitlines()[-1])
        k=d["roofline"]["kernels"]
        print(n, d["config"]["workload"][:12], d["ms_per_step"], {x:k[x]["avg_us"] for x in ("schur","schur_pairs","schur_reduce_finalize","cholesky_solve","build")}, d["final_rms_px"], d.get("engine",{}).get("schur_stream_len"), d.get("engine",{}).get("schur_wide"))
        for a,v in d.get("also",{}).items(): print("   ",a,v["ms_per_step"],v["final_rms_px"],v["roofline"]["avg_launch_us"], v["roofline"].get("kernels",{}).get("schur_pairs",{}).get("avg_us"))
    except Exception as e: print(n,"failed",e)
for n in ("default","narrow"): show(n)
PY
