#!/bin/bash
# wide (32 x 32) Schur tiles: parity suite + timing against the narrow kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/r2s; mkdir -p $O
timeout 700 python -m pytest tests -m gpu -x -q --timeout=150 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
timeout 200 python bench.py --no-cpu --also cfg3 --steps 30 --warmup 6 > $O/bench_wide.json 2> $O/bench_wide.err
CBA_SCHUR_WIDE=0 timeout 200 python bench.py --no-cpu --also cfg3 --steps 30 --warmup 6 > $O/bench_narrow.json 2> $O/bench_narrow.err
CBA_PLAN_REGION=32 timeout 200 python bench.py --no-cpu --also "" --steps 30 --warmup 6 > $O/bench_wide_r32.json 2> $O/bench_wide_r32.err
tail -4 $O/tests.log
python - <<'PY'
import json
for n in ("wide","narrow","wide_r32"):
    try:
        d=json.loads(open(f"gpurun_out/r2s/bench_{n}.json").read().strip().splitlines()[-1])
        k=d["roofline"]["kernels"]
        print(n, d["ms_per_step"], {x:k[x]["avg_us"] for x in ("schur","schur_pairs","schur_reduce_finalize","cholesky_solve","build")}, d["final_rms_px"], d.get("engine",{}).get("schur_stream_len"))
        for a,v in d.get("also",{}).items(): print("   ",a,v["ms_per_step"],v["final_rms_px"],v["roofline"]["avg_launch_us"])
    except Exception as e: print(n,"failed",e)
PY
