#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/c13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=400 -k "(evaluation_parity or step_parity or point_ordered or build_kernel_variants or camera_table or mixed_fisheye or ragged or limits) and not C420 and not C300 and not C200 and not C128" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -6 $O/tests.log
for cs in 1 0; do
CBA_BUILD_CS=$cs timeout 300 python bench.py --no-cpu --also cfg5 --steps 20 --warmup 5 > $O/bench_$cs.json 2> $O/bench_$cs.err
python - $cs <<'PY'
import json, sys
try:
    d=json.loads(open(f"gpurun_out/c13/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
    print("cs", sys.argv[1], d["ms_per_step"], "build", k["build"]["avg_us"], d["final_rms_px"], d["solve"]["nfev"], d["solve"]["cost"], d["setup_ms"])
    for n, v in d["also"].items(): print("   ", n, v.get("ms_per_step"), v.get("nfev"), v.get("final_rms_px"), (v.get("roofline") or {}).get("kernels", {}).get("build"), v.get("error"))
except Exception as e:
    print("FAILED", e); print(open(f"gpurun_out/c13/bench_{sys.argv[1]}.err").read()[-1500:])
PY
done
