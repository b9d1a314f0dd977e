#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/c10; mkdir -p $O
for g in 1 0; do
CBA_STEP_GRAPH=$g timeout 300 python bench.py --no-cpu --also cfg2,cfg3 --steps 40 --warmup 8 > $O/bench_$g.json 2> $O/bench_$g.err
python - $g <<'PY'
import json, sys
try:
    d=json.loads(open(f"gpurun_out/c10/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    print("graph", sys.argv[1], d["ms_per_step"], d["final_rms_px"], d["solve"]["nfev"], d["solve"]["cost"], {n: (v["ms_per_step"], v["nfev"], v["final_rms_px"]) for n, v in d["also"].items()})
except Exception as e:
    print("FAILED", e); print(open(f"gpurun_out/c10/bench_{sys.argv[1]}.err").read()[-1500:])
PY
done
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_multi_device.py -m gpu -x -q --timeout=400 -k "not C420 and not C300 and not C200 and not C128" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
