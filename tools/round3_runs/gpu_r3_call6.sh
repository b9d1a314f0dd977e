#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/c6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_constraints.py tests/test_multi_device.py -m gpu -x -q --timeout=400 -k "not C420 and not C300 and not C200 and not C128" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
for spec in 1 0; do
CBA_SPECULATE=$spec timeout 200 python bench.py --no-cpu --also cfg2,cfg3 --steps 30 --warmup 6 > $O/bench_$spec.json 2> $O/bench_$spec.err
python - $spec <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/c6/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print("spec", sys.argv[1], d["ms_per_step"], d["final_rms_px"], d["solve"]["nfev"], d["solve"]["cost"], {n: (v["ms_per_step"], v["nfev"], v["final_rms_px"]) for n, v in d["also"].items()})
PY
done
