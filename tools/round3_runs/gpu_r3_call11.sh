#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/c11; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q --timeout=400 -k "not C420 and not C300 and not C200 and not C128" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
for sm in 1 0; do
CBA_CHOL_SMALL=$sm timeout 300 python bench.py --no-cpu --workload cfg2 --also "" --steps 40 --warmup 8 > $O/bench_$sm.json 2> $O/bench_$sm.err
python - $sm <<'PY'
import json, sys
d=json.loads(open(f"gpurun_out/c11/bench_{sys.argv[1]}.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
print("chol_small", sys.argv[1], d["ms_per_step"], d["final_rms_px"], d["solve"]["nfev"], {x:k[x]["avg_us"] for x in k})
PY
CBA_CHOL_SMALL=$sm timeout 100 python tools/real_session_timing.py 2>&1 | tail -3
done
