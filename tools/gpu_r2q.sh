#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=gpurun_out/r2q; mkdir -p $O
CBA_CHOL_TRACE=1 timeout 120 python tools/chol_trace.py > $O/chol_trace.log 2>&1
CBA_SCHUR=reg2 timeout 100 python tools/schur_split.py cfg4 0 > $O/split.log 2>&1
timeout 200 python bench.py --no-cpu --also "" --steps 12 --warmup 3 > $O/bench.json 2> $O/bench.err
grep "step  5\|step -1\|step 11" $O/chol_trace.log | tail -3; cat $O/split.log
python - <<PY
import json
d=json.loads(open('gpurun_out/r2q/bench.json').read().strip().splitlines()[-1])
k=d['roofline']['kernels']
print('ms/step',d['ms_per_step'],{n:k[n]['avg_us'] for n in k})
PY
rocm-smi --showclocks 2>/dev/null | head -20
