#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=gpurun_out/r2h; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q --timeout=150 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
CBA_PLAN_TIMING=1 timeout 400 python bench.py --steps 20 --warmup 4 > $O/bench.json 2> $O/bench.err
timeout 400 python tools/parity_at_size.py $O/parity.json > $O/parity.log 2>&1
tail -6 $O/tests.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2h/bench.json').read().strip().splitlines()[-1])
print('ms/step',d['ms_per_step'],'value',d['value'],{n:v['avg_us'] for n,v in d['roofline']['kernels'].items()})
print('cpu',d.get('cpu_baseline')); print('parity',d.get('parity'))
for k,v in d.get('also',{}).items(): print(k, v.get('ms_per_step'), v.get('final_rms_px'), v.get('nfev'), v.get('accepted_steps'), v.get('rejected_trials'), v.get('error'))
PY
tail -30 $O/parity.log
