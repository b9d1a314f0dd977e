#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=gpurun_out/r2l; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q --timeout=150 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
timeout 300 python bench.py --no-cpu --also cfg2 --steps 20 --warmup 4 > $O/bench.json 2> $O/bench.err
tail -5 $O/tests.log
python - <<PY
import json
d=json.loads(open('gpurun_out/r2l/bench.json').read().strip().splitlines()[-1])
print('ms/step',d['ms_per_step'],{n:v['avg_us'] for n,v in d['roofline']['kernels'].items()})
for k,v in d.get('also',{}).items(): print('  also',k,v.get('ms_per_step'), (v.get('roofline') or {}).get('avg_launch_us'), v.get('final_rms_px'))
PY
