#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=gpurun_out/r2k; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q --timeout=150 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
CBA_PLAN_TIMING=1 timeout 300 python bench.py --no-cpu --also cfg3,cfg5 --steps 20 --warmup 4 > $O/bench_reg3.json 2> $O/bench_reg3.err
CBA_SCHUR=reg2 timeout 300 python bench.py --no-cpu --also cfg5 --steps 20 --warmup 4 > $O/bench_reg2.json 2> $O/bench_reg2.err
tail -5 $O/tests.log
for m in reg3 reg2; do python - <<PY
import json
d=json.loads(open('gpurun_out/r2k/bench_$m.json').read().strip().splitlines()[-1])
print('$m ms/step',d['ms_per_step'],{n:v['avg_us'] for n,v in d['roofline']['kernels'].items() if n in ('schur','schur_pairs','cholesky_solve','build')})
for k,v in d.get('also',{}).items(): print('  also',k,v.get('ms_per_step'), (v.get('roofline') or {}).get('avg_launch_us'), v.get('final_rms_px'))
PY
done; grep "plan:" $O/bench_reg3.err | grep tiles
