#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=gpurun_out/r2p; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_constraints.py -m gpu -x -q --timeout=150 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
for m in 2 3; do
  CBA_GRID_MULT=$m timeout 300 python bench.py --no-cpu --also cfg5 --steps 12 --warmup 3 > $O/bench_m$m.json 2> $O/bench_m$m.err
done
tail -4 $O/tests.log
for m in m2 m3; do python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r2p/bench_$m.json').read().strip().splitlines()[-1])
    k=d['roofline']['kernels']
    print('$m','ms/step',d['ms_per_step'],{n:k[n]['avg_us'] for n in ('build','jv','schur','schur_pairs','backsub','cholesky_solve')})
    c=(d.get('also') or {}).get('cfg5',{}); kk=(c.get('roofline') or {}).get('kernels',{})
    print('   cfg5', c.get('ms_per_step'), {n:kk[n]['avg_us'] for n in ('build','jv','schur','schur_pairs','backsub','cost') if n in kk})
except Exception as e: print('$m ERR', e)
PY
done
