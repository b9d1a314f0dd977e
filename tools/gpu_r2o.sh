#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/r2o; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=150 -k "deterministic or evaluation" > $O/tests_det.log 2>&1; echo "rc=$?" >> $O/tests_det.log
for r in 8 16 32 128; do
  CBA_PLAN_REGION=$r CBA_PLAN_TIMING=1 timeout 300 python bench.py --no-cpu --also cfg5 --steps 12 --warmup 3 > $O/bench_r$r.json 2> $O/bench_r$r.err
done
CBA_DETERMINISTIC=1 timeout 200 python bench.py --no-cpu --also "" --steps 12 --warmup 3 > $O/bench_det.json 2> $O/bench_det.err
cd /tmp
for r in 16 32; do
  CBA_PLAN_REGION=$r timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_r$r -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --also "" --steps 8 --warmup 2 > /dev/null 2>&1
done
find $O -name "*kernel_trace.csv" -delete
cd $GRAFT_REPO_ROOT
tail -4 $O/tests_det.log
for r in 8 16 32 128 det; do python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r2o/bench_$r.json').read().strip().splitlines()[-1])
    k=d['roofline']['kernels']
    print('$r','ms/step',d['ms_per_step'],'schur',k['schur']['avg_us'],'pairs',k['schur_pairs']['avg_us'],'build',k['build']['avg_us'], 'cfg5', (d.get('also') or {}).get('cfg5',{}).get('ms_per_step'), ((d.get('also') or {}).get('cfg5',{}).get('roofline') or {}).get('avg_launch_us'))
except Exception as e: print('$r ERR', e)
PY
done
