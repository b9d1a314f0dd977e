#!/bin/bash
# round 4, call 13: four-wave block factorisation with optimistic reads in the main wave: phase stamps and a step-parity subset
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r4c13; mkdir -p $O
P=$GRAFT_REPO_ROOT/caliscope_amd/libcaliscope_ba_prof.so
CALISCOPE_BA_LIB=$P CBA_CHOL_TRACE=1 timeout 120 python tools/newton_probe.py cfg4 1 2> $O/chol_trace.log > /dev/null; head -6 $O/chol_trace.log
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "step_parity" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; grep -E "passed|failed|rc=" $O/tests.log | tail -3
timeout 200 python bench.py --no-cpu --steps 40 --warmup 8 --also "" 2> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg4', d['ms_per_step'], d['roofline']['kernels']['cholesky_solve']['avg_us'], d['final_rms_px'])"
