#!/bin/bash
# round 4, call 1: the pruned library + the two-set pair kernel.  Full GPU suite; cfg4 A/B of the pair kernel (CBA_SCHUR_PP=0 / default) by bench
# line, phase clocks (profiling build) and kernel trace.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/r4c1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=400 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -5 $O/tests.log
for pp in 0 1; do
  CBA_SCHUR_PP=$pp timeout 300 python bench.py --no-cpu --also "" --steps 40 --warmup 8 > $O/bench_pp$pp.json 2> $O/bench_pp$pp.err
  python - <<PY
import json
d=json.loads(open("$O/bench_pp$pp.json").read().strip().splitlines()[-1])
k=d["roofline"]["kernels"]
print("pp=$pp ms_per_step", d["ms_per_step"], "pairs", k.get("schur_pairs"), "schur", k.get("schur"), "chol", k.get("cholesky_solve"), "rms", d["final_rms_px"], "nfev", d["solve"]["nfev"])
PY
done
P=$GRAFT_REPO_ROOT/caliscope_amd/libcaliscope_ba_prof.so
for pp in 0 1; do
  CALISCOPE_BA_LIB=$P CBA_SCHUR_PP=$pp CBA_SCHUR_CLOCK=1 timeout 200 python tools/newton_probe.py cfg4 1 2> $O/clock_pp$pp.log > /dev/null; grep "phases" $O/clock_pp$pp.log | tail -1
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_cfg4 -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --also "" --steps 20 --warmup 4 > $O/bench_trace.json 2> $O/trace.err
cd $GRAFT_REPO_ROOT
DB=$(find $O/trace_cfg4 -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB $O/cfg4_kernel_trace.md | head -14
find $O -name "*.db" -size +20M -delete; find $O -name "*kernel_trace.csv" -size +2M -delete; du -sh $O
