#!/bin/bash
# round 2, second GPU call: region size of the dealt plan vs L2 reuse of the T-record gather
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2b; mkdir -p $O
for r in 2 4 8 16 32; do
  CBA_PLAN_REGION=$r CBA_PLAN_TIMING=1 timeout 200 python bench.py --no-cpu --also "" --steps 20 --warmup 4 > $O/bench_r$r.json 2> $O/bench_r$r.err
done
CBA_PLAN_REGION=8 timeout 300 python bench.py --no-cpu --also cfg5 --steps 10 --warmup 2 > $O/bench_cfg5_r8.json 2> $O/bench_cfg5_r8.err
CBA_PLAN_REGION=24 timeout 300 python bench.py --no-cpu --also cfg5 --steps 10 --warmup 2 > $O/bench_cfg5_r24.json 2> $O/bench_cfg5_r24.err
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu --also '' --steps 6 --warmup 2"
cd /tmp
for r in 8 128; do
  export CBA_PLAN_REGION=$r
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/$O/pmc_fetch_r$r -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --also "" --steps 6 --warmup 2 > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $GRAFT_REPO_ROOT/$O/pmc_tcc_r$r -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --also "" --steps 6 --warmup 2 > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $GRAFT_REPO_ROOT/$O/pmc_sq_r$r -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --also "" --steps 6 --warmup 2 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
# keep only the csv summaries small: drop per-dispatch kernel traces
find $O -name "*kernel_trace.csv" -delete
du -sh $O; ls $O
