#!/bin/bash
# round 2 profiles: kernel traces (cfg4, cfg2+cfg3, cfg5) and PMC traffic (cfg4, cfg5), Cholesky phase trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r2n; mkdir -p $O
CBA_CHOL_TRACE=1 timeout 120 python tools/chol_trace.py > $O/chol_trace.log 2>&1

cd /tmp
B=$GRAFT_REPO_ROOT/bench.py
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace_cfg4 -o t -- python $B --no-cpu --also "" --steps 20 --warmup 4 > $O/bench_cfg4.json 2> $O/trace_cfg4.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_cfg5 -o t -- python $B --no-cpu --workload cfg5 --also "" --steps 8 --warmup 2 > $O/bench_cfg5.json 2> $O/trace_cfg5.err
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace_cfg23 -o t -- python $B --no-cpu --workload cfg2 --also cfg3 --steps 20 --warmup 4 > $O/bench_cfg23.json 2> $O/trace_cfg23.err
for w in cfg4 cfg5; do
  st=12; [ $w = cfg5 ] && st=6
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_$w -o p --output-format csv -- python $B --no-cpu --workload $w --also "" --steps $st --warmup 2 > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_$w -o p --output-format csv -- python $B --no-cpu --workload $w --also "" --steps $st --warmup 2 > /dev/null 2>&1
done
find $O -name "*kernel_trace.csv" -delete
cd $GRAFT_REPO_ROOT; grep "step  5\|step -1" $O/chol_trace.log | tail -2; du -sh $O
