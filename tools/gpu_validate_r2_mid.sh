#!/bin/bash
# round 2 final validation + profiles
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/r2r; mkdir -p $O
timeout 700 python -m pytest tests -m gpu -x -q --timeout=150 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
timeout 60 python __graft_entry__.py --smoke > $O/smoke.log 2>&1
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
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
timeout 200 rocprofv3 --kernel-trace --marker-trace -d $O/markers -o m --output-format csv -- python $GRAFT_REPO_ROOT/tools/newton_probe.py cfg2 2 > /dev/null 2>&1
find $O -name "*kernel_trace.csv" -size +2M -delete
cd $GRAFT_REPO_ROOT; tail -4 $O/tests.log; cat $O/smoke.log | tail -2; tail -c 600 $O/bench.json; du -sh $O
