#!/bin/bash
# the validation + profile run whose outputs are summarised under profiles/ (gpurun -- bash tools/gpu_profile_run.sh): full GPU suite, smoke, bench with
# the CPU baseline and the parity legs, rocprofv3 kernel traces (cfg4, cfg5, cfg2+3), HBM counters (cfg4, cfg5; FETCH_SIZE and WRITE_SIZE in separate
# passes, --kernel-trace only), phase clocks of the pair kernel, Cholesky trace, set-up timing
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/profile_run; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=400 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
timeout 120 python __graft_entry__.py --smoke > $O/smoke.log 2>&1
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
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
find $O -name "*kernel_trace.csv" -size +2M -delete
cd $GRAFT_REPO_ROOT; tail -4 $O/tests.log; cat $O/smoke.log | tail -2; cat $O/bench.time | tail -3; tail -c 400 $O/bench.json; du -sh $O
CBA_SCHUR_CLOCK=1 timeout 120 python tools/newton_probe.py cfg4 1 2> $O/schur_clock_cfg4.log; CBA_CHOL_TRACE=1 timeout 120 python tools/newton_probe.py cfg4 1 2> $O/chol_trace.log
timeout 200 python tools/create_timing.py > $O/create_timing.log 2>&1
timeout 200 python tools/real_session_timing.py > $O/real_session.log 2>&1
