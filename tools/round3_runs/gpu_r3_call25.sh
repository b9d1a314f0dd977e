#!/bin/bash
# evidence runs on the final library: kernel trace of the launch-bound workload (cfg2) with the dispatch sequence of one iteration and the idle share;
# two ranks of the sharded protocol on ONE device (peer-to-peer device group: RCCL refuses two ranks on a device)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/c25; mkdir -p $O
B=$GRAFT_REPO_ROOT/bench.py
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace_cfg2 -o t -- python $B --no-cpu --workload cfg2 --also "" --steps 40 --warmup 8 > $O/bench_cfg2.json 2> $O/trace_cfg2.err )
DB=$(find $O/trace_cfg2 -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/cfg2_kernel_trace.md > /dev/null
python tools/trace_gaps.py $DB k_schur_reg3 > $O/cfg2_gaps.txt 2>&1; cat $O/cfg2_gaps.txt
python tools/trace_iteration.py $DB k_tprep > $O/cfg2_iteration.txt 2>&1; tail -3 $O/cfg2_iteration.txt
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -size +20M -delete
timeout 200 python bench.py --gpus 2 --devices 0,0 --xchg direct --no-cpu --also "" --steps 20 --warmup 4 > $O/bench_two_ranks.json 2> $O/bench_two_ranks.err; tail -c 600 $O/bench_two_ranks.json
