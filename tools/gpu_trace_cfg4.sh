#!/bin/bash
# kernel trace of a few cfg4 steps (rocprofv3), summarised with tools/rocpd_stats.py
cd /tmp; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/trace4; mkdir -p $O
timeout 200 rocprofv3 --kernel-trace --stats -d $O/t -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --also "" --steps 12 --warmup 3 > $O/bench.json 2> $O/err.log
cd $GRAFT_REPO_ROOT; python tools/rocpd_stats.py $O/t/t_results.db $O/trace.md | cut -c1-125 | head -14
