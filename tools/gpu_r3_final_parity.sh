#!/bin/bash
# last GPU run of the round: the bench line as the driver will run it (-> profiles/r03_end_bench.json) and GPU-against-scipy at BASELINE sizes
# (tools/parity_at_size.py -> profiles/parity_r03.json: cfg2, cfg3 product settings, cfg3 to scipy's own convergence, the 1M-observation cfg5 sample)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/final; mkdir -p $O
( time timeout 900 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time
tail -3 $O/bench.time; tail -c 300 $O/bench.json
timeout 2400 python tools/parity_at_size.py $O/parity.json > $O/parity.log 2>&1; tail -3 $O/parity.log
