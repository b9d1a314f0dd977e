#!/bin/bash
# GPU against scipy on the same arrays at BASELINE sizes -> profiles/parity_r03.json (copy it from gpurun_out/parity/)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/parity; mkdir -p $O
timeout 1500 python tools/parity_at_size.py $O/parity.json > $O/log.txt 2>&1; tail -5 $O/log.txt
