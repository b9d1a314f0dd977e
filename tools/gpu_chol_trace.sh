#!/bin/bash
# phase stamps of the dense solve (CBA_CHOL_TRACE=1)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/choltrace; mkdir -p $O
for w in cfg4 cfg5; do CBA_CHOL_TRACE=1 timeout 120 python tools/newton_probe.py $w 2 2> $O/$w.log; grep "chol backward" $O/$w.log | tail -1; done
