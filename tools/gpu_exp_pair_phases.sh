#!/bin/bash
# what binds the pair kernel: the product library against builds without the pair arithmetic, without the pairs' LDS reads, without the gather
# (-DCBA_EXP_PAIR_NOMATH / _PAIR_NOREAD / _NOGATHER, built by hand into tools/exp/)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
W=${1:-cfg4}
timeout 60 python tools/pair_phase_probe.py $W 10 2>&1 | tail -1
for v in PAIR_NOMATH PAIR_NOREAD NOGATHER; do CALISCOPE_BA_LIB=$GRAFT_REPO_ROOT/tools/exp/libcba_$v.so timeout 60 python tools/pair_phase_probe.py $W 10 2>&1 | tail -1; done
