#!/bin/bash
# the whole GPU suite + smoke + real-session timing
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/suite; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -x -q --timeout=400 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
timeout 120 python __graft_entry__.py --smoke 2>&1 | tail -2
timeout 200 python tools/real_session_timing.py 2>&1 | cut -c1-150
