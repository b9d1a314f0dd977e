#!/bin/bash
# the whole GPU suite + the real-session timing
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/suite; mkdir -p $O
timeout 700 python -m pytest tests -m gpu -x -q --timeout=150 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -4 $O/tests.log
timeout 200 python tools/real_session_timing.py 2>&1 | cut -c1-150
CALISCOPE_HIP_ENGINE_CACHE=0 timeout 200 python tools/real_session_timing.py 2>&1 | cut -c1-110
