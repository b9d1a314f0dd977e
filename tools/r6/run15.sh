#!/bin/bash
# round 6, GPU call 15: where a first optimize() with the board spends its time (host stages, phases of cba_set_constraints); the full-size cfg5 property check
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=30
O=$GRAFT_REPO_ROOT/gpurun_out/r6_run15; mkdir -p $O
CBA_PLAN_TIMING=1 timeout 300 python tools/real_session_timing.py --breakdown > $O/breakdown.txt 2>&1; grep -v "^  plan\|^  create" $O/breakdown.txt | tail -30
timeout 300 python tools/real_session_timing.py > $O/real_session.txt 2>&1; cat $O/real_session.txt
( time timeout 900 python -c "
import sys, json; sys.path.insert(0, 'tools')
import parity_at_size as pas
print(json.dumps(pas.full_size_linear_algebra('cfg5'), indent=1))" ) > $O/full.txt 2>&1; tail -22 $O/full.txt
