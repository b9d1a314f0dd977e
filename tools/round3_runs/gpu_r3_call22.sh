#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 500 python tools/plan_threads_sweep.py 2>&1 | grep -v "^cba_create\|^  plan"
