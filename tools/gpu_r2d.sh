#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=gpurun_out/r2d; mkdir -p $O
timeout 240 python tools/schur_split.py cfg4 0,16,1,2,4,8,10,15 > $O/split_cfg4.log 2>&1
timeout 400 python -m pytest tests/test_multi_device.py -m gpu -x -v --timeout=60 > $O/tests_md.log 2>&1; echo "rc=$?" >> $O/tests_md.log
tail -25 $O/tests_md.log; cat $O/split_cfg4.log
