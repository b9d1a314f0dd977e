#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2c; mkdir -p $O
timeout 900 python -m pytest tests/test_multi_device.py tests/test_library_abi.py -m gpu -x -q > $O/tests_md.log 2>&1; echo "rc=$?" >> $O/tests_md.log
timeout 600 python tools/schur_split.py cfg4 > $O/split_cfg4.log 2>&1
rocprofv3 -L > $O/counters.txt 2>&1
tail -15 $O/tests_md.log; cat $O/split_cfg4.log
