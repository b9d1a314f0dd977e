#!/bin/bash
# round 6, GPU call 6: small-component constraint kernels (dense blocks in LDS)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=30
O=$GRAFT_REPO_ROOT/gpurun_out/r6_run6; mkdir -p $O
timeout 900 python -m pytest tests/test_constraints.py tests/test_fuzz_structures.py tests/test_multi_device.py tests/test_stage_driver.py tests/test_scenarios.py tests/test_reference_host_fixtures.py -m gpu -x -q --timeout=400 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -15 $O/tests.log
timeout 200 python tools/real_session_timing.py > $O/real_session.log 2>&1; cat $O/real_session.log
CBA_CON_SMALL=0 timeout 200 python tools/real_session_timing.py > $O/real_session_old.log 2>&1; cat $O/real_session_old.log
CBA_SOLVE_TRACE=1 timeout 200 python tools/real_session_timing.py 2>&1 | grep -A12 "cba_solve trace" | tail -45
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace_session -o t -- python $GRAFT_REPO_ROOT/tools/real_session_timing.py > /dev/null 2> $O/trace_session.err
cd $GRAFT_REPO_ROOT
DB=$(find $O/trace_session -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB $O/real_session_kernel_trace.md > /dev/null; head -20 $O/real_session_kernel_trace.md | cut -c1-160
find $O -name "*.db" -size +8M -delete
