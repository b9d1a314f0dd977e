#!/bin/bash
# round 2, first GPU call: parity tests on the dealt plan + k_schur_reg2, then A/B of the Schur variants
mkdir -p gpurun_out/r2a && cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r2a
timeout 600 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
for mode in reg1 v2_128 v2_64; do
  case $mode in
    reg1) export CBA_SCHUR=reg1; unset CBA_PLAN_REGION;;
    v2_128) unset CBA_SCHUR; export CBA_PLAN_REGION=128;;
    v2_64) unset CBA_SCHUR; export CBA_PLAN_REGION=64;;
  esac
  CBA_PLAN_TIMING=1 timeout 300 python bench.py --no-cpu --also cfg3,cfg5 --steps 20 --warmup 4 > $O/bench_$mode.json 2> $O/bench_$mode.err
done
unset CBA_SCHUR CBA_PLAN_REGION
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o cfg4 -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --also "" --steps 20 --warmup 4 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R $O/prof | head -30
tail -3 $O/tests.log
