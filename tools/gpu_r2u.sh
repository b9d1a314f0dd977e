#!/bin/bash
# phase clocks of the pair kernel, wide and narrow
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r2u; mkdir -p $O
CBA_SCHUR_CLOCK=1 timeout 120 python tools/newton_probe.py cfg4 3 2> $O/clock_wide.log
CBA_SCHUR_CLOCK=1 CBA_SCHUR_WIDE=0 timeout 120 python tools/newton_probe.py cfg4 3 2> $O/clock_narrow.log
CBA_SCHUR_CLOCK=1 CBA_SCHUR_WIDE=0 CBA_GRID_MULT=1 timeout 120 python tools/newton_probe.py cfg4 3 2> $O/clock_narrow_g1.log
grep -h "k_schur_reg3" $O/clock_wide.log | tail -2; grep -h "k_schur_reg3" $O/clock_narrow.log | tail -2; grep -h "k_schur_reg3" $O/clock_narrow_g1.log | tail -1
rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -2
