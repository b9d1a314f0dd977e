#!/bin/bash
# round 4, call 17: SQ counters of the dominant kernels on cfg4 (two --pmc passes, --kernel-trace only): issue / wait / LDS shares of the pair kernel
cd /tmp; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r4c17; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*\|GRBM_[A-Z_]*" | sort -u > $O/counters.txt; wc -l $O/counters.txt
B=$GRAFT_REPO_ROOT/bench.py
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS -d $O/passA -o p --output-format csv -- python $B --no-cpu --also "" --steps 12 --warmup 2 > /dev/null 2> $O/passA.err; echo "passA rc=$?"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -d $O/passB -o p --output-format csv -- python $B --no-cpu --also "" --steps 12 --warmup 2 > /dev/null 2> $O/passB.err; echo "passB rc=$?"
cd $GRAFT_REPO_ROOT
python tools/sq_counter_summary.py $O/sq_cfg4.md $O/passA $O/passB 2>&1 | tail -30
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete

