#!/bin/bash
# round 4, call 18: the same SQ counters on cfg5 (nine-parameter pair kernel) (two --pmc passes, --kernel-trace only): issue / wait / LDS shares of the pair kernel
cd /tmp; export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r4c18; mkdir -p $O
B=$GRAFT_REPO_ROOT/bench.py
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS -d $O/passA -o p --output-format csv -- python $B --no-cpu --workload cfg5 --also "" --steps 6 --warmup 2 > /dev/null 2> $O/passA.err; echo "passA rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -d $O/passB -o p --output-format csv -- python $B --no-cpu --workload cfg5 --also "" --steps 6 --warmup 2 > /dev/null 2> $O/passB.err; echo "passB rc=$?"
cd $GRAFT_REPO_ROOT
python tools/sq_counter_summary.py $O/sq_cfg5.md $O/passA $O/passB 2>&1 | tail -30
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete

