#!/bin/bash
# The validation + profile run whose outputs are summarised under profiles/ (gpurun -- bash tools/gpu_profile_run.sh [quick]; round 5: everything the
# round measures on its final library goes through THIS script — the per-call scripts of rounds 3-4 are gone): full GPU suite, smoke, the bench line as
# the driver runs it (CPU baseline + parity legs), rocprofv3 kernel traces (cfg4, cfg5, cfg2+3) and the dispatch list of one iteration, HBM counters
# (cfg4, cfg5; FETCH_SIZE and WRITE_SIZE in separate passes, --kernel-trace only), SQ counters (two passes each), phase clocks of the pair kernel and
# stamps of the dense solve (profiling build), set-up timing, the reference's 4-camera session, the sharded protocol on one device (2 and 8 ranks,
# cfg4 and cfg5), rank 0's shard alone for N = 1, 2, 4, 8 (tools/shard_projection.py), whose device memory it is (tools/device_memory_probe.py), a 40M-observation
# solve (tools/large_size_probe.py), the soak (tools/soak.py); tools/parity_at_size.py runs right behind the bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=30
O=$GRAFT_REPO_ROOT/gpurun_out/profile_run; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=400 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
timeout 120 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
( time timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; tail -3 $O/bench.time; tail -c 300 $O/bench.json; echo
# (parity_at_size right behind the bench line: the two carry the library digest that tests/test_bench_contract.py compares; a run cut short loses the tail, not these)
[ "$1" = quick ] || ( time timeout 1500 python tools/parity_at_size.py $O/parity.json > $O/parity.log 2>&1 ) 2> $O/parity.time
tail -3 $O/parity.time 2>/dev/null
cd /tmp
B=$GRAFT_REPO_ROOT/bench.py
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace_cfg4 -o t -- python $B --no-cpu --no-first-call --also "" --steps 20 --warmup 4 > $O/bench_cfg4.json 2> $O/trace_cfg4.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_cfg5 -o t -- python $B --no-cpu --no-first-call --workload cfg5 --also "" --steps 8 --warmup 2 > $O/bench_cfg5.json 2> $O/trace_cfg5.err
timeout 200 rocprofv3 --kernel-trace --stats -d $O/trace_cfg23 -o t -- python $B --no-cpu --workload cfg2 --also cfg3 --steps 40 --warmup 8 > $O/bench_cfg23.json 2> $O/trace_cfg23.err
for w in cfg4 cfg5; do
  st=12; [ $w = cfg5 ] && st=6
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch_$w -o p --output-format csv -- python $B --no-cpu --no-first-call --workload $w --also "" --steps $st --warmup 2 > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write_$w -o p --output-format csv -- python $B --no-cpu --no-first-call --workload $w --also "" --steps $st --warmup 2 > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS -d $O/sqA_$w -o p --output-format csv -- python $B --no-cpu --no-first-call --workload $w --also "" --steps $st --warmup 2 > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -d $O/sqB_$w -o p --output-format csv -- python $B --no-cpu --no-first-call --workload $w --also "" --steps $st --warmup 2 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
for w in cfg4 cfg5 cfg23; do DB=$(find $O/trace_$w -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB $O/${w}_kernel_trace.md > /dev/null; done
DB=$(find $O/trace_cfg23 -name "*.db" | head -1); [ -n "$DB" ] && python tools/trace_iteration.py $DB k_tprep 0.04 > $O/cfg2_iteration.txt 2>&1  # (cfg2 runs first: 48 of the ~640 iterations)
[ -n "$DB" ] && python tools/trace_iteration.py $DB k_tprep 0.6 > $O/cfg3_iteration.txt 2>&1
DB4=$(find $O/trace_cfg4 -name "*.db" | head -1); [ -n "$DB4" ] && python tools/trace_iteration.py $DB4 k_tprep 0.6 > $O/cfg4_iteration.txt 2>&1
for w in cfg4 cfg5; do
  python tools/pmc_summary.py $O/pmc_fetch_$w $O/pmc_write_$w $O/pmc_$w.md $O/pmc_$w.json > /dev/null 2>&1
  python tools/sq_counter_summary.py $O/sq_$w.md $O/sqA_$w $O/sqB_$w > /dev/null 2>&1
done
find $O -name "*.db" -size +8M -delete; find $O -name "*kernel_trace.csv" -size +2M -delete; find $O -name "*counter_collection.csv" -size +4M -delete
P=$GRAFT_REPO_ROOT/caliscope_amd/libcaliscope_ba_prof.so
# device-side stamps of one fused iteration (round 6: what sits between the kernels without a profiler attached; per-tile / XCD / dispatch-order lifetimes of the pair kernel)
for w in cfg2 cfg3 cfg4 cfg5; do
  CALISCOPE_BA_LIB=$P CBA_STAMPS=1 CBA_PLAN=full timeout 300 python bench.py --no-cpu --no-first-call --workload $w --also "" --steps 12 --warmup 4 > /dev/null 2> $O/stamps_$w.raw
  awk '/device stamps of the last fused iteration/{n++} n==1' $O/stamps_$w.raw > $O/stamps_$w.txt
done
CALISCOPE_BA_LIB=$P CBA_SCHUR_CLOCK=1 timeout 120 python tools/newton_probe.py cfg4 1 2> $O/schur_clock_cfg4.log > /dev/null
CALISCOPE_BA_LIB=$P CBA_CHOL_TRACE=1 timeout 120 python tools/newton_probe.py cfg4 1 2> $O/chol_trace.log > /dev/null
timeout 400 python tools/create_timing.py > $O/create_timing.log 2>&1
timeout 200 python tools/real_session_timing.py > $O/real_session.log 2>&1
timeout 200 python tools/real_session_timing.py --breakdown >> $O/real_session.log 2>&1
timeout 200 python tools/device_memory_probe.py > $O/device_memory_probe.log 2>&1
timeout 200 python tools/end_to_end.py cfg4 > $O/end_to_end_cfg4.log 2>&1
timeout 400 python tools/large_size_probe.py > $O/large_size_probe.log 2>&1
for w in 2 8; do
  devs=$(python -c "print(','.join(['0']*$w))")
  timeout 400 python bench.py --gpus $w --devices $devs --xchg direct --no-cpu --also "" --steps 20 --warmup 4 > $O/ranks${w}_cfg4.json 2> $O/ranks${w}_cfg4.err
  timeout 700 python bench.py --gpus $w --devices $devs --xchg direct --no-cpu --workload cfg5 --also "" --steps 8 --warmup 2 > $O/ranks${w}_cfg5.json 2> $O/ranks${w}_cfg5.err
done
timeout 300 python tools/shard_projection.py cfg4 > $O/shard_projection_cfg4.log 2>&1
timeout 600 python tools/shard_projection.py cfg5 > $O/shard_projection_cfg5.log 2>&1
timeout 400 python tools/soak.py 60 > $O/soak.log 2>&1
du -sh $O
