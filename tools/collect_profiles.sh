#!/bin/bash
# Copies the summaries of the last tools/gpu_profile_run.sh run (gpurun_out/profile_run, scratch) into profiles/ under this round's names.
R=${1:-r06}; S=gpurun_out/profile_run; D=profiles
cd "$(dirname "$0")/.." || exit 1
tail -1 $S/bench.json > $D/${R}_end_bench.json
[ -f $S/parity.json ] && cp $S/parity.json $D/parity_${R}.json
for w in cfg4 cfg5 cfg23; do cp $S/${w}_kernel_trace.md $D/${R}_${w}_kernel_trace.md; done
for w in cfg2 cfg3 cfg4; do [ -f $S/${w}_iteration.txt ] && cp $S/${w}_iteration.txt $D/${R}_${w}_iteration.txt; done
for w in cfg4 cfg5; do cp $S/pmc_$w.md $D/${R}_${w}_pmc.md; cp $S/pmc_$w.json $D/pmc_$w.json; [ -f $S/sq_$w.md ] && cp $S/sq_$w.md $D/${R}_${w}_sq_counters.md; done
for w in cfg2 cfg3 cfg4 cfg5; do [ -s $S/stamps_$w.txt ] && cp $S/stamps_$w.txt $D/${R}_${w}_device_stamps.txt; done
cp $S/schur_clock_cfg4.log $D/${R}_schur_phase_clocks.txt
cp $S/chol_trace.log $D/${R}_chol_trace.txt
cp $S/create_timing.log $D/${R}_setup_timing.txt
cp $S/real_session.log $D/${R}_real_session_timing.txt
cp $S/soak.log $D/${R}_soak.txt
for f in device_memory_probe end_to_end_cfg4 large_size_probe; do [ -s $S/$f.log ] && cp $S/$f.log $D/${R}_$f.txt; done
for w in 2 8; do n=two; [ $w = 8 ] && n=eight; for c in cfg4 cfg5; do [ -s $S/ranks${w}_$c.json ] && tail -1 $S/ranks${w}_$c.json > $D/${R}_${n}_ranks_one_device_$c.json; done; done
for c in cfg4 cfg5; do [ -s $S/shard_projection_$c.log ] && cp $S/shard_projection_$c.log $D/${R}_shard_projection_$c.txt; done
{ echo "GPU suite:"; tail -3 $S/tests.log; echo "smoke:"; tail -1 $S/smoke.log; echo "bench.py wall:"; cat $S/bench.time; echo "parity_at_size wall:"; cat $S/parity.time 2>/dev/null; } > $D/${R}_final_run.txt
ls -la $D | grep ${R}
