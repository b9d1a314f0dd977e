#!/bin/bash
# round 6, GPU call 14: nine-parameter pair kernel split by sub-block (234 instead of 324 FP64 instructions per pair) against the previous library,
# same box; the board session after the host-side change of the constraint rows
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=30
O=$GRAFT_REPO_ROOT/gpurun_out/r6_run14; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q --timeout=400 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -5 $O/tests.log
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]; print(d["config"]["workload"][:5], "ms_per_step", d["ms_per_step"], "rms", d.get("final_rms_px"), {n: round(v["avg_us"],1) for n,v in k.items()})'
OLD=$GRAFT_REPO_ROOT/caliscope_amd/libcaliscope_ba_old.so
for rep in 1 2; do
for lib in new old; do
  for w in cfg5 cfg4; do
  st=20; [ $w = cfg5 ] && st=8
  echo "== $w $lib"
  if [ $lib = old ]; then export CALISCOPE_BA_LIB=$OLD; else unset CALISCOPE_BA_LIB; fi
  timeout 300 python bench.py --no-cpu --no-first-call --workload $w --also "" --steps $st --warmup 4 2> $O/bench_${w}_$lib.err | tee $O/bench_${w}_$lib.json | python -c "$pick"
  done
done
done > $O/ab.txt 2>&1
unset CALISCOPE_BA_LIB
cat $O/ab.txt
timeout 300 python tools/real_session_timing.py > $O/real_session.txt 2>&1; cat $O/real_session.txt
