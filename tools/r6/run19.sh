#!/bin/bash
# round 6, GPU call 19: k_begin (one launch at the start of a solve instead of six) and four chains in k_reg_finalize's fold role, against the previous library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=30
O=$GRAFT_REPO_ROOT/gpurun_out/r6_run19; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q --timeout=400 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]; print(d["config"]["workload"][:5], "ms_per_step", d["ms_per_step"], {n: round(v["avg_us"],1) for n,v in k.items() if n in ("cam_prep","schur_reduce_finalize","vector_ops")})'
OLD=$GRAFT_REPO_ROOT/caliscope_amd/libcaliscope_ba_old.so
for rep in 1 2; do
for lib in new old; do
  for w in cfg4 cfg2 cfg3; do
  st=20; [ $w = cfg2 ] && st=40
  echo "== $w $lib"
  if [ $lib = old ]; then export CALISCOPE_BA_LIB=$OLD; else unset CALISCOPE_BA_LIB; fi
  timeout 300 python bench.py --no-cpu --no-first-call --workload $w --also "" --steps $st --warmup 4 2> $O/bench_${w}_$lib.err | tee $O/bench_${w}_$lib.json | python -c "$pick"
  done
done
done > $O/ab.txt 2>&1
cat $O/ab.txt
for lib in new old new old; do
  if [ $lib = old ]; then export CALISCOPE_BA_LIB=$OLD; else unset CALISCOPE_BA_LIB; fi
  echo "== session $lib"; timeout 300 python tools/real_session_timing.py 2>&1 | cut -c1-150
done
