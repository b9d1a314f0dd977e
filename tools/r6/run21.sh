#!/bin/bash
# round 6, GPU call 21: eight loads in flight in the row sums of k_reg_reduce / k_reg_finalize (cfg3: the diagonal camera blocks were 38 dependent steps deep), against the previous library
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=30
O=$GRAFT_REPO_ROOT/gpurun_out/r6_run21; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q --timeout=400 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -4 $O/tests.log
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]; print(d["config"]["workload"][:5], "ms_per_step", d["ms_per_step"], {n: round(v["avg_us"],1) for n,v in k.items() if n in ("schur_reduce_finalize",)})'
OLD=$GRAFT_REPO_ROOT/caliscope_amd/libcaliscope_ba_old.so
for rep in 1 2 3; do
for lib in new old; do
  for w in cfg3 cfg2 cfg4; do
  st=20; [ $w = cfg2 ] && st=40
  echo "== $w $lib"
  if [ $lib = old ]; then export CALISCOPE_BA_LIB=$OLD; else unset CALISCOPE_BA_LIB; fi
  timeout 300 python bench.py --no-cpu --no-first-call --workload $w --also "" --steps $st --warmup 4 2> $O/bench_${w}_$lib.err | tee $O/bench_${w}_$lib.json | python -c "$pick"
  done
done
done > $O/ab.txt 2>&1
unset CALISCOPE_BA_LIB
cat $O/ab.txt
P=$GRAFT_REPO_ROOT/caliscope_amd/libcaliscope_ba_prof.so
for w in cfg2 cfg3 cfg4; do
CALISCOPE_BA_LIB=$P CBA_STAMPS=1 CBA_PLAN=full timeout 300 python bench.py --no-cpu --no-first-call --workload $w --also "" --steps 12 --warmup 4 > $O/stamps_$w.json 2> $O/stamps_$w.txt
egrep "k_reg_reduce|kernels," $O/stamps_$w.txt | head -2
done
