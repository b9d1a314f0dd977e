#!/bin/bash
# round 6, GPU call 17: k_reg_reduce + k_schur_finalize as one launch (k_reg_finalize) against the two launches (CBA_REG_FINALIZE=0), same box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=30
O=$GRAFT_REPO_ROOT/gpurun_out/r6_run17; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_structures.py -m gpu -x -q --timeout=400 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -5 $O/tests.log
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]; print(d["config"]["workload"][:5], "ms_per_step", d["ms_per_step"], "rms", d.get("final_rms_px"), {n: round(v["avg_us"],1) for n,v in k.items() if n in ("schur","schur_reduce_finalize","cholesky_solve")})'
for rep in 1 2; do
for f in 1 0; do
  for w in cfg4 cfg3 cfg5; do
  st=20; [ $w = cfg5 ] && st=8
  echo "== $w fused=$f"
  CBA_REG_FINALIZE=$f timeout 300 python bench.py --no-cpu --no-first-call --workload $w --also "" --steps $st --warmup 4 2> $O/bench_${w}_$f.err | tee $O/bench_${w}_$f.json | python -c "$pick"
  done
done
done > $O/ab.txt 2>&1
cat $O/ab.txt
P=$GRAFT_REPO_ROOT/caliscope_amd/libcaliscope_ba_prof.so
for w in cfg4 cfg5; do
CALISCOPE_BA_LIB=$P CBA_STAMPS=1 CBA_PLAN=full timeout 300 python bench.py --no-cpu --no-first-call --workload $w --also "" --steps 12 --warmup 4 > $O/stamps_$w.json 2> $O/stamps_$w.txt
grep -A8 "k_tprep | " $O/stamps_$w.txt | head -9
done
