#!/bin/bash
# round 6, GPU call 12: factored linearisation (no J_l per observation) in k_jv, k_tprep, k_build_cs; full pipelining in k_tprep<9>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=30
O=$GRAFT_REPO_ROOT/gpurun_out/r6_run12; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --timeout=400 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -5 $O/tests.log
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]; print(d["config"]["workload"][:5], "ms_per_step", d["ms_per_step"], "rms", d.get("final_rms_px"), {n: round(v["avg_us"],1) for n,v in k.items()})'
for w in cfg4 cfg4 cfg5 cfg3 cfg2; do
  st=20; [ $w = cfg5 ] && st=8; [ $w = cfg2 ] && st=40
  echo "== $w"
  timeout 300 python bench.py --no-cpu --no-first-call --workload $w --also "" --steps $st --warmup 4 2> $O/bench_$w.err | tee $O/bench_$w.json | python -c "$pick"
done > $O/ab.txt 2>&1
cat $O/ab.txt
P=$GRAFT_REPO_ROOT/caliscope_amd/libcaliscope_ba_prof.so
for w in cfg4 cfg5; do
CALISCOPE_BA_LIB=$P CBA_STAMPS=1 CBA_PLAN=full timeout 300 python bench.py --no-cpu --no-first-call --workload $w --also "" --steps 12 --warmup 4 > $O/stamps_$w.json 2> $O/stamps_$w.txt
grep -A60 "k_tprep | " $O/stamps_$w.txt | head -61 | grep -v "chol_step" | head -14
done
