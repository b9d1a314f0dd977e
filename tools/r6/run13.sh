#!/bin/bash
# round 6, GPU call 13: the dense solve in 64-row blocks (k_chol_step64) against the 32-row one, same box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=30
O=$GRAFT_REPO_ROOT/gpurun_out/r6_run13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_structures.py tests/test_scenarios.py -m gpu -x -q --timeout=400 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -8 $O/tests.log
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]; print(d["config"]["workload"][:5], "ms_per_step", d["ms_per_step"], "rms", d.get("final_rms_px"), {n: round(v["avg_us"],1) for n,v in k.items()})'
for nb in 64 32 64 32; do
for w in cfg4 cfg5 cfg3; do
  st=20; [ $w = cfg5 ] && st=8
  echo "== $w NB=$nb"
  CBA_CHOL_NB=$nb timeout 300 python bench.py --no-cpu --no-first-call --workload $w --also "" --steps $st --warmup 4 2> $O/bench_${w}_$nb.err | tee $O/bench_${w}_$nb.json | python -c "$pick"
done
done > $O/ab.txt 2>&1
cat $O/ab.txt
P=$GRAFT_REPO_ROOT/caliscope_amd/libcaliscope_ba_prof.so
for w in cfg4 cfg5; do
CALISCOPE_BA_LIB=$P CBA_STAMPS=1 CBA_PLAN=full timeout 300 python bench.py --no-cpu --no-first-call --workload $w --also "" --steps 12 --warmup 4 > $O/stamps_$w.json 2> $O/stamps_$w.txt
grep -A40 "k_tprep | " $O/stamps_$w.txt | head -34
done
CALISCOPE_BA_LIB=$P CBA_CHOL_TRACE=1 timeout 200 python tools/chol_trace.py > $O/chol_trace.txt 2>&1; head -20 $O/chol_trace.txt
