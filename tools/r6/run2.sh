#!/bin/bash
# round 6, GPU call 2: stamps with workgroup lifetimes, back-substitution occupancy, the new bounded-retry test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=30
O=$GRAFT_REPO_ROOT/gpurun_out/r6_run2; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=400 -k "failed_factorisation or bounded or step_parity" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -5 $O/tests.log
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]; print(d["config"]["workload"][:5], "ms_per_step", d["ms_per_step"], "rms", d.get("final_rms_px"), {n: round(v["avg_us"],1) for n,v in k.items()})'
for v in 2 3 4; do
  echo "== cfg4 CBA_BACKSUB_WGS=$v"
  CBA_BACKSUB_WGS=$v timeout 300 python bench.py --no-cpu --no-first-call --workload cfg4 --also "" --steps 20 --warmup 4 2> $O/bench_cfg4_$v.err | python -c "$pick"
done > $O/ab.txt 2>&1
cat $O/ab.txt
P=$GRAFT_REPO_ROOT/caliscope_amd/libcaliscope_ba_prof.so
for w in cfg4 cfg3 cfg5; do
  CALISCOPE_BA_LIB=$P CBA_STAMPS=1 CBA_PLAN=full CBA_BACKSUB_WGS=3 timeout 300 python bench.py --no-cpu --no-first-call --workload $w --also "" --steps 12 --warmup 4 > $O/stamps_$w.json 2> $O/stamps_$w.txt
done
grep -B2 -A30 "k_tprep | " $O/stamps_cfg4.txt | tail -34
grep -A60 "k_tprep | " $O/stamps_cfg5.txt | grep -v chol_step | tail -16
