#!/bin/bash
# round 6, GPU call 11: rotating wave priority in the persistent per-observation kernels, A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_run11; mkdir -p $O
P=$GRAFT_REPO_ROOT/caliscope_amd/libcaliscope_ba_prof.so
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]; print(d["config"]["workload"][:5], "ms_per_step", d["ms_per_step"], {n: round(v["avg_us"],1) for n,v in k.items() if n in ("schur","schur_pairs","build","jv","backsub")})'
for w in cfg4 cfg5 cfg3; do
for v in 1 0 1 0; do
  st=20; [ $w = cfg5 ] && st=8
  echo "== $w CBA_ROT_PRIO=$v"; CBA_ROT_PRIO=$v timeout 300 python bench.py --no-cpu --no-first-call --workload $w --also "" --steps $st --warmup 4 2>/dev/null | python -c "$pick"
done; done
for v in 1 0; do
  CBA_ROT_PRIO=$v CALISCOPE_BA_LIB=$P CBA_STAMPS=1 CBA_PLAN=full timeout 300 python bench.py --no-cpu --no-first-call --workload cfg4 --also "" --steps 12 --warmup 4 > /dev/null 2> $O/stamps_$v.txt
  echo "== stamps CBA_ROT_PRIO=$v"; grep -A26 "k_tprep | " $O/stamps_$v.txt | head -27 | grep -v chol_step
done
