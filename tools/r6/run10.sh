#!/bin/bash
# round 6, GPU call 10: alternating wave priority between the two workgroups of a CU in the pair kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_run10; mkdir -p $O
P=$GRAFT_REPO_ROOT/caliscope_amd/libcaliscope_ba_prof.so
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]; print(d["config"]["workload"][:5], "ms_per_step", d["ms_per_step"], {n: round(v["avg_us"],1) for n,v in k.items() if n in ("schur","schur_pairs")})'
for v in -1 0 1 2 3 -1 0 1; do
  echo "== CBA_PAIR_PRIO=$v"; CBA_PAIR_PRIO=$v timeout 300 python bench.py --no-cpu --no-first-call --workload cfg4 --also "" --steps 20 --warmup 4 2>/dev/null | python -c "$pick"
done
for v in -1 0 2; do
  CBA_PAIR_PRIO=$v CALISCOPE_BA_LIB=$P CBA_STAMPS=1 CBA_PLAN=full timeout 300 python bench.py --no-cpu --no-first-call --workload cfg4 --also "" --steps 12 --warmup 4 > /dev/null 2> $O/stamps_$v.txt
  echo "== stamps CBA_PAIR_PRIO=$v"; grep -A1 "k_tprep | " $O/stamps_$v.txt | head -2 | tail -1; grep -A2 "by dispatch order" $O/stamps_$v.txt | head -2
done
for v in -1 0 2; do
  echo "== cfg5 CBA_PAIR_PRIO=$v"; CBA_PAIR_PRIO=$v timeout 300 python bench.py --no-cpu --no-first-call --workload cfg5 --also "" --steps 8 --warmup 3 2>/dev/null | python -c "$pick"
done
