#!/bin/bash
# round 6, GPU call 9: pair-kernel lifetimes by tile / XCD / dispatch order under both bindings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r6_run9; mkdir -p $O
P=$GRAFT_REPO_ROOT/caliscope_amd/libcaliscope_ba_prof.so
for b in fine coarse; do
  CBA_BIND=$b CALISCOPE_BA_LIB=$P CBA_STAMPS=1 CBA_PLAN=full timeout 300 python bench.py --no-cpu --no-first-call --workload cfg4 --also "" --steps 12 --warmup 4 > $O/stamps_$b.json 2> $O/stamps_$b.txt
  echo "== $b"; grep -A1 "k_tprep | " $O/stamps_$b.txt | tail -1; grep -A24 "lifetimes by tile" $O/stamps_$b.txt | tail -25
done
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]; print(d["config"]["workload"][:5], "ms_per_step", d["ms_per_step"], {n: round(v["avg_us"],1) for n,v in k.items() if n in ("schur","schur_pairs")})'
for b in fine coarse fine coarse; do
  echo "== $b"; CBA_BIND=$b timeout 300 python bench.py --no-cpu --no-first-call --workload cfg4 --also "" --steps 20 --warmup 4 2>/dev/null | python -c "$pick"
done
