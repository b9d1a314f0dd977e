#!/bin/bash
# round 6, GPU call 16: chunks dealt together (region size of the plan) swept on cfg5 and cfg4: lane utilisation against the reach of the gathers
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=30
O=$GRAFT_REPO_ROOT/gpurun_out/r6_run16; mkdir -p $O
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]; print(d["config"]["workload"][:5], "ms_per_step", d["ms_per_step"], {n: round(v["avg_us"],1) for n,v in k.items() if n in ("schur","schur_pairs")}, "lane_util", d.get("plan",{}).get("lane_utilisation"))'
for w in cfg5 cfg4; do
for rg in 32 8 16 64 128 32; do
  st=20; [ $w = cfg5 ] && st=8
  echo "== $w region $rg"
  CBA_PLAN_REGION=$rg CBA_PLAN_TIMING=1 timeout 300 python bench.py --no-cpu --no-first-call --workload $w --also "" --steps $st --warmup 4 2> $O/bench_${w}_$rg.err | tee $O/bench_${w}_$rg.json | python -c "$pick"
  grep "lane utilisation" $O/bench_${w}_$rg.err | tail -1
done
done > $O/ab.txt 2>&1
cat $O/ab.txt
