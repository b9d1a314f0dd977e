#!/bin/bash
# round 6, GPU call 1: GPU suite on the record-streaming back-substitution, A/B of the kernel, device-side stamps of one iteration
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=30
O=$GRAFT_REPO_ROOT/gpurun_out/r6_run1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q --timeout=400 > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -5 $O/tests.log
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]; print(d["config"]["workload"][:5], "ms_per_step", d["ms_per_step"], "rms", d.get("final_rms_px"), {n: round(v["avg_us"],1) for n,v in k.items()})'
for w in cfg4 cfg5 cfg2; do
  st=20; [ $w = cfg5 ] && st=8; [ $w = cfg2 ] && st=40
  for v in "0 2" "1 2" "1 3"; do
    set -- $v
    echo "== $w CBA_BACKSUB_REC=$1 CBA_BACKSUB_WGS=$2"
    CBA_BACKSUB_REC=$1 CBA_BACKSUB_WGS=$2 timeout 300 python bench.py --no-cpu --no-first-call --workload $w --also "" --steps $st --warmup 4 2> $O/bench_${w}_$1_$2.err | python -c "$pick"
  done
done > $O/ab.txt 2>&1
cat $O/ab.txt
P=$GRAFT_REPO_ROOT/caliscope_amd/libcaliscope_ba_prof.so
for w in cfg4 cfg2 cfg5; do
  CALISCOPE_BA_LIB=$P CBA_STAMPS=1 CBA_PLAN=full timeout 300 python bench.py --no-cpu --no-first-call --workload $w --also "" --steps 12 --warmup 4 > $O/stamps_$w.json 2> $O/stamps_$w.txt
done
grep -A40 "device stamps" $O/stamps_cfg4.txt | head -60
