#!/bin/bash
# cfg5: is the pair kernel bound by the bytes it gathers?  Variants whose gathers all land in a window of records (wrong sums, timing only):
# 4k records = 0.7 MB (L2), 16k = 2.9 MB (L2, barely), 1M = 184 MB (Infinity Cache) against the product (1.76 GB of records, every one gathered by 8 tiles).
# + where cba_create spends its time (CBA_PLAN_TIMING) for cfg4 and half-cfg4.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/c3; mkdir -p $O
run() {
  local lib="X_UNUSED=1"; [ -n "$2" ] && lib="CALISCOPE_BA_LIB=$GRAFT_REPO_ROOT/tools/exp/$2"
  env $lib timeout 200 python bench.py --no-cpu --also "" --workload $3 --steps 12 --warmup 3 > $O/$1.json 2> $O/$1.err
  python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/c3/{sys.argv[1]}.json").read().strip().splitlines()[-1]); k = d["roofline"]["kernels"]
    print(sys.argv[1], d["ms_per_step"], "setup", d["setup_ms"], {x: k[x]["avg_us"] for x in k}, "nfev", d["solve"]["nfev"])
except Exception as e:
    print(sys.argv[1], "FAILED", e); print(open(f"gpurun_out/c3/{sys.argv[1]}.err").read()[-800:])
PY
}
run cfg5_product "" cfg5
run cfg5_win4k libcba_win4k.so cfg5
run cfg5_win16k libcba_win16k.so cfg5
run cfg5_win1m libcba_win1m.so cfg5
run cfg4_win4k libcba_win4k.so cfg4
timeout 200 python tools/create_timing.py > $O/create.log 2>&1; tail -40 $O/create.log
