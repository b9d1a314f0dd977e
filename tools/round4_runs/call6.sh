#!/bin/bash
# round 4, call 6: k_small_solve (the dense solve of small rigs in one workgroup): parity tests, cfg2 bench + kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/r4c6; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=300 -k "step_parity and not C1" > $O/tests_a.log 2>&1; tail -3 $O/tests_a.log
timeout 300 python bench.py --no-cpu --workload cfg2 --also "" --steps 40 --warmup 8 > $O/bench_cfg2.json 2> $O/bench_cfg2.err
python - <<PY
import json
d=json.loads(open("$O/bench_cfg2.json").read().strip().splitlines()[-1])
k=d["roofline"]["kernels"]
print("cfg2 ms_per_step", d["ms_per_step"], {n: v.get("avg_us") for n, v in k.items()}, "rms", d["final_rms_px"], "nfev", d["solve"]["nfev"])
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_cfg2 -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --workload cfg2 --also "" --steps 40 --warmup 8 > $O/bench_trace.json 2> $O/trace.err
cd $GRAFT_REPO_ROOT
DB=$(find $O/trace_cfg2 -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB $O/cfg2_kernel_trace.md | head -12
find $O -name "*.db" -size +20M -delete
