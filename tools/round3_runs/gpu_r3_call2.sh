#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1 CBA_GROUP_TIMEOUT_S=15
O=$GRAFT_REPO_ROOT/gpurun_out/c2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout=300 -k "build_kernel_variants or C128 or evaluation_parity" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log
tail -5 $O/tests.log
timeout 300 python -m pytest tests/test_multi_device.py -m gpu -x -q --timeout=150 > $O/tests_md.log 2>&1; echo "rc=$?" >> $O/tests_md.log
tail -3 $O/tests_md.log
timeout 200 python bench.py --no-cpu --also cfg2 --steps 30 --warmup 6 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
timeout 200 python bench.py --no-cpu --also "" --steps 10 --warmup 3 --gpus 2 --devices 0,0 --xchg direct > $O/bench2.json 2> $O/bench2.err; tail -c 600 $O/bench2.err
python - <<'PY'
import json
for f in ("bench","bench2"):
    try:
        d=json.loads(open(f"gpurun_out/c2/{f}.json").read().strip().splitlines()[-1]); k=d["roofline"]["kernels"]
        print(f, d["ms_per_step"], d["setup_ms"], d["comm_ms_per_step"], d["rccl_ranks"], {x:k[x]["avg_us"] for x in k}, d["final_rms_px"], d["config"]["parallelism"])
    except Exception as e: print(f, "FAILED", e)
PY
