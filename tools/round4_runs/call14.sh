#!/bin/bash
# round 4, call 14: set-up with the lean cheap-plan path: seam calls on cfg4 and cfg5, plan timing, the two-stage and plan tests on the device
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r4c14; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "two_stage or limits or step_parity" > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log | tail -2
python - > $O/seam.log 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from caliscope_amd import engine_cache
from caliscope_amd.least_squares import least_squares
os.environ["CBA_PLAN_TIMING"] = "1"
for name, reps in (("cfg4", 10), ("cfg5", 4)):
    sc, par, x0, prob, cfg = bench.build_problem(name)
    for rep in range(reps):
        engine_cache.clear()
        time.sleep(0.4)
        t = time.perf_counter()
        r = least_squares(None, x0, jac=None, bounds=par.bounds(), x_scale="jac", method="trf", args=(par, sc.camera_indices, sc.image_coords, sc.obj_indices))
        print(f"== {name}: end to end {(time.perf_counter() - t) * 1e3:.1f} ms (set-up {r.setup_seconds * 1e3:.1f}, solve {r.solve_seconds * 1e3:.1f}, nfev {r.nfev})", flush=True)
engine_cache.clear()
PY
grep -n "==\|cheap plan took\|plan: upload\|cba_create: 0" $O/seam.log | cut -c1-170 | tail -60
