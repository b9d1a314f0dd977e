#!/bin/bash
# round 4, call 11: seam calls on cfg4 with the two-stage plan, 24 in a row (a stall of the solve shows as "solve" >> 3.4 ms)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r4c11; mkdir -p $O
python - > $O/seam.log 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from caliscope_amd import engine_cache
from caliscope_amd.least_squares import least_squares
sc, par, x0, prob, cfg = bench.build_problem("cfg4")
os.environ["CBA_PLAN_TIMING"] = "1"
for rep in range(24):
    engine_cache.clear()
    time.sleep(0.3)
    t = time.perf_counter()
    r = least_squares(None, x0, jac=None, bounds=par.bounds(), x_scale="jac", method="trf", args=(par, sc.camera_indices, sc.image_coords, sc.obj_indices))
    print(f"== cfg4: end to end {(time.perf_counter() - t) * 1e3:.1f} ms (set-up {r.setup_seconds * 1e3:.1f}, solve {r.solve_seconds * 1e3:.1f}, nfev {r.nfev})", flush=True)
engine_cache.clear()
PY
grep -n "==\|XX" $O/seam.log | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
