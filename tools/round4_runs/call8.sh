#!/bin/bash
# round 4, call 8: the two-stage plan with the dealt plan made resident by the plan thread: seam calls on cfg4 (full against swap), the two-stage
# test, and cfg5's set-up
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r4c8; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "two_stage or limits or thousand" > $O/tests.log 2>&1; tail -3 $O/tests.log
python - > $O/seam.log 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from caliscope_amd import engine_cache
from caliscope_amd.least_squares import least_squares
for name in ("cfg4", "cfg5"):
    sc, par, x0, prob, cfg = bench.build_problem(name)
    for mode in ("full", "swap", "swap", "swap", "full", "swap"):
        os.environ["CBA_PLAN"] = mode
        engine_cache.clear()
        t = time.perf_counter()
        r = least_squares(None, x0, jac=None, bounds=par.bounds(), x_scale="jac", method="trf", args=(par, sc.camera_indices, sc.image_coords, sc.obj_indices))
        print(f"== {name} {mode}: end to end {(time.perf_counter() - t) * 1e3:.1f} ms (set-up {r.setup_seconds * 1e3:.1f}, solve {r.solve_seconds * 1e3:.1f}, nfev {r.nfev})", flush=True)
        time.sleep(1.0)
    os.environ.pop("CBA_PLAN", None)
    engine_cache.clear()
    t = time.perf_counter()
    r = least_squares(None, x0, jac=None, bounds=par.bounds(), x_scale="jac", method="trf", args=(par, sc.camera_indices, sc.image_coords, sc.obj_indices))
    print(f"== {name} default: end to end {(time.perf_counter() - t) * 1e3:.1f} ms (set-up {r.setup_seconds * 1e3:.1f}, solve {r.solve_seconds * 1e3:.1f}, nfev {r.nfev})", flush=True)
engine_cache.clear()
PY
grep -n "==" $O/seam.log | cut -c1-160
