#!/bin/bash
# round 4, call 9: which call of the solver thread waits when the plan thread makes the dealt plan resident beside a running solve
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r4c9; mkdir -p $O
python - > $O/seam.log 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from caliscope_amd import engine_cache
from caliscope_amd.least_squares import least_squares
os.environ["CBA_SOLVE_TRACE"] = "1"
os.environ["CBA_PLAN_TIMING"] = "1"
sc, par, x0, prob, cfg = bench.build_problem("cfg4")
for mode in ("full", "swap", "swap", "swap", "swap", "swap", "swap"):
    os.environ["CBA_PLAN"] = mode
    engine_cache.clear()
    t = time.perf_counter()
    r = least_squares(None, x0, jac=None, bounds=par.bounds(), x_scale="jac", method="trf", args=(par, sc.camera_indices, sc.image_coords, sc.obj_indices))
    print(f"== cfg4 {mode}: end to end {(time.perf_counter() - t) * 1e3:.1f} ms (set-up {r.setup_seconds * 1e3:.1f}, solve {r.solve_seconds * 1e3:.1f}, nfev {r.nfev})", flush=True)
    time.sleep(1.0)
engine_cache.clear()
PY
grep -n "==\|longest\|upload\|binding\|swapped\|dealt" $O/seam.log | cut -c1-200
