#!/bin/bash
# round 4, call 15: where the cheap plan's wall time goes on the box (phases of build_reg2_plan), cfg4 x 6, cfg5 x 3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r4c15; mkdir -p $O
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag 2>/dev/null
python - > $O/seam.log 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from caliscope_amd import engine_cache
from caliscope_amd.least_squares import least_squares
os.environ["CBA_PLAN_TIMING"] = "1"
for name, reps in (("cfg4", 6), ("cfg5", 3)):
    sc, par, x0, prob, cfg = bench.build_problem(name)
    for rep in range(reps):
        engine_cache.clear()
        time.sleep(0.4)
        t = time.perf_counter()
        r = least_squares(None, x0, jac=None, bounds=par.bounds(), x_scale="jac", method="trf", args=(par, sc.camera_indices, sc.image_coords, sc.obj_indices))
        print(f"== {name}: end to end {(time.perf_counter() - t) * 1e3:.1f} ms (set-up {r.setup_seconds * 1e3:.1f}, solve {r.solve_seconds * 1e3:.1f}, nfev {r.nfev})", flush=True)
engine_cache.clear()
PY
grep -n "==\|cheap plan took\|its phases\|plan: upload\|cba_create: " $O/seam.log | cut -c1-170 | tail -90
