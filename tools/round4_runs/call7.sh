#!/bin/bash
# round 4, call 7: where a seam call on full cfg4 loses time when the dealt plan is swapped in mid-solve
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r4c7; mkdir -p $O
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; cat /proc/loadavg
python - > $O/seam.log 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from caliscope_amd import engine_cache
from caliscope_amd.least_squares import least_squares
sc, par, x0, prob, cfg = bench.build_problem("cfg4")
for mode, trace in (("full", "0"), ("swap", "0"), ("swap", "0"), ("swap", "1"), ("full", "0"), ("swap", "0")):
    os.environ["CBA_PLAN"] = mode
    if trace == "1": os.environ["CBA_SOLVE_TRACE"] = "1"
    else: os.environ.pop("CBA_SOLVE_TRACE", None)
    engine_cache.clear()
    t = time.perf_counter()
    r = least_squares(None, x0, jac=None, bounds=par.bounds(), x_scale="jac", method="trf", args=(par, sc.camera_indices, sc.image_coords, sc.obj_indices))
    print(f"== {mode}: end to end {(time.perf_counter() - t) * 1e3:.1f} ms (set-up {r.setup_seconds * 1e3:.1f}, solve {r.solve_seconds * 1e3:.1f}, nfev {r.nfev})", flush=True)
    time.sleep(0.5)
engine_cache.clear()
PY
grep -n "==\|cba_solve trace\|calls" $O/seam.log | cut -c1-160
