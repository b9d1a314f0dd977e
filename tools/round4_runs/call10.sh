#!/bin/bash
# round 4, call 10: the cgroup CPU quota and the plan threads: throttled time (cpu.stat) per seam call on cfg4 for several thread counts
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r4c10; mkdir -p $O
cat /sys/fs/cgroup/cpu.max; cat /sys/fs/cgroup/cpu.stat
python - > $O/seam.log 2>&1 <<'PY'
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from caliscope_amd import engine_cache
from caliscope_amd.least_squares import least_squares
def throttled():
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0)), int(d.get("usage_usec", 0))
    except Exception:
        return (0, 0, 0)
for name in ("cfg4",):
    sc, par, x0, prob, cfg = bench.build_problem(name)
    for fg, bg in ((64, 8), (64, 4), (64, 2), (32, 8), (32, 4), (16, 8), (16, 4), (16, 2), (24, 4)):
        os.environ["CBA_X_FG"] = str(fg); os.environ["CBA_X_BG"] = str(bg)
        rows = []
        for rep in range(8):
            engine_cache.clear()
            time.sleep(0.37)
            a = throttled()
            t = time.perf_counter()
            r = least_squares(None, x0, jac=None, bounds=par.bounds(), x_scale="jac", method="trf", args=(par, sc.camera_indices, sc.image_coords, sc.obj_indices))
            dt = (time.perf_counter() - t) * 1e3
            b = throttled()
            rows.append((dt, r.setup_seconds * 1e3, r.solve_seconds * 1e3, b[0] - a[0], (b[1] - a[1]) / 1e3, (b[2] - a[2]) / 1e3))
        print(f"== {name} fg {fg} bg {bg}: " + "  ".join(f"{d:.0f}/{s:.0f}/{v:.1f}[{n}x {th:.0f}ms cpu {u:.0f}]" for d, s, v, n, th, u in rows), flush=True)
engine_cache.clear()
PY
grep -n "==" $O/seam.log | cut -c1-400
