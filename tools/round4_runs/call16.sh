#!/bin/bash
# round 4, call 16: 150 seam calls in a row over three problem sizes (create / solve / destroy each time): host RSS, device memory and time per call — the
# host block pool and the device pool must reach a steady state
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
O=$GRAFT_REPO_ROOT/gpurun_out/r4c16; mkdir -p $O
python - > $O/soak.log 2>&1 <<'PY'
import os, sys, time, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, psutil
import bench
from caliscope_amd import engine_cache
from caliscope_amd.least_squares import least_squares
proc = psutil.Process()
hip = ctypes.CDLL("libamdhip64.so")
def dev_free():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return (t.value - f.value) / 2**20
probs = [bench.build_problem("cfg4"), bench.build_problem("cfg4", n_points=100_000, n_obs=1_000_000), bench.build_problem("cfg5", n_points=100_000, n_obs=1_000_000)]
ref = {}
rows = {0: [], 1: [], 2: []}
for i in range(150):
    k = i % 3
    sc, par, x0, prob, cfg = probs[k]
    engine_cache.clear()
    t = time.perf_counter()
    r = least_squares(None, x0, jac=None, bounds=par.bounds(), x_scale="jac", method="trf", args=(par, sc.camera_indices, sc.image_coords, sc.obj_indices))
    dt = (time.perf_counter() - t) * 1e3
    if k not in ref: ref[k] = (r.nfev, r.cost)
    if i >= 3: rows[k].append((dt, r.setup_seconds * 1e3, r.solve_seconds * 1e3))
    assert r.nfev == ref[k][0] and abs(r.cost - ref[k][1]) <= 1e-9 * ref[k][1], (i, r.nfev, r.cost, ref[k])
    if i % 3 == 2 or i < 6:
        print(f"call {i:3d} (problem {k}): {dt:6.1f} ms (set-up {r.setup_seconds * 1e3:5.1f}, solve {r.solve_seconds * 1e3:5.1f})  RSS {proc.memory_info().rss / 2**20:7.0f} MB  device in use {dev_free():7.0f} MB  threads {proc.num_threads()}", flush=True)
engine_cache.clear()
for k in rows:
    a = np.array(rows[k])
    print(f"problem {k}: {len(a)} warm calls: end to end min / median / max {a[:, 0].min():.1f} / {np.median(a[:, 0]):.1f} / {a[:, 0].max():.1f} ms, set-up {a[:, 1].min():.1f} / {np.median(a[:, 1]):.1f} / {a[:, 1].max():.1f}, solve {a[:, 2].min():.1f} / {np.median(a[:, 2]):.1f} / {a[:, 2].max():.1f}; solves above twice the median: {int((a[:, 2] > 2 * np.median(a[:, 2])).sum())}")
time.sleep(1.5)
print(f"after clear: RSS {proc.memory_info().rss / 2**20:.0f} MB, device in use {dev_free():.0f} MB, threads {proc.num_threads()}")
PY
tail -40 $O/soak.log
