#!/bin/bash
# what creating a handle costs on a tiny problem (the reference's 4-camera session)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 200 python - <<'PY' 2>&1 | tail -12
import sys, time
sys.path.insert(0, ".")
from pathlib import Path
from caliscope_amd.bundle_parameterization import BundleParameterization
from caliscope_amd.cameras import CameraArray
from caliscope_amd.capture_volume import CaptureVolume
from caliscope_amd.engine import BAProblem
from caliscope_amd.hip_engine import HipEngine
from caliscope_amd.point_data import ImagePoints, WorldPoints
d = Path("tests/golden/post_optimization")
cv = CaptureVolume(CameraArray.from_toml(d / "camera_array.toml"), ImagePoints.from_csv(d / "xy_CHARUCO.csv"), WorldPoints.from_csv(d / "xyz_CHARUCO.csv"))
_, cam, uv, obj = cv._matched_arrays()
par = BundleParameterization.from_camera_array(cv.camera_array, n_points=len(cv.world_points), refine_intrinsics=False)
x0 = par.pack(cv.camera_array, cv.world_points.points)
prob = BAProblem(par, cam, uv, obj)
HipEngine(prob).close()
for label, kw in (("full handle", {}), ("evaluation-only handle", dict(evaluation_only=True))):
    ts = []
    for _ in range(5):
        t = time.perf_counter(); e = HipEngine(prob, **kw); t1 = time.perf_counter(); e.close(); t2 = time.perf_counter()
        ts.append(((t1 - t) * 1e3, (t2 - t1) * 1e3))
    print(label, "create / destroy ms:", [(round(a, 2), round(b, 2)) for a, b in ts])
e = HipEngine(prob)
t = time.perf_counter(); r = e.solve(x0); print("solve %.2f ms, %d evaluations" % ((time.perf_counter() - t) * 1e3, r.nfev))
t = time.perf_counter(); r = e.solve(x0); print("solve again %.2f ms" % ((time.perf_counter() - t) * 1e3))
e.close()
PY
