#!/usr/bin/env python
"""End-to-end wall time of CaptureVolume.optimize() at a bench configuration (DataFrames in, CaptureVolume out):
    python tools/end_to_end.py [cfg2|cfg3|cfg4]
Splits the call into its host stages by timing the same steps separately."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from caliscope_amd.bundle_parameterization import BundleParameterization  # noqa: E402
from caliscope_amd.capture_volume import CaptureVolume  # noqa: E402
from caliscope_amd.engine import BAProblem  # noqa: E402
from caliscope_amd.hip_engine import HipEngine  # noqa: E402
from caliscope_amd.synthetic import make_config  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "cfg4"
t = time.perf_counter()
sc = make_config(name)
print(f"{name}: scene generated in {time.perf_counter() - t:.2f} s ({len(sc.camera_indices)} observations)")
cam_ids = np.array(sorted(sc.cameras_init.cameras))[sc.camera_indices]
t = time.perf_counter()
vol = CaptureVolume.from_arrays(sc.cameras_init, cam_ids, sc.image_coords, sc.obj_indices, sc.points_init)
t_vol = time.perf_counter() - t
HipEngine(BAProblem(*(lambda p: (p, sc.camera_indices[:64], sc.image_coords[:64], sc.obj_indices[:64]))(
    BundleParameterization.from_camera_array(sc.cameras_init, n_points=len(sc.points_init), refine_intrinsics=False)))).close()  # HIP start-up
t = time.perf_counter()
out = vol.optimize(loss=sc.loss, f_scale=vol.pixel_f_scale(1.0) if sc.loss != "linear" else 1.0, refine_intrinsics=sc.refine_intrinsics)
t_opt = time.perf_counter() - t
st = out.optimization_status
# the same stages one by one
t = time.perf_counter(); _, cam, uv, obj = vol._matched_arrays(); t_marshal = time.perf_counter() - t
par = BundleParameterization.from_camera_array(vol.camera_array, n_points=len(vol.world_points), refine_intrinsics=sc.refine_intrinsics)
x0 = par.pack(vol.camera_array, vol.world_points.points)
t = time.perf_counter(); eng = HipEngine(BAProblem(par, cam, uv, obj, loss=sc.loss, f_scale=vol.pixel_f_scale(1.0) if sc.loss != "linear" else 1.0)); t_create = time.perf_counter() - t
lb, ub = par.bounds(); ncp = par.n_camera_params
kw = dict(lb=lb[:ncp], ub=ub[:ncp]) if par.has_finite_bounds else {}
t = time.perf_counter(); res = eng.solve(x0, **kw); t_solve = time.perf_counter() - t
eng.close()
t = time.perf_counter(); rep = out.reprojection_report; t_rep = time.perf_counter() - t
print(f"CaptureVolume(...) construction from DataFrames: {t_vol:.3f} s")
print(f"optimize(): {t_opt:.3f} s total, status {st.termination_reason}, {st.iterations} evaluations")
print(f"  marshalling DataFrames -> arrays {t_marshal:.3f} s | engine set-up (sort, plan, upload) {t_create:.3f} s | "
      f"solve {t_solve:.3f} s ({res.nfev} evaluations) | rest (deepcopy, unpack, new volume) {max(t_opt - t_marshal - t_create - t_solve, 0):.3f} s")
print(f"reprojection_report (device residuals + group-bys): {t_rep:.3f} s, RMS {rep.overall_rmse:.4f} px")
t = time.perf_counter(); flt = out.filter_by_percentile_error(2.5); t_flt = time.perf_counter() - t
print(f"filter_by_percentile_error(2.5) on the cached report: {t_flt:.3f} s ({len(out.image_points) - len(flt.image_points)} observations removed)")
if "--profile-report" in sys.argv:  # where the report's time goes (host group-bys against the device evaluation)
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); out.compute_reprojection_report(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)

