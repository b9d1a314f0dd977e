"""Static markers at scale: 3 static markers seen by 6 cameras in 600 frames (~3300 rows per corner, 13 chunk fragments each) plus a
mobile marker per frame, 3620 constraint rows — device solve against the numpy engine.  python tools/heavy_stress.py"""
import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from tests.constrained_scene import marker_volume
from caliscope_amd.engine import BAProblem
from caliscope_amd.hip_engine import HipEngine
from oracle.engine import OracleEngine
from oracle.trf_driver import trf_solve
from tests.helpers import aligned_difference
t=time.time(); vol, par = marker_volume(n_frames=600, n_markers=3, n_cams=6); print("scene", time.time()-t)
_, cam, uv, obj = vol._matched_arrays()
ga, gb, dist, sig = vol._build_constraint_arrays()
con = (ga, gb, dist, (1.0/1394.6)/sig)
x0 = par.pack(vol.camera_array, vol.world_points.points)
print("obs", len(cam), "points", par.n_points, "constraints", len(dist), "max obs/point", np.bincount(obj).max())
prob = BAProblem(par, cam, uv, obj, constraint_groups_a=con[0], constraint_groups_b=con[1], constraint_distances=con[2], constraint_weights=con[3])
t=time.time(); hip = HipEngine(prob); print("create", time.time()-t, hip.info()["n_heavy_points"])
hip.solve(x0)
t=time.time(); got = hip.solve(x0); dt=time.time()-t
print("hip solve", got.status, got.nfev, got.cost, "%.3f ms, %.1f us/eval" % (dt*1e3, dt*1e6/max(got.nfev-1,1)))
t=time.time(); ora = OracleEngine(par, cam, uv, obj, constraints=con); ref = trf_solve(ora, x0); print("oracle solve", ref.status, ref.nfev, ref.cost, time.time()-t)
print("aligned diff", aligned_difference(par, got.x, ref.x), "cost rel", abs(got.cost-ref.cost)/ref.cost)
