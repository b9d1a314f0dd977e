#!/usr/bin/env python
"""Parity at BASELINE.json's sizes: the product's solver on the GPU against the reference's scipy call (oracle callables) on
identical arrays and x0, written to profiles/parity_r02.json.

    python tools/parity_at_size.py [out.json]

cfg2  8 cams / 5k points / 40k obs, linear loss, tight tolerances (both solvers reach the minimum)
cfg3  32 cams / 50k points / 400k obs, 5 % outliers, Huber at 1 px, ftol 1e-4 and max_nfev 60 — the settings
      ``calibrate_extrinsics`` passes for its robust stage (reference core/calibrate_extrinsics.py:231-238): neither solver
      converges in 60 evaluations, the comparison is cost and RMS at the stopping point
(the half-cfg4 sample is compared in every bench.py run: ``parity`` in its JSON line)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import bench
from caliscope_amd.least_squares import least_squares
from oracle.residuals import joint_residuals
from oracle.solver import optimize_scipy


def run(name, tol):
    sc, par, x0, prob, cfg = bench.build_problem(name)
    fs = prob.f_scale
    t0 = time.perf_counter()
    ref = optimize_scipy(par, sc.camera_indices, sc.image_coords, sc.obj_indices, x0, loss=prob.loss, f_scale=fs, **tol)
    t_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    got = least_squares(None, x0, jac=None, bounds=par.bounds(), x_scale="jac", loss=prob.loss, f_scale=fs, method="trf",
                        args=(par, sc.camera_indices, sc.image_coords, sc.obj_indices), **tol)
    t_got = time.perf_counter() - t0
    fx = np.array([b.fx_initial for b in par.blocks])[sc.camera_indices]

    def rms(x):
        e = joint_residuals(x, par, sc.camera_indices, sc.image_coords, sc.obj_indices).reshape(-1, 2) * fx[:, None]
        return float(np.sqrt(np.mean(np.sum(e * e, axis=1))))

    pos, ang = bench.solution_parity(par, got.x, ref.x)
    return {
        "workload": f"{name}: {len(par.blocks)} cams / {par.n_points} points / {prob.n_obs} obs, {prob.loss} loss", "settings": tol,
        "scipy": {"nfev": int(ref.nfev), "njev": int(ref.njev), "status": int(ref.status), "cost": float(ref.cost), "rms_px": rms(ref.x), "seconds": round(t_ref, 2)},
        "gpu": {"nfev": int(got.nfev), "njev": int(got.njev), "accepted_steps": int(got.njev) - 1, "rejected_trials": int(got.nfev) - int(got.njev),
                "status": int(got.status), "cost": float(got.cost), "rms_px": rms(got.x), "seconds_end_to_end": round(t_got, 3)},
        "d_rms_px": rms(got.x) - rms(ref.x), "rel_cost": (float(got.cost) - float(ref.cost)) / float(ref.cost),
        "aligned_pos": pos, "aligned_ang_rad": ang,
    }


if __name__ == "__main__":
    out = {
        "cfg2": run("cfg2", dict(ftol=1e-13, xtol=1e-13, gtol=1e-13, max_nfev=400)),
        "cfg3": run("cfg3", dict(ftol=1e-4, xtol=1e-8, gtol=1e-8, max_nfev=60)),
        "host_cores": os.cpu_count(),
    }
    path = sys.argv[1] if len(sys.argv) > 1 else "profiles/parity_r02.json"
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
