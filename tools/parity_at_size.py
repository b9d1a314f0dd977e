#!/usr/bin/env python
"""Parity at BASELINE.json's sizes: the product's solver on the GPU against the reference's scipy call (oracle callables) on
identical arrays and x0, written to profiles/parity_r03.json by the last GPU run of the round.

    python tools/parity_at_size.py [out.json] [--skip-converged]

cfg2            8 cams / 5k points / 40k obs, linear loss, tight tolerances (both solvers reach the minimum)
cfg3_product    32 cams / 50k points / 400k obs, 5 % outliers, Huber at 1 px, ftol 1e-4 and max_nfev 60 — the settings
                ``calibrate_extrinsics`` passes for its robust stage (reference core/calibrate_extrinsics.py:231-238): neither solver
                converges in 60 evaluations, the comparison is cost and RMS at the stopping point
cfg3_converged  the same arrays run to scipy's OWN convergence at the reference's default tolerances (ftol = xtol = gtol = 1e-8, no
                evaluation cap: ~530 evaluations, minutes of one host core — BASELINE.md 1b), the product with the same settings
cfg3_tight      the product alone at 1e-13: how far the default-tolerance stopping points are from the minimum (both solvers stop at
                ftol while still crawling; their distance to each other is bounded by their distances to this point)
(full cfg4 and a cfg5-recipe sample are compared in every bench.py run: ``parity`` / ``also.cfg5.parity`` in its JSON line)
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


_REF_CACHE = {}


def run(name, tol, ref_from=None, tight_gpu=False, polish=False):
    sc, par, x0, prob, cfg = bench.build_problem(name)
    fs = prob.f_scale
    t0 = time.perf_counter()
    if ref_from is not None:  # compare the product with a scipy solution computed earlier in this run (other product settings, same arrays)
        ref, t_ref = _REF_CACHE[ref_from]
    else:
        ref = optimize_scipy(par, sc.camera_indices, sc.image_coords, sc.obj_indices, x0, loss=prob.loss, f_scale=fs, **tol)
        t_ref = time.perf_counter() - t0
        _REF_CACHE[(name, json.dumps(tol, sort_keys=True))] = (ref, t_ref)
    if tight_gpu:
        tol = dict(ftol=1e-13, xtol=1e-13, gtol=1e-13, max_nfev=20000)
    t0 = time.perf_counter()
    got = least_squares(None, x0, jac=None, bounds=par.bounds(), x_scale="jac", loss=prob.loss, f_scale=fs, method="trf",
                        args=(par, sc.camera_indices, sc.image_coords, sc.obj_indices), **tol)
    t_got = time.perf_counter() - t0
    fx = np.array([b.fx_initial for b in par.blocks])[sc.camera_indices]

    def rms(x):
        e = joint_residuals(x, par, sc.camera_indices, sc.image_coords, sc.obj_indices).reshape(-1, 2) * fx[:, None]
        return float(np.sqrt(np.mean(np.sum(e * e, axis=1))))

    pos, ang = bench.solution_parity(par, got.x, ref.x)
    extra = {}
    if polish:
        # Is the distance between the two stopping points the solvers' or the problem's?  The product started again from scipy's stopping
        # point, tight tolerances: where it ends is the minimum nearest to scipy's answer.  `scipy_to_its_minimum` is how far scipy stopped
        # from it, `minimum_from_x0_vs_minimum_from_scipy` whether the product's own solve (from x0) found the same minimum.
        tight = dict(ftol=1e-13, xtol=1e-13, gtol=1e-13, max_nfev=20000)
        near = least_squares(None, ref.x, jac=None, bounds=par.bounds(), x_scale="jac", loss=prob.loss, f_scale=fs, method="trf",
                             args=(par, sc.camera_indices, sc.image_coords, sc.obj_indices), **tight)
        extra["polish"] = {
            "gpu_from_scipy_x": {"nfev": int(near.nfev), "status": int(near.status), "cost": float(near.cost), "rms_px": rms(near.x)},
            "scipy_to_its_minimum": bench.solution_parity(par, ref.x, near.x, detail=True),
            "minimum_from_x0_vs_minimum_from_scipy": bench.solution_parity(par, got.x, near.x, detail=True),
            "rel_cost_scipy_above_minimum": (float(ref.cost) - float(near.cost)) / float(near.cost),
            "rel_cost_gpu_above_minimum": (float(got.cost) - float(near.cost)) / float(near.cost),
        }
    return {**extra, "detail": bench.solution_parity(par, got.x, ref.x, detail=True),
        "workload": f"{name}: {len(par.blocks)} cams / {par.n_points} points / {prob.n_obs} obs, {prob.loss} loss", "settings": tol,
        "scipy": {"nfev": int(ref.nfev), "njev": int(ref.njev), "status": int(ref.status), "cost": float(ref.cost), "rms_px": rms(ref.x), "seconds": round(t_ref, 2)},
        "gpu": {"nfev": int(got.nfev), "njev": int(got.njev), "accepted_steps": int(got.njev) - 1, "rejected_trials": int(got.nfev) - int(got.njev),
                "status": int(got.status), "cost": float(got.cost), "rms_px": rms(got.x), "seconds_end_to_end": round(t_got, 3)},
        "d_rms_px": rms(got.x) - rms(ref.x), "rel_cost": (float(got.cost) - float(ref.cost)) / float(ref.cost),
        "aligned_pos": pos, "aligned_ang_rad": ang,
    }


if __name__ == "__main__":
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    out = {
        "cfg2": run("cfg2", dict(ftol=1e-13, xtol=1e-13, gtol=1e-13, max_nfev=400)),
        "cfg3_product": run("cfg3", dict(ftol=1e-4, xtol=1e-8, gtol=1e-8, max_nfev=60)),
        "host_cores": os.cpu_count(),
    }
    if "--skip-converged" not in sys.argv:
        default = dict(ftol=1e-8, xtol=1e-8, gtol=1e-8, max_nfev=None)
        out["cfg3_converged"] = run("cfg3", default)
        out["cfg3_tight"] = run("cfg3", default, ref_from=("cfg3", json.dumps(default, sort_keys=True)), tight_gpu=True, polish=True)
        out["cfg3_tight"]["note"] = "gpu at 1e-13 against scipy at its default-tolerance stopping point"
    if "--skip-cfg5" not in sys.argv:  # SURVEY.md 8d's cfg5 sample: 128 cams / 100k points / 1M obs, free intrinsics + bounds (~96 scipy evaluations)
        t0 = time.perf_counter()
        out["cfg5_sample_1M"] = bench.cfg5_sample_parity(n_points=100_000)
        out["cfg5_sample_1M"]["wall_seconds"] = round(time.perf_counter() - t0, 1)
    path = argv[0] if argv else "profiles/parity_r03.json"
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
