#!/usr/bin/env python
"""scipy reference solutions at BASELINE sizes, computed on the CPU (no GPU needed) and committed as fixtures.

    python tests/golden/make_scipy_refs.py [name ...]          # all, or a subset of CASES

The reference's solver call (oracle/solver.py = core/capture_volume.py:387-411 on the oracle callables; scipy 1.15.3,
single-threaded) takes minutes at these sizes: cfg3 to its own convergence ~530 evaluations, the 1M-observation cfg5 sample ~100.
Run beside the product on the GPU box they were 12 of the 13 minutes of the parity call.  They do not depend on the product, so
they are computed HERE once and stored: ``tests/golden/scipy_refs/<case>.npz`` holds the solution ``x``, ``nfev / njev / status /
cost``, the wall seconds, the settings, and ``x0_sha256`` — the digest of the start vector, which the consumers
(tools/parity_at_size.py, bench.py, tests/test_gpu_parity.py) compare with their own x0 before trusting the file.

SURVEY.md 7 hard part 1, protocol (iii): both solvers from the same x0, scipy at the reference's defaults AND at
ftol = xtol = gtol = 1e-15.  The ``*_tight`` cases are the second reference.
"""
import hashlib
import json
import os
import sys
import time
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
import numpy as np  # noqa: E402

OUT = HERE / "scipy_refs"
DEFAULT = dict(ftol=1e-8, xtol=1e-8, gtol=1e-8, max_nfev=None)
TIGHT = dict(ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=None)
# The tight reference of the larger cases.  With scipy's default LSMR tolerances (atol = btol = 1e-6, scipy:optimize/_lsq/trf.py:433-437) the
# inner solves stop short along weakly determined directions (focal scale against camera distance): measured on 6 cameras / 300 points with free
# intrinsics, 2000 evaluations at ftol = 1e-15 leave the intrinsics 5e-3 from the point scipy's own tr_solver="exact" reaches in 53.  Tightening
# the INNER tolerance as well (tr_options, the route SURVEY.md 7 protocol (ii) names) makes the LSMR steps exact to rounding: scipy then converges
# like its exact solver and stops on its own criteria.  max_nfev bounds the run, it is not expected to bind.
TIGHT_LSMR = dict(ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=3000, tr_options=dict(atol=1e-14, btol=1e-14))

# name -> (problem, settings).  problem: ("config", name) = bench.build_problem(name); ("cfg5-sample", n_points) = bench's cfg5 recipe;
# ("small", kwargs) = tests.helpers.small_problem(**kwargs)
CASES = {
    "cfg2_tight": (("config", "cfg2"), TIGHT),
    "cfg3_default": (("config", "cfg3"), DEFAULT),
    "cfg3_tight": (("config", "cfg3"), TIGHT_LSMR),
    "cfg5s100k_default": (("cfg5-sample", 10_000), DEFAULT),
    "cfg5s100k_tight": (("cfg5-sample", 10_000), TIGHT_LSMR),
    "cfg5s1M_default": (("cfg5-sample", 100_000), DEFAULT),
    "cfg5s1M_tight": (("cfg5-sample", 100_000), TIGHT_LSMR),
    # tests/test_gpu_parity.py CASES["pinhole_refine_C6"] against scipy's OWN exact trust-region solver (dense SVD steps: minutes of one core)
    "refine_C6_exact": (("small", dict(n_cams=6, n_points=300, k=6, refine=True)), dict(ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=400, tr_solver="exact")),
}


def x0_digest(x0):
    return hashlib.sha256(np.ascontiguousarray(x0, dtype=np.float64).tobytes()).hexdigest()


def problem(spec):
    """(scene, parameterization, x0, loss, f_scale) of a case — the arrays bench.py and tools/parity_at_size.py build."""
    import bench

    kind, arg = spec
    if kind == "small":
        from tests.helpers import small_problem

        sc, par, x0 = small_problem(**arg)
        return sc, par, x0, "linear", 1.0
    if kind == "config":
        sc, par, x0, prob, _ = bench.build_problem(arg)
        return sc, par, x0, prob.loss, prob.f_scale
    from caliscope_amd.bundle_parameterization import BundleParameterization
    from caliscope_amd.synthetic import make_scene

    sc = make_scene("cfg5-sample", n_cams=128, n_points=arg, n_obs=10 * arg, refine=True)
    par = BundleParameterization.from_camera_array(sc.cameras_init, n_points=arg, refine_intrinsics=True)
    return sc, par, par.pack(sc.cameras_init, sc.points_init), "linear", 1.0


def load(name, x0=None):
    """The stored solution of a case as a dict, or None when the file is absent or belongs to another x0."""
    path = OUT / f"{name}.npz"
    if not path.exists():
        return None
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    if x0 is not None and meta["x0_sha256"] != x0_digest(x0):
        return None
    return {"x": z["x"], **meta}


def make(name):
    from oracle.solver import optimize_scipy

    spec, tol = CASES[name]
    sc, par, x0, loss, f_scale = problem(spec)
    t0 = time.perf_counter()
    res = optimize_scipy(par, sc.camera_indices, sc.image_coords, sc.obj_indices, x0, loss=loss, f_scale=f_scale, **tol)
    dt = time.perf_counter() - t0
    import scipy

    meta = {"case": name, "settings": tol, "loss": loss, "f_scale": f_scale, "nfev": int(res.nfev), "njev": int(res.njev),
            "status": int(res.status), "cost": float(res.cost), "optimality": float(res.optimality), "seconds": round(dt, 2),
            "x0_sha256": x0_digest(x0), "n": int(x0.size), "n_obs": int(sc.n_obs), "scipy": scipy.__version__, "numpy": np.__version__,
            "host_cores": os.cpu_count()}
    OUT.mkdir(exist_ok=True)
    np.savez(OUT / f"{name}.npz", x=res.x, meta=np.array(json.dumps(meta)))
    print(json.dumps(meta), flush=True)


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(CASES)):
        make(n)
