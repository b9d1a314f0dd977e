#!/usr/bin/env python
"""Parity at BASELINE.json's sizes: the product's solver on the GPU against the reference's scipy call (oracle callables) on
identical arrays and x0.  The round's LAST GPU call runs it, after the last kernel change; the file carries the digest of the library
sources it ran on (tests/test_bench_contract.py compares it with the committed bench line's).

    python tools/parity_at_size.py [out.json] [--skip-cfg5-1M]

Both directions of SURVEY.md 7's converged-level protocol (iii), per case:
  * the product against scipy from the same x0 — scipy at the reference's defaults AND at ftol = xtol = gtol = 1e-15 (inner LSMR 1e-14);
    those scipy solves take minutes of one host core and do not depend on the product, so they come from tests/golden/scipy_refs/
    (tests/golden/make_scipy_refs.py; the x0 digest in the file must match) and are recomputed only when absent;
  * ``oracle_polish``: scipy started AT the product's converged x with 1e-15 tolerances — it must stop within a few evaluations, having
    moved (gauge-aligned) by less than 1e-6 and gained less than 1e-12 of the cost.

cfg2            8 cams / 5k points / 40k obs, linear: product at 1e-13 against scipy at 1e-15
cfg3_product    32 cams / 50k points / 400k obs, 5 % outliers, Huber at 1 px, ftol 1e-4 and max_nfev 60 — the settings ``calibrate_extrinsics``
                passes for its robust stage (reference core/calibrate_extrinsics.py:231-238): the same trajectory, evaluation for evaluation
cfg3_default    the same arrays, both solvers at the reference's default tolerances
cfg3_tight      the product at 1e-13 against scipy at 1e-15 with tight inner solves, + oracle_polish
cfg5_sample_*   bench.cfg5_sample_parity: the cfg5 recipe (128 cameras, free intrinsics, the reference's bounds) at 100k and 1M observations
cfg5_full_linear_algebra   FULL cfg5 (10M observations) through size-independent properties, the oracle's rows and sparse Jacobian taken half a million
                observations at a time: residual rows, cost, gradient J^T f and Jacobi scale of the device against the oracle's, and the device's damped
                step s put into the oracle's normal equations: ||(J^T J + lam D^2) s + g||_inf / ||g||_inf (the whole Schur route — records, pair
                products, dense solve, back-substitution — checked without a second solver; --skip-cfg5-full leaves it out)
(full cfg4 is compared in every bench.py run: ``parity`` in its JSON line)
"""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import bench
from caliscope_amd.build import source_digest
from caliscope_amd.least_squares import least_squares
from oracle.residuals import joint_jacobian, joint_residuals
from oracle.solver import optimize_scipy

TIGHT_GPU = dict(ftol=1e-13, xtol=1e-13, gtol=1e-13, max_nfev=20000)


def run(name, tol_scipy, tol_gpu, stored=None, polish=False):
    sc, par, x0, prob, cfg = bench.build_problem(name)
    fs = prob.f_scale
    ref_file = bench.stored_scipy_reference(stored, x0) if stored else None
    if ref_file is not None:
        ref_x, ref = ref_file["x"], ref_file
        source = f"tests/golden/scipy_refs/{stored}.npz"
    else:
        t0 = time.perf_counter()
        r = optimize_scipy(par, sc.camera_indices, sc.image_coords, sc.obj_indices, x0, loss=prob.loss, f_scale=fs, **tol_scipy)
        ref_x = r.x
        ref = {"nfev": int(r.nfev), "njev": int(r.njev), "status": int(r.status), "cost": float(r.cost), "seconds": round(time.perf_counter() - t0, 2)}
        source = "computed in this run"
    t0 = time.perf_counter()
    got = least_squares(None, x0, jac=None, bounds=par.bounds(), x_scale="jac", loss=prob.loss, f_scale=fs, method="trf",
                        args=(par, sc.camera_indices, sc.image_coords, sc.obj_indices), **tol_gpu)
    t_got = time.perf_counter() - t0
    fx = np.array([b.fx_initial for b in par.blocks])[sc.camera_indices]

    def rms(x):
        e = joint_residuals(x, par, sc.camera_indices, sc.image_coords, sc.obj_indices).reshape(-1, 2) * fx[:, None]
        return float(np.sqrt(np.mean(np.sum(e * e, axis=1))))

    detail = bench.solution_parity(par, got.x, ref_x, detail=True)
    cost_gpu = bench.oracle_cost(sc, par, got.x, prob.loss, fs)  # the oracle's cost of the product's answer
    out = {
        "workload": f"{name}: {len(par.blocks)} cams / {par.n_points} points / {prob.n_obs} obs, {prob.loss} loss",
        "settings": {"scipy": {k: v for k, v in tol_scipy.items()}, "gpu": tol_gpu}, "scipy_source": source,
        "scipy": {"nfev": int(ref["nfev"]), "njev": int(ref["njev"]), "status": int(ref["status"]), "cost": float(ref["cost"]), "rms_px": rms(ref_x),
                  "seconds": ref["seconds"]},
        "gpu": {"nfev": int(got.nfev), "njev": int(got.njev), "accepted_steps": int(got.njev) - 1, "rejected_trials": int(got.nfev) - int(got.njev),
                "status": int(got.status), "cost": float(got.cost), "cost_by_oracle": cost_gpu, "rms_px": rms(got.x), "seconds_end_to_end": round(t_got, 3)},
        "d_rms_px": rms(got.x) - rms(ref_x), "rel_cost": (cost_gpu - float(ref["cost"])) / float(ref["cost"]),
        "aligned_pos": detail["aligned_pos"], "aligned_ang_rad": detail["aligned_ang_rad"], "detail": detail,
        "within_north_star": bool(detail["aligned_pos"] <= 1e-6 and detail["aligned_ang_rad"] <= 1e-6 and abs(rms(got.x) - rms(ref_x)) <= 1e-4),
    }
    if detail["points_above_1e-6"] and detail["points_above_1e-6"] <= 50:
        out["weak_points"] = bench.weak_points(sc, par, got.x, ref_x, prob.loss, fs)
    if polish:
        out["oracle_polish"] = bench.oracle_polish(sc, par, got.x, prob.loss, fs, tight_inner=prob.n_obs <= 500_000)
    return out


def full_size_linear_algebra(name="cfg5", lam=1e-3, chunk=500_000):
    from caliscope_amd.hip_engine import HipEngine

    sc, par, x0, prob, cfg = bench.build_problem(name)
    assert prob.loss == "linear"  # (a robust loss would need scipy's row scaling in the oracle leg)
    t0 = time.perf_counter()
    with HipEngine(prob) as eng:
        r_dev, cost_dev = eng.residuals(x0)
        eng.begin(x0)
        eng.linearize()
        g_dev, d_dev = eng.get_vector(2), eng.get_vector(4)
        ok = eng.newton_step(lam).ok
        s_dev = eng.get_vector(3)
    t_dev = time.perf_counter() - t0
    n_obs, n = prob.n_obs, len(x0)
    g, d2, JtJs = np.zeros(n), np.zeros(n), np.zeros(n)
    cost, rows_diff = 0.0, 0.0
    t0 = time.perf_counter()
    for a in range(0, n_obs, chunk):
        sl = slice(a, min(a + chunk, n_obs))
        args = (par, sc.camera_indices[sl], sc.image_coords[sl], sc.obj_indices[sl])
        r = joint_residuals(x0, *args)
        J = joint_jacobian(x0, *args)
        rows_diff = max(rows_diff, float(np.abs(r - r_dev[2 * sl.start:2 * sl.stop]).max()))
        cost += 0.5 * float(r @ r)
        g += J.T @ r
        d2 += np.asarray(J.multiply(J).sum(axis=0)).ravel()
        JtJs += J.T @ (J @ s_dev)
    t_orc = time.perf_counter() - t0
    d = np.sqrt(d2)
    d[d == 0] = 1.0  # scipy's compute_jac_scale: an all-zero column keeps scale 1
    lin_res = JtJs + lam * d * d * s_dev + g
    return {
        "workload": f"{name}, FULL: {len(par.blocks)} cams / {par.n_points} points / {n_obs} obs, free intrinsics, x0, lam = {lam:g}",
        "residual_rows_max_abs_diff": rows_diff, "cost_rel_diff": abs(cost_dev - cost) / cost,
        "gradient_rel_inf": float(np.abs(g_dev - g).max() / np.abs(g).max()), "jacobi_scale_rel_max": float(np.abs(d_dev / d - 1.0).max()),
        "step_ok": bool(ok), "step_norm_inf": float(np.abs(s_dev).max()),
        "normal_equation_residual_of_the_device_step_rel_inf": float(np.abs(lin_res).max() / np.abs(g).max()),
        "seconds": {"device_incl_set_up": round(t_dev, 2), "oracle_chunks": round(t_orc, 1)},
    }


if __name__ == "__main__":
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    default = dict(ftol=1e-8, xtol=1e-8, gtol=1e-8, max_nfev=None)
    tight15 = dict(ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=None)
    tight_inner = dict(ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=3000, tr_options=dict(atol=1e-14, btol=1e-14))
    robust_stage = dict(ftol=1e-4, xtol=1e-8, gtol=1e-8, max_nfev=60)
    out = {
        "library_source_sha256": source_digest(),
        "host_cores": os.cpu_count(),
        "cfg2": run("cfg2", tight15, TIGHT_GPU, stored="cfg2_tight", polish=True),
        "cfg3_product": run("cfg3", robust_stage, robust_stage),
        "cfg3_default": run("cfg3", default, default, stored="cfg3_default"),
        # (1e-15 on the product's side as well: one of cfg3's 50 000 points has all its observations in Huber's linear region — nearly free along its
        # ray — and two answers 1e-13 of the cost apart still differ by 1.5e-6 there)
        "cfg3_tight": run("cfg3", tight_inner, dict(ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=20000), stored="cfg3_tight", polish=True),
    }
    t0 = time.perf_counter()
    out["cfg5_sample_100k"] = bench.cfg5_sample_parity(n_points=10_000)
    out["cfg5_sample_100k"]["wall_seconds"] = round(time.perf_counter() - t0, 1)
    if "--skip-cfg5-1M" not in sys.argv:  # SURVEY.md 8d's cfg5 sample: 128 cams / 100k points / 1M obs, free intrinsics + bounds
        t0 = time.perf_counter()
        out["cfg5_sample_1M"] = bench.cfg5_sample_parity(n_points=100_000)
        out["cfg5_sample_1M"]["wall_seconds"] = round(time.perf_counter() - t0, 1)
    if "--skip-cfg5-full" not in sys.argv:
        out["cfg5_full_linear_algebra"] = full_size_linear_algebra("cfg5")
    path = argv[0] if argv else "profiles/parity_r06.json"
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))
