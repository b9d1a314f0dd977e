"""Drop-in for the one solver call of the reference's hot path.

``CaptureVolume.optimize`` (reference ``core/capture_volume.py:387-411``) calls::

    result = least_squares(joint_residuals, x0,
        args=(parameterization, camera_indices, image_coords, image_to_world_indices,
              constraint_groups_a, constraint_groups_b, constraint_distances, constraint_weights),
        jac=joint_jacobian, verbose=verbose, x_scale="jac", loss=loss, f_scale=f_scale,
        ftol=ftol, max_nfev=max_nfev, method="trf", bounds=parameterization.bounds())

and consumes ``result.status``, ``.x``, ``.nfev`` and ``.cost`` (``:413-432``).  :func:`least_squares` here
accepts exactly that call and runs it on the MI355X engine: the problem is identified by ``args`` (the
callables ``fun`` / ``jac`` are the reference's ``joint_residuals`` / ``joint_jacobian``, whose arithmetic the
HIP kernels implement; they are never invoked).  INTEGRATION.md shows the one-line patch.

Semantics kept from scipy 1.15.3: cost definition, robust losses, ``x_scale='jac'``, termination codes,
``nfev`` accounting, ``max_nfev=None -> 100 n``, ``ValueError`` for bad options, infeasible ``x0`` or
non-finite initial residuals, constraint rows (the four trailing ``args``), finite bounds through scipy's
bounded variant (Coleman-Li scaling, reflective steps).  Difference: the regularised
Gauss-Newton step is exact (Schur complement) instead of LSMR at 1e-6.  The loop itself runs in the library
(``cba_solve``, behind the engine's ``solve()``); ``engine_factory`` is the test hook through which the CPU tests plug in the numpy
engine of ``oracle/engine.py`` (its ``solve()`` runs a Python restatement of the loop that only guards bounds by rejecting
infeasible trial points).
"""

from __future__ import annotations

import math
import time

import numpy as np

from caliscope_amd.bundle_parameterization import n_params_of
from caliscope_amd.engine import LOSS_CODES, BAProblem
from caliscope_amd.exceptions import BackendError
from caliscope_amd.engine import STATUS_REASONS

TERMINATION_MESSAGES = {
    -1: "Improper input parameters status returned from `leastsq`",
    0: "The maximum number of function evaluations is exceeded.",
    1: "`gtol` termination condition is satisfied.",
    2: "`ftol` termination condition is satisfied.",
    3: "`xtol` termination condition is satisfied.",
    4: "Both `ftol` and `xtol` termination conditions are satisfied.",
}


class OptimizeResult(dict):
    """Attribute-style result, same fields the reference reads from scipy's OptimizeResult."""

    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _make_strictly_feasible(x, lb, ub, rstep=1e-10):
    """scipy's ``make_strictly_feasible`` (common.py:437-463): nudge points sitting on a bound inside."""
    x = x.copy()
    # only entries with a finite bound can move (here: the camera block — a few hundred of millions of parameters; the full-vector form cost more
    # than a small solve)
    at = np.flatnonzero(np.isfinite(lb) | np.isfinite(ub))
    if at.size == 0:
        return x
    xs, lbs, ubs = x[at], lb[at], ub[at]
    lower_thr = rstep * np.maximum(1, np.abs(lbs))
    upper_thr = rstep * np.maximum(1, np.abs(ubs))
    with np.errstate(invalid="ignore"):
        lo = np.isfinite(lbs) & (xs - lbs <= np.minimum(ubs - xs, lower_thr))
        hi = np.isfinite(ubs) & (ubs - xs <= np.minimum(xs - lbs, upper_thr))
    xs[lo] = lbs[lo] + lower_thr[lo]
    xs[hi] = ubs[hi] - upper_thr[hi]
    bad = (xs < lbs) | (xs > ubs)
    xs[bad] = 0.5 * (lbs[bad] + ubs[bad])
    x[at] = xs
    return x


def least_squares(
    fun,
    x0,
    jac="2-point",
    bounds=(-np.inf, np.inf),
    method="trf",
    ftol=1e-8,
    xtol=1e-8,
    gtol=1e-8,
    x_scale=1.0,
    loss="linear",
    f_scale=1.0,
    diff_step=None,
    tr_solver=None,
    tr_options=None,
    jac_sparsity=None,
    max_nfev=None,
    verbose=0,
    args=(),
    kwargs=None,
    *,
    engine_factory=None,
    devices=None,
):
    """Solve the bundle-adjustment problem described by ``args`` on the MI355X.

    ``engine_factory(problem) -> BAEngine`` is a test hook (defaults to the HIP engine).  ``devices`` (or the
    environment variable ``CALISCOPE_HIP_DEVICES=0,1,...``) names the GPUs of this node the solve is sharded over — by
    world point, one host thread per device inside this call (:func:`caliscope_amd.distributed.solve_multi_device`); the
    reference's call site needs no launcher and no change for it.
    """
    if method != "trf":
        raise ValueError("the MI355X backend implements method='trf' only (what the reference uses)")
    if not (isinstance(x_scale, str) and x_scale == "jac"):
        raise ValueError("the MI355X backend implements x_scale='jac' only (what the reference uses)")
    if loss not in LOSS_CODES:
        raise ValueError(f"`loss` must be one of {sorted(LOSS_CODES)} (callables are not supported)")
    if verbose not in (0, 1, 2):
        raise ValueError("`verbose` must be in [0, 1, 2].")
    if max_nfev is not None and max_nfev <= 0:
        raise ValueError("`max_nfev` must be None or positive integer.")
    if len(args) < 4:
        raise ValueError("args must be (parameterization, camera_indices, image_coords, obj_indices, ...) as in CaptureVolume.optimize")
    parameterization, camera_indices, image_coords, obj_indices = args[:4]
    con = tuple(args[4:8]) + (None,) * (4 - len(args[4:8]))
    if any(a is not None for a in con) and any(a is None for a in con):
        raise ValueError("constraint args (groups_a, groups_b, distances, weights) must be all given or all None")
    if con[0] is not None and len(con[2]) == 0:
        con = (None, None, None, None)
    x0 = np.atleast_1d(np.asarray(x0, dtype=np.float64))
    if x0.ndim != 1:
        raise ValueError("`x0` must have at most 1 dimension.")
    n_expected = n_params_of(parameterization)
    if x0.size != n_expected:
        raise ValueError(f"x0 has {x0.size} entries, the parameterization expects {n_expected}")
    for name, tol in (("ftol", ftol), ("xtol", xtol), ("gtol", gtol)):
        if tol is None:
            raise ValueError(f"`{name}` must be a number for the MI355X backend")
    eps = np.finfo(np.float64).eps
    if ftol < eps and xtol < eps and gtol < eps:
        raise ValueError(f"At least one of the tolerances must be higher than machine epsilon ({eps:.2e}).")

    lb, ub = (np.broadcast_to(np.asarray(b, dtype=np.float64), x0.shape).copy() for b in bounds)
    if np.any(lb >= ub):
        raise ValueError("Each lower bound must be strictly less than each upper bound.")
    if np.any((x0 < lb) | (x0 > ub)):
        raise ValueError("Initial guess is outside of provided bounds")
    bounded = bool(np.any(np.isfinite(lb)) or np.any(np.isfinite(ub)))
    if bounded:
        ncp = parameterization.n_camera_params
        if np.any(np.isfinite(lb[ncp:])) or np.any(np.isfinite(ub[ncp:])):
            raise BackendError("bounds on world points are not supported (the reference never sets them)")
        x0 = _make_strictly_feasible(x0, lb, ub)

    problem = BAProblem(parameterization, camera_indices, image_coords, obj_indices, loss=loss, f_scale=float(f_scale),
                        constraint_groups_a=con[0], constraint_groups_b=con[1], constraint_distances=con[2], constraint_weights=con[3])
    if engine_factory is None:
        from caliscope_amd.distributed import devices_from_env, solve_multi_device
        if devices is None:
            devices = devices_from_env()
        if devices is not None and len(devices) > 1:
            ncp = parameterization.n_camera_params
            res = solve_multi_device(problem, x0, devices, ftol=ftol, xtol=xtol, gtol=gtol, max_nfev=max_nfev, verbose=verbose,
                                     cam_bounds=(lb[:ncp], ub[:ncp]) if bounded else None)
            if res.status == -1:
                raise ValueError("Residuals are not finite in the initial point.")
            return _result_of(res, verbose)
        # the default path: a handle built for the same observations by the previous call is reused (engine_cache)
        from caliscope_amd import engine_cache

        cache_key = None
        t_setup = time.perf_counter()
        engine, cache_key = engine_cache.checkout(problem, device_id=int(devices[0]) if devices else -1)
        t_setup = time.perf_counter() - t_setup  # fingerprint + (on a miss) sort, Schur plan, upload: reported, not hidden
        cached = True
    else:
        t_setup = time.perf_counter()
        engine = engine_factory(problem)
        t_setup = time.perf_counter() - t_setup
        cached = False
    solved = False
    t_solve = time.perf_counter()
    try:
        # the whole loop behind the engine's solve(): cba_solve in the library for the device engine (the one driver of the package)
        ncp = parameterization.n_camera_params
        res = engine.solve(x0, ftol=ftol, xtol=xtol, gtol=gtol, max_nfev=max_nfev, verbose=verbose,
                           lb=lb[:ncp] if bounded else None, ub=ub[:ncp] if bounded else None)
        if res.status == -1:
            raise ValueError("Residuals are not finite in the initial point.")
        solved = True
        t_solve = time.perf_counter() - t_solve
    finally:
        if cached and solved:
            engine_cache.checkin(cache_key, engine)
        else:  # a handle that raised is not kept
            close = getattr(engine, "close", None)
            if close is not None:
                close()

    out = _result_of(res, verbose)
    out["setup_seconds"], out["solve_seconds"] = t_setup, t_solve  # extra fields next to scipy's (the reference reads status, x, nfev, cost)
    return out


def _result_of(res, verbose):
    out = OptimizeResult(
        x=res.x, cost=res.cost, optimality=res.optimality, nfev=res.nfev, njev=res.njev, status=res.status,
        message=TERMINATION_MESSAGES.get(res.status, STATUS_REASONS.get(res.status, "")), success=res.status > 0,
        active_mask=np.zeros_like(res.x), n_iterations=res.n_iterations, trace=res.trace,
    )
    if verbose >= 1:
        print(out.message)
        print(f"Function evaluations {out.nfev}, initial cost {res.trace[0]['cost'] if res.trace else out.cost:.4e}, "
              f"final cost {out.cost:.4e}, first-order optimality {out.optimality:.2e}.")
    if not math.isfinite(out.cost):
        out.status, out.success = -1, False
    return out
