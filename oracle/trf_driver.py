"""TEST INFRASTRUCTURE (round 5: moved out of the package, which has ONE driver: ``csrc/cba_solve.cpp``).  A Python restatement of that
driver's unbounded loop on the engine primitives: what ``OracleEngine.solve()`` runs for the CPU tests, and what the GPU tests and the smoke
check run on the device primitives to compare with ``cba_solve`` evaluation by evaluation (tests/test_gpu_parity.py).  No bounded
(Coleman-Li) variant: with bounds it only rejects infeasible trial points.

The reference hands the problem to ``scipy.optimize.least_squares(method="trf",
x_scale="jac", jac=<sparse>)`` (``core/capture_volume.py:387-411``), i.e. scipy's Trust-Region-
Reflective loop with a regularised Gauss-Newton step from LSMR and a 2-D subspace trust-region
solve (scipy 1.15.3 ``optimize/_lsq/trf.py:401-560``).  This module keeps that
outer loop — same regularisation rule, same 2-D subspace ``span{g_h, gn_h}``, same radius update,
same termination codes and ``nfev`` accounting — so that ``OptimizationStatus`` and iteration
counts stay comparable, but replaces the inner solver: the regularised step is the exact solution
of the Marquardt-damped normal equations

    (J^T J + lam * D^2) s = -g ,      lam = reg_term,  D = Jacobi column norms (x_scale='jac'),

computed on the device by Schur complement on the reduced camera system (one Levenberg-Marquardt
step), instead of LSMR iterating to ``atol=btol=1e-6``.  Everything O(n) or O(N_obs) happens in the
engine (:mod:`caliscope_amd.engine`); only a dozen scalars per iteration reach this code.

The quadratic model restricted to the subspace needs ``S^T J_h^T J_h S``.  scipy multiplies
``J_h @ S`` explicitly; here it follows from the step equation itself: with ``H = J_h^T J_h`` and
``(H + lam I) p = -g_h``,

    g_h^T H p = -||g_h||^2 - lam <g_h, p>,      p^T H p = -<g_h, p> - lam ||p||^2,

so besides ``||J_h g_h||^2`` (already needed for ``reg_term``) no further pass over the
observations is required.
"""

from __future__ import annotations

import math
import numpy as np

from caliscope_amd.engine import STATUS_REASONS, BAEngine, TrfResult  # noqa: F401 (re-exported for the tests)

# ||w||^2 / ||p||^2 below which the subspace model is built from explicit J.v products
SUBSPACE_EXPLICIT_BELOW = 1e-6

DAMPING_FLOOR = 1e-13  # csrc/trf_math.h


def _min_quadratic_on_segment(a: float, b: float, hi: float) -> float:
    """min over t in [0, hi] of a t^2 + b t."""
    best = min(0.0, hi * (a * hi + b))
    if a != 0.0:
        t = -0.5 * b / a
        if 0.0 < t < hi:
            best = min(best, t * (a * t + b))
    return best


def solve_subspace_2d(B: np.ndarray, g: np.ndarray, radius: float) -> np.ndarray:
    """argmin 0.5 p^T B p + g^T p  s.t. ||p|| <= radius, in two dimensions.

    Interior Newton point when B is positive definite and the point is inside; otherwise the
    boundary is parameterised by ``p = radius * (2t, 1 - t^2) / (1 + t^2)`` and the stationarity
    condition becomes a quartic in ``t`` (the formulation scipy's ``solve_trust_region_2d`` uses,
    ``common.py:171-219``), whose real roots are compared by model value.
    """
    b00, b01, b11 = float(B[0, 0]), float(B[0, 1]), float(B[1, 1])
    if b00 > 0.0:
        schur = b11 - b01 * b01 / b00
        if schur > 0.0:
            det = b00 * schur
            p = np.array([-(b11 * g[0] - b01 * g[1]) / det, -(b00 * g[1] - b01 * g[0]) / det])
            if p @ p <= radius * radius:
                return p
    r2 = radius * radius
    a, b, c = b00 * r2, b01 * r2, b11 * r2
    d, f = g[0] * radius, g[1] * radius
    roots = np.roots(np.array([-b + d, 2.0 * (a - c + f), 6.0 * b, 2.0 * (-a + c + f), -b - d]))
    t = np.real(roots[np.isreal(roots)])
    if t.size == 0:  # numerically degenerate quartic: fall back to the steepest-descent boundary point
        n = math.hypot(g[0], g[1])
        return -radius * np.asarray(g) / n if n > 0 else np.zeros(2)
    cand = radius * np.vstack((2.0 * t / (1.0 + t * t), (1.0 - t * t) / (1.0 + t * t)))
    # t -> infinity (p = (0, -radius)) is not representable; add it explicitly
    cand = np.hstack([cand, np.array([[0.0], [-radius]])])
    val = 0.5 * np.sum(cand * (B @ cand), axis=0) + g @ cand
    return cand[:, int(np.argmin(val))]


def _update_radius(radius, actual, predicted, step_norm, bound_hit):
    if predicted > 0:
        ratio = actual / predicted
    elif predicted == actual == 0:
        ratio = 1.0
    else:
        ratio = 0.0
    if ratio < 0.25:
        radius = 0.25 * step_norm
    elif ratio > 0.75 and bound_hit:
        radius *= 2.0
    return radius, ratio


def _termination(dF, F, dx_norm, x_norm, ratio, ftol, xtol):
    f_ok = dF < ftol * F and ratio > 0.25
    x_ok = dx_norm < xtol * (xtol + x_norm)
    if f_ok and x_ok:
        return 4
    if f_ok:
        return 2
    if x_ok:
        return 3
    return None


def trf_solve(
    engine: BAEngine,
    x0: np.ndarray,
    *,
    ftol: float = 1e-8,
    xtol: float = 1e-8,
    gtol: float = 1e-8,
    max_nfev: int | None = None,
    verbose: int = 0,
    max_damping_retries: int = 12,
    feasible=None,
    fetch_x: bool = True,
) -> TrfResult:
    """Run the trust-region loop on ``engine`` starting from ``x0``.

    ``feasible(camera_params) -> bool`` (optional) is evaluated on the camera part of every trial point;
    an infeasible trial is treated like a non-finite one (the radius shrinks), which keeps the iterates
    strictly inside the intrinsic bounds of ``BundleParameterization.bounds()``.
    """
    if x0 is None:  # restart from the x0 the engine already holds (bench loops; no host transfer)
        if max_nfev is None:
            max_nfev = int(engine.n_params) * 100
        cost = engine.restart()
    else:
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        if max_nfev is None:
            max_nfev = int(x0.size) * 100
        cost = engine.begin(x0)
    if not math.isfinite(cost):
        raise ValueError("Residuals are not finite in the initial point.")
    nfev = 1
    lin = engine.linearize()
    njev = 1

    radius = lin.x_scaled_norm if lin.x_scaled_norm > 0 else 1.0
    status = None
    iteration = 0
    trace = []
    g_norm = lin.g_norm_inf
    if verbose == 2:
        print(f"{'Iteration':^15}{'Total nfev':^15}{'Cost':^15}{'Cost reduction':^15}{'Step norm':^15}{'Optimality':^15}")
    step_norm = None
    actual = None

    while True:
        g_norm = lin.g_norm_inf
        if g_norm < gtol:
            status = 1
        if verbose == 2:
            ar = "" if actual is None else f"{actual:^15.2e}"
            sn = "" if step_norm is None else f"{step_norm:^15.2e}"
            print(f"{iteration:^15}{nfev:^15}{cost:^15.4e}{ar:^15}{sn:^15}{g_norm:^15.2e}")
        if status is not None or nfev >= max_nfev:
            break

        gh_sq = lin.gh_sq
        gh_norm = math.sqrt(gh_sq)
        # Regularisation = the model decrease along -g_h inside the region, per unit radius^2.
        # (floored where the damped system can still be factorised in double precision: csrc/trf_math.h, trf::damping)
        lam = max(-_min_quadratic_on_segment(0.5 * lin.jg_sq, -gh_sq, radius / gh_norm) / (radius * radius), DAMPING_FLOOR)

        st = engine.newton_step(lam)
        retries = 0
        while not st.ok:
            # The damped system is positive definite in exact arithmetic; rounding on a gauge-singular
            # problem can still break the factorisation.  More damping is the Levenberg-Marquardt remedy.
            retries += 1
            if retries > max_damping_retries:
                raise FloatingPointError("normal equations could not be factorised even with heavy damping")
            lam = max(lam * 10.0, 1e-14 * 10.0**retries)
            st = engine.newton_step(lam)

        # Orthonormal basis of span{g_h, p}:  q1 = g_h/||g_h||,  q2 = w/||w||,  w = p - c g_h.
        c = st.gh_dot_p / gh_sq
        w_sq = st.w_sq
        two_d = w_sq > 0.0 and w_sq > 1e-30 * st.p_sq
        B = np.zeros((2, 2))
        w_norm = math.sqrt(w_sq) if two_d else 1.0
        if not two_d:
            B[0, 0] = lin.jg_sq / gh_sq
            B[1, 1] = 1.0
        elif w_sq > SUBSPACE_EXPLICIT_BELOW * st.p_sq:
            # model from the step equation (no extra pass over the observations)
            H_gg = lin.jg_sq
            H_gp = -gh_sq - lam * st.gh_dot_p
            H_pp = -st.gh_dot_p - lam * st.p_sq
            B[0, 0] = H_gg / gh_sq
            B[0, 1] = B[1, 0] = (H_gp - c * H_gg) / (gh_norm * w_norm)
            B[1, 1] = (H_pp - 2.0 * c * H_gp + c * c * H_gg) / w_sq
        else:
            # p is (nearly) collinear with g_h — heavy damping.  The identities above cancel
            # catastrophically, so form J_h q1, J_h q2 explicitly as scipy does (one extra pass).
            b00, b01, b11 = engine.subspace_gram(1.0 / gh_norm, 0.0, -c / w_norm, 1.0 / w_norm)
            B[0, 0], B[0, 1], B[1, 0], B[1, 1] = b00, b01, b01, b11
        g_S = np.array([gh_norm, 0.0])

        actual = -1.0
        cost_new = cost
        while actual <= 0 and nfev < max_nfev:
            p_S = solve_subspace_2d(B, g_S, radius)
            if not two_d:
                p_S[1] = 0.0
            predicted = -(0.5 * float(p_S @ B @ p_S) + float(g_S @ p_S))
            # step_h = p_S[0] q1 + p_S[1] q2 = alpha g_h + beta p ;  step = d * step_h = alpha d^2 g + beta s
            beta = p_S[1] / w_norm if two_d else 0.0
            alpha = p_S[0] / gh_norm - beta * c
            tr = engine.trial(alpha, beta)
            nfev += 1
            step_h_norm = math.hypot(p_S[0], p_S[1])
            if not tr.finite or (feasible is not None and not feasible(engine.camera_params(1))):
                radius = 0.25 * step_h_norm
                continue
            cost_new = tr.cost
            actual = cost - cost_new
            radius_new, ratio = _update_radius(radius, actual, predicted, step_h_norm, step_h_norm > 0.95 * radius)
            step_norm = tr.step_norm
            status = _termination(actual, cost, step_norm, lin.x_norm, ratio, ftol, xtol)
            if status is not None:
                break
            radius = radius_new

        trace.append({"cost": cost, "g_norm": g_norm, "radius": radius, "lam": lam, "nfev": nfev})
        if actual > 0:
            engine.accept()
            cost = cost_new
            lin = engine.linearize()
            njev += 1
        else:
            step_norm = 0.0
            actual = 0.0
        iteration += 1

    if status is None:
        status = 0
    return TrfResult(
        x=engine.current_x() if fetch_x else None, cost=float(cost), optimality=float(g_norm), nfev=nfev, njev=njev, status=status,
        n_iterations=iteration, trace=trace,
    )
