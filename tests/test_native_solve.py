"""csrc/cba_solve.cpp (the native trust-region driver) on the CPU: compiled by g++ against a dense test double of the
device primitives (tests/native/dense_engine.cpp) and compared, evaluation by evaluation, with the Python driver
(oracle/trf_driver.py) on the numpy engine and, at convergence, with scipy."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

from caliscope_amd import _lib
from oracle.trf_driver import solve_subspace_2d, trf_solve
from oracle.engine import OracleEngine
from oracle.residuals import joint_jacobian, joint_residuals
from oracle.solver import optimize_scipy
from tests.helpers import aligned_difference, small_problem

ROOT = Path(__file__).resolve().parent.parent
FUN = C.CFUNCTYPE(None, _lib.c_double_p, _lib.c_double_p)


@pytest.fixture(scope="module")
def native(tmp_path_factory):
    out = tmp_path_factory.mktemp("ns") / "libnative_solve.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", str(ROOT / "caliscope_amd" / "csrc" / "cba_solve.cpp"),
                    str(ROOT / "tests" / "native" / "dense_engine.cpp"), "-o", str(out)], check=True)
    lib = C.CDLL(str(out))
    lib.de_create.restype = C.c_void_p
    lib.de_create.argtypes = [C.c_int, C.c_int, C.c_int, FUN, FUN]
    lib.de_destroy.argtypes = [C.c_void_p]
    lib.cba_solve.restype = C.c_int
    lib.cba_solve.argtypes = [C.c_void_p, _lib.c_double_p, C.POINTER(_lib.SolveOptions), _lib.c_double_p, C.POINTER(_lib.Result)]
    lib.cba_last_error.restype = C.c_char_p
    return lib


def _solve(lib, fun, jac, x0, m, ncp=0, lb=None, ub=None, **kw):
    n = x0.size
    evals = []

    def f_cb(xp, rp):
        x = np.ctypeslib.as_array(xp, (n,))
        evals.append(x.copy())
        np.ctypeslib.as_array(rp, (m,))[:] = fun(x)

    def j_cb(xp, jp):
        np.ctypeslib.as_array(jp, (m * n,))[:] = np.asarray(jac(np.ctypeslib.as_array(xp, (n,)))).reshape(-1)

    fc, jc = FUN(f_cb), FUN(j_cb)
    h = lib.de_create(m, n, ncp, fc, jc)
    opt = _lib.SolveOptions(ftol=kw.get("ftol", 1e-8), xtol=kw.get("xtol", 1e-8), gtol=kw.get("gtol", 1e-8), max_nfev=kw.get("max_nfev", 0),
                            lb=None if lb is None else lb.ctypes.data_as(_lib.c_double_p), ub=None if ub is None else ub.ctypes.data_as(_lib.c_double_p),
                            verbose=0, max_damping_retries=0)
    res, x = _lib.Result(), np.empty(n)
    rc = lib.cba_solve(h, x0.ctypes.data_as(_lib.c_double_p), C.byref(opt), x.ctypes.data_as(_lib.c_double_p), C.byref(res))
    counters = (C.c_long * 2)()
    lib.de_counters(C.c_void_p(h), counters)
    res.damped_steps, res.failed_factorisations = int(counters[0]), int(counters[1])
    lib.de_destroy(h)
    assert rc == 0, lib.cba_last_error()
    return res, x, evals


@pytest.mark.parametrize("refine", [False, True])
def test_native_driver_follows_the_python_driver_on_a_small_bundle(native, refine):
    sc, par, x0 = small_problem(n_cams=4, n_points=40, k=4, refine=refine)
    args = (par, sc.camera_indices, sc.image_coords, sc.obj_indices)
    m = 2 * len(sc.camera_indices)
    lb, ub = par.bounds()
    ncp = par.n_camera_params
    kw = dict(lb=np.ascontiguousarray(lb[:ncp]), ub=np.ascontiguousarray(ub[:ncp])) if refine else {}
    res, x, _ = _solve(native, lambda x: joint_residuals(x, *args), lambda x: joint_jacobian(x, *args).toarray(), x0, m, ncp=ncp, **kw)
    if not refine:  # no finite bound: the loop of trf.py, evaluation by evaluation
        ref = trf_solve(OracleEngine(*args), x0)
        assert (res.status, res.nfev, res.njev, res.n_iterations) == (ref.status, ref.nfev, ref.njev, ref.n_iterations)
        assert abs(res.cost - ref.cost) <= 1e-9 * ref.cost and abs(res.optimality - ref.optimality) <= 1e-9  # a converged gradient is rounding noise
        pos, ang, _ = aligned_difference(par, x, ref.x)  # raw x differs along the gauge directions (damping ~1e-15 there)
        assert pos < 1e-6 and ang < 1e-6
    sci = optimize_scipy(*args, x0)
    assert res.status > 0 and abs(res.cost - sci.cost) <= 1e-6 * sci.cost
    if refine:
        # this scene drives k2 of one camera onto its lower bound (-2): the Coleman-Li scaling has to carry the iterate
        # there (a driver that merely rejects infeasible trials stalls at ~700x this cost)
        k2 = x[:ncp].reshape(-1, 9)[:, 8]
        assert np.all(x[:ncp] > lb[:ncp]) and np.all(x[:ncp] < ub[:ncp]) and k2.min() < -1.999
        # the fully converged points agree (k2 of the other cameras is weakly determined by 40 points: compare both
        # sides at tight tolerances, scipy with exact SVD steps)
        tight = dict(ftol=1e-15, xtol=1e-15, gtol=1e-11)
        res_t, x_t, _ = _solve(native, lambda x: joint_residuals(x, *args), lambda x: joint_jacobian(x, *args).toarray(), x0, m, ncp=ncp,
                               max_nfev=100, **tight, **kw)
        sci_t = optimize_scipy(*args, x0, tr_solver="exact", max_nfev=100, **tight)
        assert abs(res_t.cost - sci_t.cost) <= 1e-10 * sci_t.cost
        assert np.abs(x_t[:ncp].reshape(-1, 9)[:, 6:] - sci_t.x[:ncp].reshape(-1, 9)[:, 6:]).max() < 1e-5
    # max_nfev and the restart path
    capped, _, evals = _solve(native, lambda x: joint_residuals(x, *args), lambda x: joint_jacobian(x, *args).toarray(), x0, m, ncp=ncp,
                              max_nfev=3, ftol=1e-15, xtol=1e-15, gtol=1e-15, **kw)
    assert capped.status == 0 and capped.nfev == 3 and len(evals) == 3


def test_native_driver_on_classic_problems(native):
    """Rosenbrock (narrow valley: rejected steps, radius shrink) and an exponential fit; status/nfev vs the Python
    driver on the same dense engine semantics, solution vs scipy."""
    from scipy.optimize import least_squares

    rosen = (lambda x: np.array([10.0 * (x[1] - x[0] ** 2), 1.0 - x[0]]), lambda x: np.array([[-20.0 * x[0], 10.0], [-1.0, 0.0]]))
    res, x, evals = _solve(native, *rosen, np.array([-1.2, 1.0]), 2)
    # the fused iteration evaluates its first trial before the host sees the gradient: when that iteration ends on gtol the
    # trial is discarded and — like scipy, which never makes it — not counted
    assert res.status > 0 and np.allclose(x, [1.0, 1.0], atol=1e-6) and len(evals) - res.nfev in (0, 1)
    t = np.linspace(0, 4, 30)
    y = 2.5 * np.exp(-1.3 * t) + 0.5 + 0.01 * np.cos(37 * t)
    fit = (lambda p: p[0] * np.exp(p[1] * t) + p[2] - y, lambda p: np.stack([np.exp(p[1] * t), p[0] * t * np.exp(p[1] * t), np.ones_like(t)], axis=1))
    res, x, _ = _solve(native, *fit, np.array([1.0, -0.5, 0.0]), t.size)
    sci = least_squares(fit[0], np.array([1.0, -0.5, 0.0]), jac=fit[1], method="trf", x_scale="jac")
    assert res.status > 0 and np.allclose(x, sci.x, rtol=1e-5, atol=1e-7) and abs(res.cost - sci.cost) < 1e-9 * max(sci.cost, 1e-12)
    # box constraints active at the solution: scipy's bounded TRF vs the native one (same unique minimiser)
    for lo, hi in ((np.array([1.5, -np.inf, -np.inf]), np.array([2.2, -1.5, np.inf])), (np.array([-np.inf, -1.0, 0.6]), np.array([np.inf, 0.0, 5.0]))):
        p0 = np.array([2.0, -2.0, 1.0]) if np.isfinite(hi[1]) and hi[1] < -1 else np.array([1.0, -0.5, 1.0])
        sci = least_squares(fit[0], p0, jac=fit[1], method="trf", x_scale="jac", bounds=(lo, hi))
        res, x, evals = _solve(native, *fit, p0, t.size, ncp=3, lb=lo, ub=hi)
        assert res.status > 0 and abs(res.cost - sci.cost) < 1e-6 * sci.cost and np.allclose(x, sci.x, rtol=1e-4, atol=1e-5)
        assert all(np.all(e > lo) and np.all(e < hi) for e in evals)  # every evaluation strictly inside
        assert sci.active_mask.any() or np.abs(x - np.clip(x, lo + 1e-3, hi - 1e-3)).max() > 0  # a bound is (nearly) active
    # non-finite start: scipy raises, the C driver reports status -1 without iterating
    res, _, evals = _solve(native, lambda p: np.full(2, np.nan), rosen[1], np.array([0.0, 0.0]), 2)
    assert res.status == -1 and len(evals) == 1


def test_subspace_solver_and_root_finder(native):
    """csrc/trf_math.h (shared by the host driver and the device-side step): the 2-D trust-region solve against brute
    force over the disc boundary / the Python solver, the real-root finder against numpy.roots."""
    native.de_subspace.argtypes = [C.c_double] * 6 + [_lib.c_double_p]
    native.de_real_roots.argtypes = [_lib.c_double_p, C.c_int, _lib.c_double_p]
    native.de_real_roots.restype = C.c_int
    rng = np.random.default_rng(5)
    th = np.linspace(0, 2 * np.pi, 20001)
    model = lambda B, g, p: 0.5 * p @ B @ p + g @ p
    for k in range(200):
        A = rng.normal(size=(2, 2))
        B = A @ A.T if k % 3 == 0 else A + A.T  # positive definite or indefinite
        g = rng.normal(size=2) * (10.0 ** rng.integers(-3, 3))
        r = float(rng.uniform(0.05, 3.0))
        out = np.zeros(2)
        native.de_subspace(B[0, 0], B[0, 1], B[1, 1], g[0], g[1], r, out.ctypes.data_as(_lib.c_double_p))
        assert out @ out <= r * r * (1 + 1e-12)
        ring = r * np.stack([np.cos(th), np.sin(th)])
        best = min((0.5 * np.sum(ring * (B @ ring), axis=0) + g @ ring).min(), model(B, g, solve_subspace_2d(B, g, r)))
        assert model(B, g, out) <= best + 1e-7 * (1 + abs(best)), (k, out)
    for k in range(300):
        deg = int(rng.integers(1, 5))
        if k % 4 == 0:  # built from known real roots, some repeated / clustered
            roots = rng.normal(size=deg) * (10.0 ** rng.integers(-2, 3))
            if deg > 1 and k % 8 == 0:
                roots[1] = roots[0] * (1 + 1e-5)
            coef = np.poly(roots) * rng.uniform(0.5, 2)
        else:
            coef = rng.normal(size=deg + 1)
        coef = np.concatenate([np.zeros(int(rng.integers(0, 2))), coef])  # leading zeros are stripped like numpy.roots
        out = np.zeros(4)
        n = native.de_real_roots(np.ascontiguousarray(coef).ctypes.data_as(_lib.c_double_p), len(coef), out.ctypes.data_as(_lib.c_double_p))
        ref = np.roots(coef)
        real = np.sort(ref[np.abs(ref.imag) < 1e-9 * (1 + np.abs(ref.real))].real)
        scale = np.abs(np.trim_zeros(coef, "f")).max()
        for t in out[:n]:  # every reported value is a root
            assert abs(np.polyval(coef, t)) <= 1e-9 * scale * max(1.0, abs(t)) ** (len(np.trim_zeros(coef, "f")) - 1), (coef, t)
        for t in real:  # every well-separated real root is reported
            if np.min(np.abs(np.delete(ref, np.argmin(np.abs(ref - t))) - t), initial=np.inf) > 1e-3 * (1 + abs(t)):
                assert n and np.min(np.abs(out[:n] - t)) <= 1e-7 * (1 + abs(t)), (coef, t, out[:n])


def test_boundary_subspace_solution_matches_the_python_solver(native):
    """The quartic root finder behind solve_subspace_2d: drive it through indefinite / boundary cases by a one-iteration
    solve is indirect, so compare the Python solver against brute force here and the native one through trajectories
    above; this test pins the Python reference used by both."""
    rng = np.random.default_rng(5)
    th = np.linspace(0, 2 * np.pi, 20001)
    for _ in range(30):
        A = rng.normal(size=(2, 2))
        B = A + A.T
        g = rng.normal(size=2)
        r = float(rng.uniform(0.1, 3.0))
        p = solve_subspace_2d(B, g, r)
        ring = r * np.stack([np.cos(th), np.sin(th)])
        best = (0.5 * np.sum(ring * (B @ ring), axis=0) + g @ ring).min()
        assert 0.5 * p @ B @ p + g @ p <= best + 1e-6 * (1 + abs(best))


def test_bounded_driver_uses_all_three_candidates_of_select_step(native):
    """Random box-constrained curve fits started near a corner: over the batch the trial steps that leave the box must have
    been replaced by each of select_step's candidates (truncated, reflected, scaled anti-gradient) at least once, and every
    solve must land on scipy's bounded minimiser."""
    from scipy.optimize import least_squares

    rng = np.random.default_rng(11)
    t = np.linspace(0, 3, 40)
    used = np.zeros(3, dtype=int)
    for k in range(40):
        true = np.array([rng.uniform(1, 3), rng.uniform(-2, -0.3), rng.uniform(0, 1), rng.uniform(-1, 1)])
        y = true[0] * np.exp(true[1] * t) + true[2] + true[3] * t + 0.01 * rng.normal(size=t.size)
        fun = lambda p: p[0] * np.exp(p[1] * t) + p[2] + p[3] * t - y
        jac = lambda p: np.stack([np.exp(p[1] * t), p[0] * t * np.exp(p[1] * t), np.ones_like(t), t], axis=1)
        lo = true - rng.uniform(0.05, 1.0, 4) * np.array([1, 1, 1, 1.0])
        hi = true + rng.uniform(0.05, 1.0, 4)
        cut = rng.integers(0, 4)  # one bound cuts the unconstrained minimiser off
        if rng.random() < 0.5:
            hi[cut] = true[cut] - 0.1 * abs(true[cut]) - 0.05
            lo[cut] = hi[cut] - 1.0
        else:
            lo[cut] = true[cut] + 0.1 * abs(true[cut]) + 0.05
            hi[cut] = lo[cut] + 1.0
        p0 = lo + rng.uniform(0.02, 0.98, 4) * (hi - lo)
        sci = least_squares(fun, p0, jac=jac, method="trf", x_scale="jac", bounds=(lo, hi), ftol=1e-12, xtol=1e-12, gtol=1e-12)
        res, x, evals = _solve(native, fun, jac, p0, t.size, ncp=4, lb=lo, ub=hi, ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=400)
        assert all(np.all(e > lo) and np.all(e < hi) for e in evals)
        assert res.status > 0 and abs(res.cost - sci.cost) <= 1e-6 * max(sci.cost, 1e-12), (k, res.cost, sci.cost)
        assert np.allclose(x, sci.x, rtol=1e-4, atol=1e-5), (k, x, sci.x)
        used += np.array([res.reserved & 1023, (res.reserved >> 10) & 1023, (res.reserved >> 20) & 1023])
    assert np.all(used > 0), used


def test_damping_rule_and_its_floor(native):
    """trf::damping (csrc/trf_math.h): scipy's regularisation rule (trf.py:477-483) above the floor, the floor below — and a gauge-free
    bundle solved to a vanishing gradient forms every damped step once (without the floor the factorisation of the late iterations fails:
    their damping, ~|g|^2, is lost in the rounding of the singular reduced system)."""
    from oracle.trf_driver import DAMPING_FLOOR, _min_quadratic_on_segment

    native.de_damping.restype = C.c_double
    native.de_damping.argtypes = [C.c_double] * 3
    rng = np.random.default_rng(7)
    for _ in range(50):
        gh_sq, radius = 10.0 ** rng.uniform(-30, 2), 10.0 ** rng.uniform(-3, 4)
        H_gg = gh_sq * 10.0 ** rng.uniform(-2, 2)
        rule = -_min_quadratic_on_segment(0.5 * H_gg, -gh_sq, radius / np.sqrt(gh_sq)) / radius ** 2
        got = native.de_damping(H_gg, gh_sq, radius)
        assert got == max(rule, DAMPING_FLOOR) or abs(got - rule) <= 1e-15 * rule
    assert native.de_damping(1.0, 1e-40, 1e3) == DAMPING_FLOOR == 1e-13

    sc, par, x0 = small_problem(n_cams=4, n_points=40, k=4)
    args = (par, sc.camera_indices, sc.image_coords, sc.obj_indices)
    res, _, _ = _solve(native, lambda x: joint_residuals(x, *args), lambda x: joint_jacobian(x, *args).toarray(), x0, 2 * len(sc.camera_indices),
                       ncp=par.n_camera_params, ftol=1e-15, xtol=1e-15, gtol=1e-13, max_nfev=60)
    assert res.status > 0 and res.optimality < 1e-10
    assert res.failed_factorisations == 0 and res.damped_steps - res.n_iterations in (0, 1)  # (+1: the fused step speculated before the gtol exit)
