"""Production trust-region driver (oracle/trf_driver.py) on the numpy oracle engine vs the reference's
scipy call (oracle/solver.py == capture_volume.py:387-411).  Three-tier parity protocol of SURVEY.md §7:
cost / RMS equality, gauge-aligned converged state <= 1e-6 relative, RMS px within 1e-4."""
import numpy as np
import pandas as pd
import pytest

from caliscope_amd.bundle_parameterization import BundleParameterization
from caliscope_amd.cameras import CameraArray
from oracle.trf_driver import solve_subspace_2d, trf_solve
from oracle.engine import OracleEngine
from oracle.solver import optimize_scipy, rms_reprojection_px
from tests.helpers import aligned_difference, small_problem


def _both(sc, par, x0, loss="linear", f_scale=1.0, **tol):
    ref = optimize_scipy(par, sc.camera_indices, sc.image_coords, sc.obj_indices, x0, loss=loss, f_scale=f_scale, **tol)
    eng = OracleEngine(par, sc.camera_indices, sc.image_coords, sc.obj_indices, loss=loss, f_scale=f_scale)
    got = trf_solve(eng, x0, **tol)
    return ref, got


def test_linear_loss_default_tolerances_match_scipy():
    sc, par, x0 = small_problem(n_cams=8, n_points=400, k=8)
    ref, got = _both(sc, par, x0)
    assert got.status in (1, 2, 3, 4) and ref.status in (1, 2, 3, 4)
    assert abs(got.nfev - ref.nfev) <= 2
    assert abs(got.cost - ref.cost) <= 1e-8 * ref.cost
    args = (par, sc.camera_indices, sc.image_coords, sc.obj_indices)
    assert abs(rms_reprojection_px(*args, got.x) - rms_reprojection_px(*args, ref.x)) < 1e-4
    pos, ang, scale = aligned_difference(par, got.x, ref.x)
    assert pos < 1e-6 and ang < 1e-6


def test_tight_tolerances_gauge_aligned_parity():
    sc, par, x0 = small_problem(n_cams=8, n_points=400, k=8)
    tol = dict(ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=200)
    ref, got = _both(sc, par, x0, **tol)
    assert abs(got.cost - ref.cost) <= 1e-12 * ref.cost
    pos, ang, _ = aligned_difference(par, got.x, ref.x)
    assert pos < 1e-7 and ang < 1e-7, (pos, ang)


@pytest.mark.parametrize("loss", ["huber", "soft_l1"])
def test_robust_losses_with_outliers(loss):
    """Convex robust losses (the ones the product uses: soft_l1 in calibrate_extrinsics.py:230-238,
    huber in BASELINE cfg 3): same minimiser as scipy, gauge-aligned."""
    sc, par, x0 = small_problem(n_cams=6, n_points=300, k=6, outliers=0.05)
    fs = sc.f_scale_1px() * 2.0
    tol = dict(ftol=1e-10, xtol=1e-10, gtol=1e-10, max_nfev=300)
    ref, got = _both(sc, par, x0, loss=loss, f_scale=fs, **tol)
    assert got.status > 0
    assert got.cost <= ref.cost * (1 + 1e-6)
    pos, ang, _ = aligned_difference(par, got.x, ref.x)
    assert pos < 1e-5 and ang < 1e-5, (pos, ang)


@pytest.mark.parametrize("loss", ["cauchy", "arctan"])
def test_nonconvex_losses_descend(loss):
    """cauchy / arctan are non-convex: scipy-LSMR and an exact damped step legitimately end in different
    local minima, so only descent and first-order progress are asserted (primitive-level parity of cost,
    gradient and blocks is covered by the engine tests)."""
    sc, par, x0 = small_problem(n_cams=6, n_points=300, k=6, outliers=0.05)
    fs = sc.f_scale_1px() * 2.0
    eng = OracleEngine(par, sc.camera_indices, sc.image_coords, sc.obj_indices, loss=loss, f_scale=fs)
    c0 = eng.begin(x0)
    g0 = eng.linearize().g_norm_inf
    got = trf_solve(eng, x0, ftol=1e-10, xtol=1e-10, gtol=1e-10, max_nfev=150)
    assert got.cost < 0.5 * c0 and got.optimality < 0.1 * g0


def test_real_session_post_optimization(golden_dir):
    d = golden_dir / "post_optimization"
    ca = CameraArray.from_toml(d / "camera_array.toml")
    xy, xyz = pd.read_csv(d / "xy_CHARUCO.csv"), pd.read_csv(d / "xyz_CHARUCO.csv")
    key = {k: i for i, k in enumerate(zip(xyz.sync_index, xyz.object_id, xyz.keypoint_id))}
    obj = np.array([key[k] for k in zip(xy.sync_index, xy.object_id, xy.keypoint_id)], dtype=np.int32)
    cam = np.array([ca.posed_cam_id_to_index[c] for c in xy.cam_id], dtype=np.int16)
    uv = xy[["img_loc_x", "img_loc_y"]].to_numpy()
    par = BundleParameterization.from_camera_array(ca, n_points=len(xyz), refine_intrinsics=False)
    x0 = par.pack(ca, xyz[["x_coord", "y_coord", "z_coord"]].to_numpy())
    ref = optimize_scipy(par, cam, uv, obj, x0)
    got = trf_solve(OracleEngine(par, cam, uv, obj), x0)
    r0, r_ref, r_got = (rms_reprojection_px(par, cam, uv, obj, x) for x in (x0, ref.x, got.x))
    assert r_ref < r0 and abs(r_got - r_ref) < 1e-4
    assert abs(got.cost - ref.cost) < 1e-8 * ref.cost


def test_max_nfev_and_status_zero():
    sc, par, x0 = small_problem(n_cams=4, n_points=100, k=4)
    eng = OracleEngine(par, sc.camera_indices, sc.image_coords, sc.obj_indices)
    got = trf_solve(eng, x0, max_nfev=2, ftol=1e-15, xtol=1e-15, gtol=1e-15)
    assert got.status == 0 and got.nfev == 2 and got.message == "max_evaluations"


def test_already_converged_returns_gtol():
    sc, par, x0 = small_problem(n_cams=4, n_points=100, k=4)
    eng = OracleEngine(par, sc.camera_indices, sc.image_coords, sc.obj_indices)
    first = trf_solve(eng, x0, ftol=1e-15, xtol=1e-15, gtol=1e-12, max_nfev=100)
    again = trf_solve(OracleEngine(par, sc.camera_indices, sc.image_coords, sc.obj_indices), first.x, gtol=1e-8)
    assert again.status == 1 and again.nfev == 1


def test_subspace_solver_against_brute_force():
    rng = np.random.default_rng(1)
    th = np.linspace(0, 2 * np.pi, 200001)
    for _ in range(50):
        A = rng.normal(size=(2, 2))
        B = A @ A.T if rng.random() < 0.7 else (A + A.T)  # PD or indefinite
        g = rng.normal(size=2)
        radius = float(rng.uniform(0.05, 3.0))
        p = solve_subspace_2d(B, g, radius)
        assert p @ p <= radius**2 * (1 + 1e-9)
        val = 0.5 * p @ B @ p + g @ p
        circ = radius * np.vstack([np.cos(th), np.sin(th)])
        best = (0.5 * np.sum(circ * (B @ circ), axis=0) + g @ circ).min()
        try:
            pn = -np.linalg.solve(B, g)
            if np.all(np.linalg.eigvalsh(B) > 0) and pn @ pn <= radius**2:
                best = min(best, 0.5 * pn @ B @ pn + g @ pn)
        except np.linalg.LinAlgError:
            pass
        assert val <= best + 1e-7 * (1 + abs(best))
