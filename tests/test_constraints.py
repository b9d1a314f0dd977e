"""Rigid-distance constraint rows: oracle consistency on CPU, device parity on the GPU."""

import numpy as np
import pytest

from caliscope_amd.engine import BAProblem
from tests.constrained_scene import board_scene


def _oracle(sc, loss="linear", f_scale=1.0):
    from oracle.engine import OracleEngine

    return OracleEngine(sc["par"], sc["cam"], sc["uv"], sc["obj"], loss=loss, f_scale=f_scale, constraints=sc["constraints"])


def test_oracle_constraint_rows_are_consistent():
    """Jacobian of the constraint rows vs central differences; rows vanish on the true geometry."""
    from oracle.residuals import joint_jacobian, joint_residuals

    sc = board_scene(perturb=False, noise_px=0.0)
    args = (sc["par"], sc["cam"], sc["uv"], sc["obj"], *sc["constraints"])
    r = joint_residuals(sc["x0"], *args)
    n_con = len(sc["constraints"][2])
    assert np.abs(r[-n_con:]).max() < 1e-9
    sc = board_scene()
    args = (sc["par"], sc["cam"], sc["uv"], sc["obj"], *sc["constraints"])
    x = sc["x0"]
    J = joint_jacobian(x, *args).toarray()[-n_con:]
    rng = np.random.default_rng(0)
    for _ in range(5):
        v = rng.normal(size=x.size)
        h = 1e-6
        fd = (joint_residuals(x + h * v, *args)[-n_con:] - joint_residuals(x - h * v, *args)[-n_con:]) / (2 * h)
        assert np.abs(J @ v - fd).max() < 1e-6 * max(1.0, np.abs(fd).max())


def test_host_loop_with_constraints_matches_scipy():
    """TRF host loop on the numpy engine (constraint rows in) against scipy on the same callables."""
    from caliscope_amd.least_squares import least_squares
    from oracle.engine import OracleEngine
    from oracle.residuals import joint_jacobian, joint_residuals
    from oracle.solver import optimize_scipy
    from tests.helpers import aligned_difference

    sc = board_scene()
    par, x0 = sc["par"], sc["x0"]
    ref = optimize_scipy(par, sc["cam"], sc["uv"], sc["obj"], x0, constraints=sc["constraints"])
    factory = lambda prob: OracleEngine(prob.parameterization, prob.camera_indices, prob.image_coords, prob.obj_indices,
                                        loss=prob.loss, f_scale=prob.f_scale, constraints=prob.constraint_args())
    res = least_squares(joint_residuals, x0, args=(par, sc["cam"], sc["uv"], sc["obj"], *sc["constraints"]), jac=joint_jacobian,
                        x_scale="jac", method="trf", bounds=par.bounds(), engine_factory=factory)
    assert res.status > 0 and ref.status > 0
    assert abs(res.cost - ref.cost) <= 1e-8 * ref.cost
    pos, ang, scale = aligned_difference(par, res.x, ref.x)
    assert pos < 1e-6 and ang < 1e-6 and abs(scale - 1.0) < 1e-6, (pos, ang, scale)


@pytest.mark.gpu
@pytest.mark.parametrize("loss", ["linear", "huber"])
def test_device_evaluation_with_constraints(loss):
    from caliscope_amd.hip_engine import HipEngine
    from oracle.residuals import joint_residuals

    sc = board_scene()
    fs = 2.0 / 1394.6 if loss != "linear" else 1.0
    ga, gb, dist, w = sc["constraints"]
    prob = BAProblem(sc["par"], sc["cam"], sc["uv"], sc["obj"], loss=loss, f_scale=fs, constraint_groups_a=ga, constraint_groups_b=gb,
                     constraint_distances=dist, constraint_weights=w)
    hip, ora = HipEngine(prob), _oracle(sc, loss, fs)
    r_ref = joint_residuals(sc["x0"], sc["par"], sc["cam"], sc["uv"], sc["obj"], ga, gb, dist, w)
    r, cost = hip.residuals(sc["x0"])
    assert r.shape == r_ref.shape and np.abs(r - r_ref).max() < 1e-12 * np.abs(r_ref).max()
    c_h, c_o = hip.begin(sc["x0"]), ora.begin(sc["x0"])
    assert abs(c_h - c_o) <= 1e-13 * c_o and abs(cost - c_o) <= 1e-13 * c_o
    lh, lo = hip.linearize(), ora.linearize()
    assert np.abs(hip.get_vector(2) - ora.g).max() < 1e-11 * np.abs(ora.g).max()          # gradient incl. the constraint rows
    # Jacobi column norms; rows in the linear part of a robust loss are scaled by sqrt(max(EPS, exact cancellation)),
    # which amplifies rounding (same note and bound as tests/test_gpu_parity.py)
    assert np.abs(hip.get_vector(4) - ora.scale_inv).max() < (1e-12 if loss == "linear" else 1e-7) * np.abs(ora.scale_inv).max()
    # at this x0 nearly every row sits in huber's linear region: gh_sq / jg_sq are sums over sqrt(EPS)-floor columns, i.e.
    # rounding noise in scipy as much as here (note in test_step_parity) — compared for the linear loss only
    for fld in ("g_norm_inf", "x_scaled_norm", "x_norm") + (("gh_sq", "jg_sq") if loss == "linear" else ()):
        assert abs(getattr(lh, fld) - getattr(lo, fld)) <= 1e-9 * abs(getattr(lo, fld)), fld
    hip.close()


@pytest.mark.gpu
def test_device_step_with_constraints():
    from caliscope_amd.hip_engine import HipEngine

    sc = board_scene()
    ga, gb, dist, w = sc["constraints"]
    prob = BAProblem(sc["par"], sc["cam"], sc["uv"], sc["obj"], constraint_groups_a=ga, constraint_groups_b=gb,
                     constraint_distances=dist, constraint_weights=w)
    hip, ora = HipEngine(prob), _oracle(sc)
    hip.begin(sc["x0"]); ora.begin(sc["x0"])
    hip.linearize(); ora.linearize()
    for lam in (1e-3, 1e-7):
        sh, so = hip.newton_step(lam), ora.newton_step(lam)
        assert sh.ok and so.ok
        s_h = hip.get_vector(3)
        assert np.abs(s_h - ora.s).max() < 1e-8 * np.abs(ora.s).max(), lam
        for fld in ("p_sq", "gh_dot_p", "w_sq"):
            assert abs(getattr(sh, fld) - getattr(so, fld)) <= 1e-7 * abs(getattr(so, fld)), (fld, lam)
        gh, go = hip.subspace_gram(0.3, -1.2, 1.1, 0.4), ora.subspace_gram(0.3, -1.2, 1.1, 0.4)
        assert np.allclose(gh, go, rtol=1e-7)
        th, to = hip.trial(-1e-3, 0.5), ora.trial(-1e-3, 0.5)
        assert th.finite and abs(th.cost - to.cost) <= 1e-9 * to.cost
    hip.close()


@pytest.mark.gpu
def test_component_larger_than_the_lds_copy():
    """An 18 x 16 board: 288 points and ~1100 rows in ONE connected component per frame.  Up to 256 points a component's
    per-point factors sit in the workgroup's LDS, beyond that in global scratch (ConPlan::big) — the reference has no
    limit (core/capture_volume.py:446-531: all static markers of a room are one component)."""
    from caliscope_amd.hip_engine import HipEngine

    sc = board_scene(n_cams=4, n_frames=2, rows=18, cols=16, spacing=0.02)
    assert sc["n_per"] == 288
    ga, gb, dist, w = sc["constraints"]
    prob = BAProblem(sc["par"], sc["cam"], sc["uv"], sc["obj"], constraint_groups_a=ga, constraint_groups_b=gb,
                     constraint_distances=dist, constraint_weights=w)
    hip, ora = HipEngine(prob), _oracle(sc)
    c_h, c_o = hip.begin(sc["x0"]), ora.begin(sc["x0"])
    assert abs(c_h - c_o) <= 1e-13 * c_o
    hip.linearize(); ora.linearize()
    assert np.abs(hip.get_vector(2) - ora.g).max() < 1e-11 * np.abs(ora.g).max()
    for lam in (1e-3, 1e-7):
        sh, so = hip.newton_step(lam), ora.newton_step(lam)
        assert sh.ok and so.ok
        assert np.abs(hip.get_vector(3) - ora.s).max() < 1e-8 * np.abs(ora.s).max(), lam
        for fld in ("p_sq", "gh_dot_p", "w_sq"):
            assert abs(getattr(sh, fld) - getattr(so, fld)) <= 1e-7 * abs(getattr(so, fld)), (fld, lam)
    res = hip.solve(sc["x0"])
    hip.close()
    assert res.status > 0
    pts = res.x[sc["par"].n_camera_params:].reshape(-1, 3)
    got = np.linalg.norm(pts[ga].mean(axis=1) - pts[gb].mean(axis=1), axis=1)
    assert np.abs(got - dist).max() < 0.01


@pytest.mark.gpu
def test_converged_parity_with_constraints():
    """Full solve through the reference seam against scipy on the oracle callables (constraints fix the scale, so the
    gauge left is a rigid motion)."""
    from caliscope_amd.least_squares import least_squares
    from oracle.residuals import joint_jacobian, joint_residuals
    from oracle.solver import optimize_scipy
    from tests.helpers import aligned_difference

    sc = board_scene()
    par, x0 = sc["par"], sc["x0"]
    ref = optimize_scipy(par, sc["cam"], sc["uv"], sc["obj"], x0, constraints=sc["constraints"])
    res = least_squares(joint_residuals, x0, args=(par, sc["cam"], sc["uv"], sc["obj"], *sc["constraints"]), jac=joint_jacobian,
                        x_scale="jac", method="trf", bounds=par.bounds())
    assert res.status > 0 and ref.status > 0
    assert abs(res.cost - ref.cost) <= 1e-8 * ref.cost
    pos, ang, scale = aligned_difference(par, res.x, ref.x)
    assert pos < 1e-6 and ang < 1e-6 and abs(scale - 1.0) < 1e-6, (pos, ang, scale)  # north_star tolerance
    # the constraints did their job: board edge lengths match the targets to a few sigma
    pts = res.x[par.n_camera_params:].reshape(-1, 3)
    ga, gb, dist, _ = sc["constraints"]
    got = np.linalg.norm(pts[ga].mean(axis=1) - pts[gb].mean(axis=1), axis=1)
    assert np.abs(got - dist).max() < 0.01


@pytest.mark.gpu
def test_capture_volume_optimize_with_constraints_real_session(golden_dir):
    """The reference's 4-camera ChArUco session (4 x 5 squares of 5.4 cm -> 3 x 4 inner corners) with the board's truss
    as constraint rows, refining intrinsics (the case the constraints exist for: without a metric row focal length
    and scale are coupled, reference capture_volume.py:337-343) — device path vs scipy on the oracle rows."""
    from caliscope_amd.bundle_parameterization import BundleParameterization
    from caliscope_amd.cameras import CameraArray
    from caliscope_amd.capture_volume import CaptureVolume
    from caliscope_amd.constraints import ConstraintSet
    from caliscope_amd.point_data import ImagePoints, WorldPoints
    from oracle.solver import optimize_scipy, rms_reprojection_px
    from tests.helpers import aligned_difference

    d = golden_dir / "post_optimization"
    pitch = 0.054
    grid = np.array([[(c + 1) * pitch, (r + 1) * pitch, 0.0] for r in range(4) for c in range(3)], dtype=np.float32)
    cs = ConstraintSet.from_grid(grid, pitch)
    cv = CaptureVolume(CameraArray.from_toml(d / "camera_array.toml"), ImagePoints.from_csv(d / "xy_CHARUCO.csv"),
                       WorldPoints.from_csv(d / "xyz_CHARUCO.csv"), cs)
    ga, gb, dist, sig = cv._build_constraint_arrays()
    assert len(dist) > 1000 and cv.rigidity_report().rmse_mm < 20.0
    _, cam, uv, obj = cv._matched_arrays()
    f_median = float(np.median([c.matrix[0, 0] for c in cv.camera_array.posed_cameras.values()]))
    con = (ga, gb, dist, (1.0 / f_median) / sig)
    for refine in (False, True):
        # the product call next to the reference's call: scipy's default for a sparse Jacobian is the inexact 'lsmr'
        # step, so the two stop at different points of the ftol = 1e-8 plateau: cost and RMS agree (north_star: RMS
        # within 1e-4 px), positions only to the termination tolerance
        opt = cv.optimize(refine_intrinsics=refine)
        assert opt.optimization_status.converged
        par = BundleParameterization.from_camera_array(cv.camera_array, n_points=len(cv.world_points), refine_intrinsics=refine)
        x0 = par.pack(cv.camera_array, cv.world_points.points)
        ref = optimize_scipy(par, cam, uv, obj, x0, constraints=con)
        assert ref.status > 0
        assert abs(opt.optimization_status.final_cost - ref.cost) <= 1e-6 * ref.cost
        assert opt.optimization_status.final_cost <= ref.cost * (1 + 1e-9)  # exact steps never end above the inexact ones here
        fx = np.array([c.matrix[0, 0] for _, c in sorted(opt.camera_array.cameras.items())])
        err = opt.reprojection_report.overall_rmse
        assert abs(err - rms_reprojection_px(par, cam, uv, obj, ref.x)) < 1e-4, (refine, fx)
        assert opt.rigidity_report().rmse_mm <= cv.rigidity_report().rmse_mm
        free = cv.optimize(use_constraints=False, refine_intrinsics=refine)
        assert opt.rigidity_report().rmse_mm <= free.rigidity_report().rmse_mm
    # fully converged points (tight tolerances, scipy with exact SVD steps): 1e-6 on poses and points (north_star)
    from caliscope_amd.least_squares import least_squares
    from oracle.residuals import joint_jacobian, joint_residuals

    tight = dict(ftol=1e-15, xtol=1e-15, gtol=1e-12)
    got = least_squares(joint_residuals, x0, args=(par, cam, uv, obj, *con), jac=joint_jacobian, x_scale="jac", method="trf",
                        bounds=par.bounds(), max_nfev=200, **tight)
    ref = optimize_scipy(par, cam, uv, obj, x0, constraints=con, tr_solver="exact", max_nfev=40, **tight)
    assert got.status > 0 and ref.optimality < 1e-8  # scipy may still be polishing at max_nfev: gradient says converged
    pos, ang, scale = aligned_difference(par, got.x, ref.x)
    assert pos < 1e-6 and ang < 1e-6 and abs(scale - 1.0) < 1e-6, (pos, ang, scale)
    assert abs(got.cost - ref.cost) <= 1e-10 * ref.cost


@pytest.mark.gpu
def test_capture_volume_static_markers_and_centroid_rows():
    """Static markers (one world point per corner for the whole recording, seen in every frame) tied by a centroid
    link and a corner link: device path vs the numpy engine."""
    from oracle.engine import OracleEngine
    from tests.constrained_scene import marker_volume
    from tests.helpers import aligned_difference

    vol, par = marker_volume()
    ga, gb, dist, sig = vol._build_constraint_arrays()
    assert (ga[:, :1] != ga).any(axis=1).sum() == 1  # exactly one centroid row
    factory = lambda prob: OracleEngine(prob.parameterization, prob.camera_indices, prob.image_coords, prob.obj_indices,
                                        loss=prob.loss, f_scale=prob.f_scale, constraints=prob.constraint_args())
    got, ref = vol.optimize(), vol.optimize(_engine_factory=factory)
    assert got.optimization_status.converged and ref.optimization_status.converged
    assert abs(got.optimization_status.final_cost - ref.optimization_status.final_cost) <= 1e-8 * ref.optimization_status.final_cost
    pos, ang, scale = aligned_difference(par, par.pack(got.camera_array, got.world_points.points),
                                         par.pack(ref.camera_array, ref.world_points.points))
    assert pos < 1e-6 and ang < 1e-6 and abs(scale - 1.0) < 1e-6
    assert got.rigidity_report().rmse_mm < vol.rigidity_report().rmse_mm


@pytest.mark.gpu
@pytest.mark.parametrize("n_frames,with_constraints", [(30, False), (80, False), (80, True)])
def test_heavy_static_points_step_and_solution(n_frames, with_constraints):
    """Static marker corners are ONE world point observed again in every frame: 30 frames x 5 cameras ~ 135 rows per point
    (heavy: per-camera Schur sums, still one chunk), 80 frames ~ 360 rows (split over chunks).  Damped step and converged
    solution against the numpy engine."""
    from caliscope_amd.hip_engine import HipEngine
    from oracle.engine import OracleEngine
    from tests.constrained_scene import marker_volume
    from tests.helpers import aligned_difference

    vol, par = marker_volume(n_frames=n_frames)
    _, cam, uv, obj = vol._matched_arrays()
    con = None
    if with_constraints:
        ga, gb, dist, sig = vol._build_constraint_arrays()
        con = (ga, gb, dist, (1.0 / 1394.6) / sig)
    x0 = par.pack(vol.camera_array, vol.world_points.points)
    kw = {} if con is None else dict(constraint_groups_a=con[0], constraint_groups_b=con[1], constraint_distances=con[2], constraint_weights=con[3])
    hip, ora = HipEngine(BAProblem(par, cam, uv, obj, **kw)), OracleEngine(par, cam, uv, obj, constraints=con)
    info = hip.info()
    assert info["n_heavy_points"] == 12 and info["max_obs_per_point"] > (256 if n_frames == 80 else 100) and info["plan_state"] == 0
    c_h, c_o = hip.begin(x0), ora.begin(x0)
    assert abs(c_h - c_o) <= 1e-13 * c_o
    hip.linearize(); ora.linearize()
    assert np.abs(hip.get_vector(2) - ora.g).max() < 1e-11 * np.abs(ora.g).max()
    assert np.abs(hip.get_vector(4) - ora.scale_inv).max() < 1e-12 * np.abs(ora.scale_inv).max()
    for lam in (1e-3, 1e-8):
        sh, so = hip.newton_step(lam), ora.newton_step(lam)
        assert sh.ok and so.ok
        tol = 1e-8 if lam > 1e-6 else 1e-6  # lam = 1e-8 leaves the gauge directions nearly singular: rounding is amplified there
        assert np.abs(hip.get_vector(3) - ora.s).max() < tol * np.abs(ora.s).max(), lam
        for fld in ("p_sq", "gh_dot_p", "w_sq"):
            assert abs(getattr(sh, fld) - getattr(so, fld)) <= 10 * tol * abs(getattr(so, fld)), (fld, lam)
    got = hip.solve(x0)
    from oracle.trf_driver import trf_solve

    ref = trf_solve(ora, x0)
    assert got.status > 0 and ref.status > 0 and abs(got.cost - ref.cost) <= 1e-9 * ref.cost
    pos, ang, _ = aligned_difference(par, got.x, ref.x)
    assert pos < 1e-6 and ang < 1e-6
    hip.close()


@pytest.mark.gpu
def test_staged_driver_with_constraints_and_static_markers():
    """Stages 4-9 of calibrate_extrinsics on the device for a volume with a ConstraintSet: three solves with constraint
    rows (linear, soft_l1, linear), the filter in between keeps the static world points."""
    from caliscope_amd.calibrate_extrinsics import refine_calibration
    from caliscope_amd.point_data import STATIC_SYNC_INDEX
    from tests.constrained_scene import marker_volume

    vol, _ = marker_volume(n_frames=20)
    seen = []
    run = refine_calibration(vol, refine_intrinsics=False, progress=lambda pct, msg: seen.append(pct))
    out = run.capture_volume
    assert seen == [40, 55, 75, 90, 100] and run.dropped_static_markers == () and out.optimization_status.converged
    assert len(out.image_points) < len(vol.image_points)  # the filter removed the worst 2.5 %
    static = out.world_points.df[out.world_points.df["sync_index"] == STATIC_SYNC_INDEX]
    assert len(static) == 12
    assert out.rigidity_report().rmse_mm < 0.25 * vol.rigidity_report().rmse_mm
    assert out.reprojection_report.overall_rmse < 1.0


@pytest.mark.gpu
def test_constraint_rows_through_the_rccl_call_sites(monkeypatch):
    """A one-rank RCCL communicator exercises every all-reduce of the sharded protocol with constraint rows present
    (cost / rho sums with the rows' partials, the reduced system after the Woodbury correction); the native driver on top."""
    from caliscope_amd.hip_engine import HipEngine

    sc = board_scene(n_frames=9)
    ga, gb, dist, w = sc["constraints"]
    prob = BAProblem(sc["par"], sc["cam"], sc["uv"], sc["obj"], constraint_groups_a=ga, constraint_groups_b=gb,
                     constraint_distances=dist, constraint_weights=w)
    with HipEngine(prob) as plain:
        ref = plain.solve(sc["x0"], ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=60)
    monkeypatch.setenv("CBA_FORCE_COMM", "1")
    with HipEngine(prob) as eng:
        eng.comm_init(eng.comm_unique_id(), 0, 1)
        got = eng.solve(sc["x0"], ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=60)
    assert got.status == ref.status and got.nfev == ref.nfev and abs(got.cost - ref.cost) <= 1e-12 * ref.cost
    from tests.helpers import aligned_difference

    pos, ang, scale = aligned_difference(sc["par"], got.x, ref.x)
    assert pos < 1e-8 and ang < 1e-8 and abs(scale - 1) < 1e-8


@pytest.mark.gpu
def test_device_rows_on_the_reference_known_answers():
    """The reference's own known answers for the constraint rows (tests/test_constraints.py:599-649, :770-805, :808-885) through the
    device residual hook: rows appended after the reprojection rows, zero at the exact distance, corner and centroid values."""
    from caliscope_amd.hip_engine import HipEngine
    from tests.test_oracle_pins import _one_camera_two_points

    cam_idx, uv, obj = np.array([0, 0], dtype=np.int32), np.array([[200.0, 200.0], [240.0, 200.0]]), np.array([0, 1], dtype=np.int32)

    def rows(points, ga, gb, dist, w):
        par, x = _one_camera_two_points(points)
        prob = BAProblem(par, cam_idx, uv, obj, constraint_groups_a=np.array(ga, dtype=np.int32), constraint_groups_b=np.array(gb, dtype=np.int32),
                         constraint_distances=np.array(dist), constraint_weights=np.array(w))
        with HipEngine(prob, evaluation_only=True) as eng:
            r, _ = eng.residuals(x)
        with HipEngine(BAProblem(par, cam_idx, uv, obj), evaluation_only=True) as eng:
            r0, _ = eng.residuals(x)
        assert r.size == r0.size + len(dist) and np.array_equal(r[: r0.size], r0)
        return r[r0.size:]

    assert abs(rows([[0.0, 0, 0], [1.0, 0, 0]], [[0, 0, 0, 0]], [[1, 1, 1, 1]], [1.0], [0.5])[0]) < 1e-15
    pts = np.array([[0.1, 0.2, 0.3], [1.4, -0.5, 0.7]])
    got = rows(pts, [[0, 0, 0, 0]], [[1, 1, 1, 1]], [1.0], [0.5])[0]
    assert abs(got - (np.linalg.norm(pts[0] - pts[1]) - 1.0) * 0.5) < 1e-15
    sq = np.array([[-0.5, 0.5, 0], [0.5, 0.5, 0], [0.5, -0.5, 0], [-0.5, -0.5, 0.0]])
    both = np.vstack([sq, sq + [2.0, 0, 0]])
    r = rows(both, [[0, 1, 2, 3], [0, 1, 2, 3]], [[4, 5, 6, 7], [4, 5, 6, 7]], [2.0, 1.5], [3.0, 3.0])
    assert abs(r[0]) < 1e-14 and abs(r[1] - 1.5) < 1e-14
