"""The situations the reference's synthetic suite puts ``CaptureVolume.optimize`` in, with the reference's own acceptance
bounds, once on the numpy engine (CPU) and once through the C ABI on the device (``-m gpu``):

* joint intrinsic + extrinsic recovery, E1-E5b (reference tests/synthetic/test_intrinsic_recovery.py:83-330);
* gross outliers: convergence, set recovery of the percentile filter, poses after re-optimisation
  (tests/synthetic/test_outlier_robustness.py:39-128);
* two-phase robust solve (linear -> soft_l1 at one pixel) (tests/synthetic/test_robust_loss.py:32-115);
* optimise -> filter -> optimise on clean data (tests/synthetic/test_multistage_flow.py:24-62);
* a 15-camera ring (tests/synthetic/test_large_ring.py:17-49).

The reference bootstraps x0 with OpenCV PnP; here x0 is perturbed ground truth (tests/scenario_scenes.py)."""
import numpy as np
import pytest

from caliscope_amd.capture_volume import CaptureVolume
from caliscope_amd.synthetic import make_scene
from tests.scenario_scenes import moving_board_volume, pose_errors


def _numpy_factory(problem):
    from oracle.engine import OracleEngine
    from oracle.residuals import joint_residuals

    eng = OracleEngine(problem.parameterization, problem.camera_indices, problem.image_coords, problem.obj_indices,
                       loss=problem.loss, f_scale=problem.f_scale, constraints=problem.constraint_args())

    def residuals(x):
        r = joint_residuals(x, problem.parameterization, problem.camera_indices, problem.image_coords, problem.obj_indices)
        return r, 0.5 * float(r @ r)

    eng.residuals = residuals
    return eng


@pytest.fixture(params=["numpy", pytest.param("hip", marks=pytest.mark.gpu)])
def factory(request):
    """``None`` = the product's default engine (the HIP library); the numpy engine runs the same host code on CPU."""
    return _numpy_factory if request.param == "numpy" else None


def _report(volume, factory):
    return volume.compute_reprojection_report(factory) if factory else volume.reprojection_report


def _intrinsic_errors(volume, truth):
    out = {}
    for cam_id, cam in volume.camera_array.posed_cameras.items():
        true = truth["cameras"].cameras[cam_id]
        out[cam_id] = (abs(cam.matrix[0, 0] - true.matrix[0, 0]) / true.matrix[0, 0], abs(cam.distortions[0] - true.distortions[0]),
                       abs(cam.distortions[1] - true.distortions[1]))
    return out


@pytest.mark.parametrize("perturbation", [dict(f_scale=1.03), dict(k1_delta=0.02), dict(f_scale=1.03, k1_delta=0.02, k2_delta=0.05)],
                         ids=["E1_f", "E2_k1", "E2b_f_k1_k2"])
def test_intrinsic_recovery(factory, perturbation):
    vol, truth = moving_board_volume(**perturbation)
    opt = vol.optimize(refine_intrinsics=True, _engine_factory=factory)
    assert opt.optimization_status.converged and opt.optimization_status.bound_warnings == ()
    for cam_id, (f_err, k1_err, k2_err) in _intrinsic_errors(opt, truth).items():
        assert f_err < 0.01 and k1_err < 0.02 and k2_err < 0.03, (cam_id, f_err, k1_err, k2_err)
    trans, rot = pose_errors(opt, truth)
    assert rot < 1.0 and trans < 0.02, (trans, rot)
    # the input volume is untouched (reference capture_volume.py:360: optimize works on a deep copy)
    assert all(abs(c.matrix[0, 0] / truth["cameras"].cameras[i].matrix[0, 0] - perturbation.get("f_scale", 1.0)) < 1e-12
               for i, c in vol.camera_array.cameras.items())


def test_constraints_as_metric_anchor(factory):
    """E3: the board's rigid distances must not degrade focal-length recovery (2x margin, 0.5 % floor) — and they pin the
    scale, which free BA leaves to the gauge."""
    free_vol, truth = moving_board_volume(f_scale=1.03)
    tied_vol, _ = moving_board_volume(f_scale=1.03, constraints=True)
    free = free_vol.optimize(refine_intrinsics=True, use_constraints=False, _engine_factory=factory)
    tied = tied_vol.optimize(refine_intrinsics=True, use_constraints=True, _engine_factory=factory)
    assert free.optimization_status.converged and tied.optimization_status.converged
    err_free, err_tied = _intrinsic_errors(free, truth), _intrinsic_errors(tied, truth)
    for cam_id in err_free:
        assert err_tied[cam_id][0] <= max(2.0 * err_free[cam_id][0], 0.005), (cam_id, err_tied[cam_id], err_free[cam_id])
    report = tied.rigidity_report()
    assert report.violations and report.rmse_mm < 1.0 and report.rmse_mm <= tied_vol.rigidity_report().rmse_mm


def test_negative_control_stationary_board(factory):
    """E4: a small stationary board far from the cameras does not determine focal length: at least half of the cameras
    keep more than half of the injected 3 % error (or end near a bound)."""
    vol, truth = moving_board_volume(radius=3.0, rows=3, cols=4, spacing=0.03, n_frames=10, stationary=True, start=(0.0, 0.0, 0.5),
                                     f_scale=1.03)
    opt = vol.optimize(refine_intrinsics=True, strict=False, _engine_factory=factory)
    if not opt.optimization_status.converged:
        return
    warned = len(opt.optimization_status.bound_warnings) > 0
    poor = sum(1 for f_err, _, _ in _intrinsic_errors(opt, truth).values() if f_err > 0.5 * 0.03 or warned)
    assert poor >= len(opt.camera_array.posed_cameras) // 2


def test_outlier_contamination_does_not_drag_focal_length(factory):
    """E5b: 5 % gross outliers, correct starting intrinsics, free intrinsics: f stays within 3 % of the truth."""
    vol, truth = moving_board_volume(outliers=0.05)
    opt = vol.optimize(refine_intrinsics=True, strict=False, _engine_factory=factory)
    if not opt.optimization_status.converged:
        pytest.skip("joint BA on corrupted data did not converge")
    assert max(e[0] for e in _intrinsic_errors(opt, truth).values()) < 0.03


# -- outliers, robust loss, multi-stage flow on the ring scene ------------------------------------------------------------
def _ring(outliers, n_cams=4, n_points=400, k=4, seed=42):
    sc = make_scene(n_cams=n_cams, n_points=n_points, n_obs=n_points * k, outliers=outliers, seed=seed)
    vol = CaptureVolume.from_arrays(sc.cameras_init, sc.camera_indices, sc.image_coords, sc.obj_indices, sc.points_init)
    truth = dict(cameras=sc.cameras_true, points=sc.points_true)
    bad = set(zip(sc.camera_indices[sc.outlier_rows].tolist(), sc.obj_indices[sc.outlier_rows].tolist()))
    return vol, truth, bad


def _keys(volume):
    df = volume.image_points.df
    return set(zip(df["cam_id"].tolist(), df["object_id"].tolist()))


def _filter_recovery(optimized, bad, factory, percentile=5):
    filtered = optimized.filter_by_percentile_error(percentile, scope="overall", _engine_factory=factory)
    removed = _keys(optimized) - _keys(filtered)
    hit = removed & bad
    return filtered, len(hit) / max(len(removed), 1), len(hit) / max(len(bad), 1)


def test_outliers_filter_set_recovery_and_poses(factory):
    vol, truth, bad = _ring(outliers=0.05)
    opt = vol.optimize(_engine_factory=factory)
    assert opt.optimization_status.converged  # 5 % gross outliers degrade but do not break the solver
    filtered, precision, recall = _filter_recovery(opt, bad, factory)
    assert precision >= 0.60 and recall >= 0.60, (precision, recall)
    again = filtered.optimize(_engine_factory=factory)
    trans, rot = pose_errors(again, _subset_truth(again, truth))
    assert rot < 1.0 and trans < 0.010, (trans, rot)


def _subset_truth(volume, truth):
    """Ground truth restricted to the world points the (filtered) volume still has."""
    ids = volume.world_points.df["object_id"].to_numpy()
    return dict(cameras=truth["cameras"], points=truth["points"][ids])


def _two_phase(vol, factory):
    linear = vol.optimize(_engine_factory=factory)
    return linear, linear.optimize(loss="soft_l1", f_scale=linear.pixel_f_scale(px=1.0), max_nfev=2000, ftol=1e-4, strict=False,
                                   _engine_factory=factory)


def test_two_phase_robust_solve(factory):
    vol, truth, bad = _ring(outliers=0.05)
    linear, robust = _two_phase(vol, factory)
    assert robust.optimization_status is not None
    lin_trans, _ = pose_errors(linear, truth)
    rob_trans, _ = pose_errors(robust, truth)
    assert rob_trans <= 1.1 * lin_trans, (rob_trans, lin_trans)
    _, _, lin_recall = _filter_recovery(linear, bad, factory)
    _, _, rob_recall = _filter_recovery(robust, bad, factory)
    assert rob_recall >= lin_recall and rob_recall >= 0.60, (rob_recall, lin_recall)


def test_filter_then_optimize_on_clean_data(factory):
    vol, truth, _ = _ring(outliers=0.0)
    first = vol.optimize(_engine_factory=factory)
    filtered = first.filter_by_percentile_error(5, scope="per_camera", _engine_factory=factory)
    second = filtered.optimize(_engine_factory=factory)
    assert second.optimization_status.converged
    assert _report(second, factory).overall_rmse <= _report(first, factory).overall_rmse
    trans, rot = pose_errors(second, _subset_truth(second, truth))
    assert rot < 0.5 and trans < 0.005, (trans, rot)


def test_large_ring(factory):
    vol, truth, _ = _ring(outliers=0.0, n_cams=15, n_points=300, k=8)
    opt = vol.optimize(_engine_factory=factory)
    assert opt.optimization_status.converged and _report(opt, factory).overall_rmse < 2.0
    trans, rot = pose_errors(opt, truth)
    assert rot < 0.5 and trans < 0.005, (trans, rot)


def test_stationary_planar_board(factory):
    """Globally coplanar, redundant frames (reference tests/synthetic/test_planar_degeneracy.py:46-76): with fixed
    intrinsics there is no extra degeneracy — valid poses at 4x the ring tolerances."""
    vol, truth = moving_board_volume(radius=2.0, rows=4, cols=6, spacing=0.05, n_frames=10, stationary=True, start=(0.0, 0.0, 0.5))
    opt = vol.optimize(strict=False, _engine_factory=factory)
    assert _report(opt, factory).overall_rmse < 5.0
    trans, rot = pose_errors(opt, truth)
    assert rot < 2.0 and trans < 0.020, (trans, rot)


def test_chain_linked_cameras(factory):
    """Every point is shared by one pair of neighbouring cameras only (reference tests/synthetic/test_chain_linked.py:52-97):
    the solve may stop on max_nfev, poses stay within the reference's ceilings and the scale stays metric."""
    from tests.helpers import camera_centres_and_rotations, umeyama
    from tests.scenario_scenes import chain_volume

    vol, truth = chain_volume()
    df = vol.image_points.df
    pairs = df.groupby("object_id")["cam_id"].agg(lambda c: tuple(sorted(c)))
    assert all(b - a == 1 for a, b in pairs)  # tridiagonal coverage
    opt = vol.optimize(strict=False, max_nfev=5000, _engine_factory=factory)
    trans, rot = pose_errors(opt, truth)
    assert rot < 10.0 and trans < 1.0, (trans, rot)
    from caliscope_amd.bundle_parameterization import BundleParameterization

    par = BundleParameterization.from_camera_array(opt.camera_array, n_points=len(truth["points"]), refine_intrinsics=False)
    got, _ = camera_centres_and_rotations(par, par.pack(opt.camera_array, opt.world_points.points))
    ref, _ = camera_centres_and_rotations(par, par.pack(truth["cameras"], truth["points"]))
    scale, _, _ = umeyama(got, ref)
    assert abs(scale - 1.0) < 0.02, scale
