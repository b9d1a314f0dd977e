"""Stages 5-9 of calibrate_extrinsics (optimize -> gate -> robust optimize -> filter -> optimize) and the
filter semantics of the reference, on CPU through the numpy engine."""
import numpy as np
import pandas as pd
import pytest

from caliscope_amd.calibrate_extrinsics import compute_depth_ratios, refine_calibration
from caliscope_amd.capture_volume import CaptureVolume
from caliscope_amd.synthetic import make_scene
from oracle.engine import OracleEngine


class _OracleFactory:
    """BAEngine factory for the host-side mirrors: numpy engine + the residual hook they use."""

    def __call__(self, problem):
        eng = OracleEngine(problem.parameterization, problem.camera_indices, problem.image_coords, problem.obj_indices,
                           loss=problem.loss, f_scale=problem.f_scale)

        def residuals(x):
            from oracle.residuals import joint_residuals

            r = joint_residuals(x, problem.parameterization, problem.camera_indices, problem.image_coords, problem.obj_indices)
            return r, 0.5 * float(r @ r)

        eng.residuals = residuals
        return eng


def _volume(outliers=0.05, n_cams=6, n_points=300, k=6):
    sc = make_scene(n_cams=n_cams, n_points=n_points, n_obs=n_points * k, outliers=outliers)
    cv = CaptureVolume.from_arrays(sc.cameras_init, sc.camera_indices, sc.image_coords, sc.obj_indices, sc.points_init)
    return sc, cv


def test_filter_by_percentile_matches_reference_semantics():
    sc, cv = _volume()
    f = _OracleFactory()
    rep = cv.compute_reprojection_report(f)
    raw = rep.raw_errors
    out = cv.filter_by_percentile_error(10.0, _engine_factory=f)
    # per camera: exactly the observations at or below that camera's 90th percentile survive
    kept = 0
    for cam_id in cv.camera_array.posed_cameras:
        e = raw.loc[raw.cam_id == cam_id, "euclidean_error"].to_numpy()
        kept += int((e <= np.percentile(e, 90)).sum())
    assert len(out.image_points) == kept and out.optimization_status is None
    overall = cv.filter_by_percentile_error(10.0, scope="overall", _engine_factory=f)
    assert len(overall.image_points) == int((raw.euclidean_error <= np.percentile(raw.euclidean_error, 90)).sum())
    # the safety floor keeps the best `min_per_camera` observations of every camera
    floor = cv.filter_by_absolute_error(1e-6, min_per_camera=25, _engine_factory=f)
    counts = floor.image_points.df.cam_id.value_counts()
    assert all(counts[c] == 25 for c in cv.camera_array.posed_cameras)
    assert len(floor.world_points) <= len(cv.world_points)  # orphaned world points are pruned
    # the filtered volumes carry the observation -> point map over from the old volume: same as a fresh merge
    for vol in (out, overall, floor):
        assert np.array_equal(vol.img_to_obj_map, vol._compute_img_to_obj_map()) and (vol.img_to_obj_map >= 0).all()
        assert vol.image_points.df.index.equals(pd.RangeIndex(len(vol.image_points)))
    with pytest.raises(ValueError):
        cv.filter_by_percentile_error(0.0)
    with pytest.raises(ValueError):
        cv.filter_by_percentile_error(5.0, scope="nope")


def test_depth_ratio_gate_values():
    sc, cv = _volume(outliers=0.0)
    ratios = compute_depth_ratios(cv)
    assert set(ratios) == set(cv.camera_array.posed_cameras)
    assert all(1.0 < r < 2.0 for r in ratios.values())  # ring scene: shallow depth range => refinement is gated off


def test_refine_calibration_stages_progress_and_outlier_rejection():
    sc, cv = _volume(outliers=0.01)  # well below the default per-camera 2.5 % filter, so stage 7 removes all of them
    seen = []
    run = refine_calibration(cv, refine_intrinsics=True, progress=lambda p, m: seen.append(p), _engine_factory=_OracleFactory())
    assert seen == [40, 55, 75, 90, 100]
    assert run.intrinsic_refinement_gated  # depth ratios < 2 on this scene (reference E4 negative control)
    out = run.capture_volume
    assert out.optimization_status.converged
    rms = out.compute_reprojection_report(_OracleFactory()).overall_rmse
    rms0 = cv.compute_reprojection_report(_OracleFactory()).overall_rmse
    assert rms < 1.0 < rms0, (rms, rms0)  # the 1 % gross outliers (10-50 px) are gone, noise floor ~0.65 px remains
    assert len(out.image_points) < len(cv.image_points)
    assert len(run.intrinsic_estimates) == len(cv.camera_array.posed_cameras)


def test_cancellation_between_stages():
    class Token:
        is_cancelled = True

    sc, cv = _volume(outliers=0.0, n_points=60, k=4)
    with pytest.raises(InterruptedError):
        refine_calibration(cv, cancellation_token=Token(), _engine_factory=_OracleFactory())


def test_static_marker_guard_drops_a_marker_that_moved():
    """Stage 4: the corners of 'static' marker 11 are triangulated as a badly non-rigid quadrilateral -> it is dropped with
    its observations, world points and constraints; marker 10 stays."""
    import pandas as pd

    from caliscope_amd.calibrate_extrinsics import apply_static_marker_guard
    from caliscope_amd.constraints import CentroidDistanceConstraint
    from tests.constrained_scene import marker_volume

    vol, _ = marker_volume(n_frames=4)
    same, none = apply_static_marker_guard(vol)
    assert same is vol and none == ()
    world = vol.world_points.df
    bad = (world["object_id"] == 11) & (world["keypoint_id"] == 2)
    world.loc[bad, "x_coord"] += 0.2  # 20 cm on a 12 cm marker
    from caliscope_amd.capture_volume import CaptureVolume
    from caliscope_amd.point_data import WorldPoints

    moved = CaptureVolume(vol.camera_array, vol.image_points, WorldPoints(world), vol.constraints)
    kept = vol.filter_by_absolute_error(0.5, min_per_camera=5, _engine_factory=_OracleFactory())  # static points: map hand-over
    assert len(kept.image_points) < len(vol.image_points) and np.array_equal(kept.img_to_obj_map, kept._compute_img_to_obj_map())
    out, dropped = apply_static_marker_guard(moved)
    assert dropped == (11,) and out.constraints.static_object_ids == frozenset({10, 12})
    assert 11 not in set(out.image_points.df["object_id"]) and 11 not in set(out.world_points.df["object_id"])
    assert all(11 not in (d.object_id_a, d.object_id_b) for d in out.constraints.distances)
    assert out.constraints.centroid_distances == ()  # the 11 -> 12 centre link went with marker 11
    assert len(out._build_constraint_arrays()[2]) < len(moved._build_constraint_arrays()[2])


def test_refresh_run_keeps_anchors_and_flags():
    from caliscope_amd.bundle_parameterization import IntrinsicEstimate
    from caliscope_amd.calibrate_extrinsics import CalibrationRun, refresh_run
    from tests.constrained_scene import marker_volume

    vol, _ = marker_volume(n_frames=3)
    prev = CalibrationRun(vol, tuple(IntrinsicEstimate(c, 1.0, 0.0, 0.0, 1400.0 + c, 0.1, -0.2) for c in vol.camera_array.cameras),
                          frozenset({2}), (11,), True)
    run = refresh_run(prev, vol)
    assert run.synthesized_cam_ids == frozenset({2}) and run.dropped_static_markers == (11,) and run.intrinsic_refinement_gated
    for e in run.intrinsic_estimates:
        cam = vol.camera_array.cameras[e.cam_id]
        assert (e.f_initial, e.k1_initial, e.k2_initial) == (1400.0 + e.cam_id, 0.1, -0.2)
        assert e.f_recovered == cam.matrix[0, 0] and e.k1_recovered == cam.distortions[0]


# -- the reference's entry point: guards + triangulation + stages 4-9 (reference tests/test_calibrate_extrinsics.py) ------------
def _oracle_triangulate(image_points, cameras, static_ids):
    import pandas as pd

    from caliscope_amd.point_data import WorldPoints
    from tests.test_triangulation import _oracle_world_points

    s, o, k, xyz = _oracle_world_points(cameras, image_points, set(static_ids), float32_io=True)
    return WorldPoints(pd.DataFrame({"sync_index": s, "object_id": o, "keypoint_id": k, "x_coord": xyz[:, 0], "y_coord": xyz[:, 1],
                                     "z_coord": xyz[:, 2]}))


def _board_session(strip=False, with_obj_loc=True):
    """Observations + posed cameras + the board's constraints, as ``calibrate_extrinsics`` receives them."""
    from copy import deepcopy

    from caliscope_amd.point_data import ImagePoints
    from tests.scenario_scenes import moving_board_volume

    vol, truth = moving_board_volume(constraints=True)
    df = vol.image_points.df
    if with_obj_loc:
        grid = np.array([[c * 0.04, r * 0.04, 0.0] for r in range(6) for c in range(9)])
        df[["obj_loc_x", "obj_loc_y", "obj_loc_z"]] = grid[df["keypoint_id"].to_numpy()]
    cameras = deepcopy(vol.camera_array)
    if strip:
        for cam in cameras.cameras.values():
            cam.matrix = cam.distortions = None
    return ImagePoints(df), cameras, vol.constraints, truth


def _engine_kwargs(kind):
    from tests.test_scenarios import _numpy_factory

    return dict(_engine_factory=_numpy_factory, _triangulate=_oracle_triangulate) if kind == "numpy" else {}


@pytest.mark.parametrize("kind", ["numpy", pytest.param("hip", marks=pytest.mark.gpu)])
def test_calibrate_extrinsics_entry_point_with_provided_intrinsics(kind):
    from caliscope_amd.calibrate_extrinsics import CalibrationRun, calibrate_extrinsics
    from tests.scenario_scenes import pose_errors

    image_points, cameras, constraints, truth = _board_session()
    before = {c: (cam.matrix.copy(), cam.rotation.copy()) for c, cam in cameras.cameras.items()}
    seen = []
    run = calibrate_extrinsics(image_points, cameras, constraints, progress=lambda p, m: seen.append(p), **_engine_kwargs(kind))
    assert isinstance(run, CalibrationRun) and run.synthesized_cam_ids == frozenset() and run.dropped_static_markers == ()
    assert seen == sorted(seen) and seen[0] == 5 and seen[-1] == 100
    assert run.capture_volume.optimization_status.converged and not run.intrinsic_refinement_gated
    for est in run.intrinsic_estimates:
        true = truth["cameras"].cameras[est.cam_id]
        assert abs(est.f_initial - true.matrix[0, 0]) < 1e-6 and abs(est.f_recovered - true.matrix[0, 0]) < 0.01 * true.matrix[0, 0]
    assert run.capture_volume.rigidity_report().rmse_mm < 1.0
    # the caller's cameras are untouched
    assert all(np.array_equal(cam.matrix, before[c][0]) and np.array_equal(cam.rotation, before[c][1]) for c, cam in cameras.cameras.items())
    vol = run.capture_volume
    if len(vol.world_points) == len(truth["points"]):
        trans, rot = pose_errors(vol, truth)
        assert trans < 0.01 and rot < 0.5, (trans, rot)


@pytest.mark.parametrize("kind", ["numpy", pytest.param("hip", marks=pytest.mark.gpu)])
def test_calibrate_extrinsics_blind_intrinsics(kind):
    """Cameras without intrinsics start from f = width / 2 and no distortion (the product workflow, reference
    TestEndToEndBlind): the pipeline converges, reports every camera as synthesised and recovers the focal length (3 % here:
    the blind model has p1 = p2 = k3 = 0, the true lens does not) and the board's rigidity."""
    from caliscope_amd.calibrate_extrinsics import calibrate_extrinsics

    image_points, cameras, constraints, truth = _board_session(strip=True)
    run = calibrate_extrinsics(image_points, cameras, constraints, **_engine_kwargs(kind))
    assert run.capture_volume.optimization_status.converged
    assert run.synthesized_cam_ids == frozenset(cameras.cameras)
    for est in run.intrinsic_estimates:
        f_true = truth["cameras"].cameras[est.cam_id].matrix[0, 0]
        assert est.f_initial == 960.0 and est.k1_initial == 0.0
        assert abs(est.f_recovered - f_true) < 0.03 * f_true, (est.cam_id, est.f_recovered)
    assert run.capture_volume.rigidity_report().rmse_mm < 2.0


def test_calibrate_extrinsics_input_guards():
    from copy import deepcopy

    from caliscope_amd.calibrate_extrinsics import calibrate_extrinsics
    from caliscope_amd.exceptions import CalibrationError

    image_points, cameras, constraints, _ = _board_session(strip=True, with_obj_loc=False)
    with pytest.raises(CalibrationError, match="Epipolar bootstrap requires calibrated intrinsics"):
        calibrate_extrinsics(image_points, cameras, constraints, _triangulate=_oracle_triangulate)

    class Token:
        is_cancelled = True

    image_points, cameras, constraints, _ = _board_session()
    with pytest.raises(InterruptedError):
        calibrate_extrinsics(image_points, cameras, constraints, cancellation_token=Token(), _triangulate=_oracle_triangulate)
    unposed = deepcopy(cameras)
    unposed.cameras[2].rotation = unposed.cameras[2].translation = None
    with pytest.raises(CalibrationError, match=r"Cameras \[2\] have observations but no pose"):
        calibrate_extrinsics(image_points, unposed, constraints, _triangulate=_oracle_triangulate)
    fewer = deepcopy(cameras)
    del fewer.cameras[3]
    with pytest.raises(CalibrationError, match="not in the CameraArray"):
        calibrate_extrinsics(image_points, fewer, constraints, _triangulate=_oracle_triangulate)


def test_two_sided_extraction_guard_and_cross_face_count():
    """Reference tests/test_calibrate_extrinsics.py:182-232."""
    import pandas as pd

    from caliscope_amd.calibrate_extrinsics import _count_firing_cross_face_rows, _validate_two_sided_extraction
    from caliscope_amd.constraints import DistanceConstraint
    from caliscope_amd.exceptions import CalibrationError
    from caliscope_amd.point_data import ImagePoints

    def points(object_ids, back_z=0.006):
        return ImagePoints(pd.DataFrame([dict(sync_index=0, cam_id=0, object_id=o, keypoint_id=k, img_loc_x=100.0, img_loc_y=100.0,
                                              obj_loc_x=0.05 * k, obj_loc_y=0.0, obj_loc_z=back_z if o == 1 else 0.0)
                                         for o in object_ids for k in (0, 1)]))

    _validate_two_sided_extraction(points([0]), thickness_m=0.0)
    _validate_two_sided_extraction(points([0, 1]), thickness_m=0.006)
    with pytest.raises(CalibrationError, match="no back-face observations"):
        _validate_two_sided_extraction(points([0]), thickness_m=0.006)
    with pytest.raises(CalibrationError, match="thickness is 0"):
        _validate_two_sided_extraction(points([0, 1]), thickness_m=0.0)
    with pytest.raises(CalibrationError, match="thickness changed"):
        _validate_two_sided_extraction(points([0, 1], back_z=0.006), thickness_m=0.012)

    cross = lambda a, b: DistanceConstraint(object_id_a=0, keypoint_id_a=a, object_id_b=1, keypoint_id_b=b, distance=0.006, sigma=0.0005)
    world = pd.DataFrame([dict(sync_index=5, object_id=0, keypoint_id=0), dict(sync_index=5, object_id=1, keypoint_id=0),
                          dict(sync_index=6, object_id=0, keypoint_id=1), dict(sync_index=7, object_id=1, keypoint_id=1)])
    assert _count_firing_cross_face_rows(world, (cross(0, 0), cross(1, 1))) == 1
    intra = DistanceConstraint(object_id_a=0, keypoint_id_a=0, object_id_b=0, keypoint_id_b=1, distance=0.05, sigma=0.002)
    assert _count_firing_cross_face_rows(world.iloc[:1], (intra,)) == 0


@pytest.mark.parametrize("kind,n_frames", [("numpy", 16), pytest.param("hip", 16, marks=pytest.mark.gpu)])
def test_thick_two_sided_board_through_the_pipeline(kind, n_frames):
    """Reference tests/synthetic/test_two_sided_charuco.py:41-87 (its tolerances): no camera ever sees both faces of the 6 mm
    board in one frame; the identity split (back face = object 1 at z = +t) with the cross-face ties and braces recovers the
    ground truth in the right basin (a reflected back face would be ~12 mm off), and beats the pre-thickness treatment
    (both faces fused into object 0 at z = 0) by more than 2x on the same footage."""
    from caliscope_amd.calibrate_extrinsics import calibrate_extrinsics
    from tests.scenario_scenes import keyed_errors, two_sided_board_session

    scene = dict(n_cams=8, radius=1.2, n_frames=n_frames, thickness=0.006)
    image_points, cameras, constraints, truth = two_sided_board_session(**scene)
    df = image_points.df
    assert df.groupby(["cam_id", "sync_index"])["object_id"].nunique().max() == 1
    per_face = df.groupby(["sync_index", "object_id"])["cam_id"].nunique().unstack(fill_value=0)
    assert ((per_face[0] >= 2) & (per_face[1] >= 2)).sum() >= n_frames // 2  # the cross-face rows do fire
    run = calibrate_extrinsics(image_points, cameras, constraints, refine_intrinsics=False, **_engine_kwargs(kind))
    assert run.capture_volume.optimization_status.converged
    trans, rot, rmse = keyed_errors(run.capture_volume, truth)
    # world points: 4 mm here (points seen by two close cameras only carry ~2 mm of depth noise at 0.5 px), still far from the
    # ~12 mm signature of the reflected basin
    assert rot < 0.2 and trans < 0.003 and rmse < 0.004, (trans, rot, rmse)

    fused_points, cameras, fused_constraints, _ = two_sided_board_session(fused=True, **scene)
    fused = calibrate_extrinsics(fused_points, cameras, fused_constraints, refine_intrinsics=False, **_engine_kwargs(kind))
    front = dict(cameras=truth["cameras"], points={(s, 0, k): v for (s, face, k), v in truth["points"].items() if face == 0})
    fused_trans, _, _ = keyed_errors(fused.capture_volume, front)
    assert trans < fused_trans / 2, (trans, fused_trans)


@pytest.mark.parametrize("kind", ["numpy", pytest.param("hip", marks=pytest.mark.gpu)])
def test_thin_footage_ends_in_the_same_local_minimum_as_scipy(kind):
    """The distance-only ties cannot tell the back face at +t from its reflection at -t (reference
    tests/synthetic/test_two_sided_charuco.py:8-12).  With only 10 frames and initial poses off by ~1 cm — more than the 6 mm
    thickness — the constrained problem has a second minimum at 4.5x the cost of the one next to the ground truth.  The host
    loop on the exact Schur step ends in the same one as the reference's scipy call (LSMR steps), after the same number of
    evaluations: parity of the trust-region logic where the landscape is not convex."""
    from caliscope_amd.bundle_parameterization import BundleParameterization
    from caliscope_amd.capture_volume import CaptureVolume
    from oracle.solver import optimize_scipy
    from tests.scenario_scenes import two_sided_board_session
    from tests.test_scenarios import _numpy_factory

    image_points, cameras, constraints, truth = two_sided_board_session(n_cams=8, radius=1.2, n_frames=10, thickness=0.006)
    vol = CaptureVolume.bootstrap(image_points, cameras, constraints, _triangulate=_oracle_triangulate)
    factory = _numpy_factory if kind == "numpy" else None  # None: the device path (cba_solve)
    got = vol.optimize(_engine_factory=factory).optimization_status
    near_truth = CaptureVolume.bootstrap(image_points, truth["cameras"], constraints, _triangulate=_oracle_triangulate)
    best = near_truth.optimize(_engine_factory=factory).optimization_status
    assert got.converged and best.converged and got.final_cost > 3.0 * best.final_cost
    _, cam, uv, obj = vol._matched_arrays()
    ga, gb, dist, sigma = vol._build_constraint_arrays()
    f_median = float(np.median([c.matrix[0, 0] for c in vol.camera_array.posed_cameras.values()]))
    par = BundleParameterization.from_camera_array(vol.camera_array, n_points=len(vol.world_points), refine_intrinsics=False)
    ref = optimize_scipy(par, cam, uv, obj, par.pack(vol.camera_array, vol.world_points.points),
                         constraints=(ga, gb, dist, (1.0 / f_median) / sigma))
    assert ref.status > 0 and abs(got.final_cost - ref.cost) <= 1e-6 * ref.cost, (got.final_cost, ref.cost)
    assert abs(got.iterations - ref.nfev) <= 2, (got.iterations, ref.nfev)


def test_a_static_marker_that_moved_is_dropped_and_the_driver_starts_again():
    """Marker 11 is declared static but was turned by 180 degrees half-way through the recording (its corner k is then seen where corner k + 2 was):
    pooled over all frames its corners triangulate to a collapsed, non-rigid quadrilateral.  The driver must decide that on the first
    triangulation, drop the marker, and — as the reference does (core/calibrate_extrinsics.py:146-203; trace pinned in
    tests/golden/reference_host/driver_03.npz) — start again: fresh copies of the caller's cameras, the 20 % progress mark, a second
    bootstrap without the marker's observations and constraints; then the three solver passes.  The REAL bootstrap and passes run here (oracle
    triangulation and numpy engine in the device's place)."""
    from caliscope_amd.calibrate_extrinsics import calibrate_extrinsics
    from caliscope_amd.point_data import ImagePoints
    from tests.constrained_scene import marker_volume

    vol, _ = marker_volume(n_frames=6)
    df = vol.image_points.df
    turned = (df["object_id"] == 11) & (df["sync_index"] >= 3)
    df.loc[turned, "keypoint_id"] = (df.loc[turned, "keypoint_id"] + 2) % 4
    calls, marks = [], []

    def counting_triangulate(image_points, cameras, static_ids):
        calls.append((sorted(int(o) for o in image_points.df["object_id"].unique()), sorted(static_ids)))
        return _oracle_triangulate(image_points, cameras, static_ids)

    kw = _engine_kwargs("numpy")
    kw["_triangulate"] = counting_triangulate
    run = calibrate_extrinsics(ImagePoints(df), vol.camera_array, vol.constraints, refine_intrinsics=False, progress=lambda p, m: marks.append((p, m)), **kw)
    assert run.dropped_static_markers == (11,)
    assert [p for p, _ in marks] == [5, 15, 20, 40, 55, 75, 90, 100] and marks[2][1] == "Re-bootstrapping after dropping markers"
    assert calls == [([0, 10, 11, 12], [10, 11, 12]), ([0, 10, 12], [10, 12])]  # triangulated twice: with the marker, then without it
    out = run.capture_volume
    assert 11 not in set(out.image_points.df["object_id"]) and 11 not in set(out.world_points.df["object_id"])
    assert out.constraints.static_object_ids == frozenset({10, 12}) and out.constraints.centroid_distances == ()
    assert all(11 not in (d.object_id_a, d.object_id_b) for d in out.constraints.distances)
    assert out.optimization_status.converged and out.rigidity_report().rmse_mm < 5.0
