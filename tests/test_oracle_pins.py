"""Pins the oracle against everything the reference's own tests hold for the path."""
import numpy as np
import pandas as pd
import pytest

from caliscope_amd.bundle_parameterization import BundleParameterization
from caliscope_amd.cameras import CameraArray, CameraData
from oracle import camera_model as cm
from oracle.residuals import joint_jacobian, joint_residuals, project_points, reprojection_errors
from oracle.scene import default_ring_scene_rows
from tests.helpers import small_problem


def test_golden_vector_default_ring_baseline(golden_dir):
    """cv2.projectPoints golden vector, atol 1e-10 (reference tests/synthetic/primitives/test_scene.py:641-664)."""
    rows, _, _ = default_ring_scene_rows(pixel_noise_sigma=0.5, random_seed=42)
    df = pd.DataFrame(rows, columns=["sync_index", "cam_id", "object_id", "keypoint_id", "img_loc_x", "img_loc_y"])
    df = df.sort_values(["sync_index", "cam_id", "object_id", "keypoint_id"]).reset_index(drop=True)
    gold = pd.read_csv(golden_dir / "default_ring_baseline" / "image_points_noisy.csv")
    assert len(df) == len(gold) == 2800
    assert np.array_equal(df["cam_id"].values, gold["cam_id"].values)
    assert np.array_equal(df["keypoint_id"].values, gold["keypoint_id"].values)
    assert np.allclose(df["img_loc_x"].values, gold["img_loc_x"].values, atol=1e-10, rtol=0)
    assert np.allclose(df["img_loc_y"].values, gold["img_loc_y"].values, atol=1e-10, rtol=0)


def _fd_jacobian(fun, x0, eps=1e-6):
    f0 = fun(x0)
    J = np.zeros((len(f0), len(x0)))
    for j in range(len(x0)):
        xp, xm = x0.copy(), x0.copy()
        xp[j] += eps
        xm[j] -= eps
        J[:, j] = (fun(xp) - fun(xm)) / (2 * eps)
    return J


def _assert_match(analytic, fd, tol=1e-6):
    """Per-column relative comparison, the reference's criterion (test_analytic_jacobian.py:37-50)."""
    diff = np.abs(analytic - fd)
    scale = np.maximum(np.abs(fd).max(axis=0), 1e-3)
    per_col = diff.max(axis=0) / scale
    assert per_col.max() < tol, f"column {int(per_col.argmax())}: rel err {per_col.max():.2e}"


@pytest.mark.parametrize("refine", [False, True])
def test_jacobian_matches_fd_pinhole(refine):
    sc, par, x0 = small_problem(n_cams=3, n_points=40, k=3, refine=refine)
    fun = lambda x: joint_residuals(x, par, sc.camera_indices, sc.image_coords, sc.obj_indices)
    J = joint_jacobian(x0, par, sc.camera_indices, sc.image_coords, sc.obj_indices).toarray()
    _assert_match(J, _fd_jacobian(fun, x0))


def _mixed_arrays():
    """The explicit numeric case of reference test_analytic_jacobian.py:115-174 (fisheye + free pinhole)."""
    rvec0, tvec0 = np.array([0.1, -0.05, 0.02]), np.array([0.0, 0.1, 3.0])
    K0 = np.array([[600.0, 0, 320], [0, 590.0, 240], [0, 0, 1]])
    dist0 = np.array([0.1, -0.05, 0.01, 0.002])
    rvec1, tvec1 = np.array([-0.08, 0.12, -0.04]), np.array([0.5, -0.1, 3.2])
    K1 = np.array([[610.0, 0, 315], [0, 605.0, 245], [0, 0, 1]])
    dist1 = np.array([0.08, -0.03, 0.001, -0.002, 0.005])
    ca = CameraArray({
        0: CameraData(cam_id=0, size=(640, 480), fisheye=True, matrix=K0, distortions=dist0,
                      rotation=cm.rodrigues(rvec0), translation=tvec0),
        1: CameraData(cam_id=1, size=(640, 480), matrix=K1, distortions=dist1,
                      rotation=cm.rodrigues(rvec1), translation=tvec1),
    })
    rng = np.random.default_rng(42)
    points = rng.uniform(-0.6, 0.6, (25, 3))
    exact = np.vstack([cm.project_fisheye(points, rvec0, tvec0, K0, dist0)[0],
                       cm.project_pinhole(points, rvec1, tvec1, K1, dist1)[0]])
    image_coords = exact + rng.normal(0, 0.5, exact.shape)
    cam_idx = np.repeat(np.array([0, 1], dtype=np.int16), len(points))
    obj_idx = np.tile(np.arange(len(points), dtype=np.int32), 2)
    return ca, points, image_coords, cam_idx, obj_idx


def test_jacobian_matches_fd_mixed_fisheye_pinhole():
    ca, points, image_coords, cam_idx, obj_idx = _mixed_arrays()
    par = BundleParameterization.from_camera_array(ca, n_points=len(points), refine_intrinsics=True)
    assert par.blocks[0].n_params == 6 and par.blocks[1].n_params == 9
    x0 = par.pack(ca, points)
    fun = lambda x: joint_residuals(x, par, cam_idx, image_coords, obj_idx)
    J = joint_jacobian(x0, par, cam_idx, image_coords, obj_idx).toarray()
    _assert_match(J, _fd_jacobian(fun, x0))


def test_constraint_rows_match_fd():
    """Corner (row repeated 4x) and centroid (4 distinct rows) endpoints, abs 1e-8 (reference :178-227)."""
    sc, par, x0 = small_problem(n_cams=3, n_points=40, k=3)
    ga = np.array([[0, 0, 0, 0], [0, 1, 2, 3]], dtype=np.int32)
    gb = np.array([[5, 5, 5, 5], [8, 9, 10, 11]], dtype=np.int32)
    dist, wts = np.array([0.11, 0.07]), np.array([2.0, 3.5])
    args = (par, sc.camera_indices, sc.image_coords, sc.obj_indices, ga, gb, dist, wts)
    J = joint_jacobian(x0, *args).toarray()
    fd = _fd_jacobian(lambda x: joint_residuals(x, *args), x0)
    n_rows = 2 * len(sc.camera_indices)
    assert J.shape[0] == n_rows + 2
    _assert_match(J, fd)
    assert np.abs(J[n_rows:] - fd[n_rows:]).max() < 1e-8
    assert np.abs(J[n_rows:]).max() > 0.1


def test_fisheye_differs_from_brown_conrady_and_rejects_5_coeffs():
    K = np.array([[600, 0, 320], [0, 600, 240], [0, 0, 1]], dtype=np.float64)
    D = np.array([0.1, -0.05, 0.01, 0.0])
    rvec, tvec = np.array([0.1, -0.05, 0.02]), np.array([0.0, 0.0, 5.0])
    pts = np.random.default_rng(42).uniform(-0.5, 0.5, (20, 3))
    pts[:, 2] += 3.0
    fe = project_points(pts, rvec, tvec, K, D, fisheye=True)
    bc = project_points(pts, rvec, tvec, K, np.array([0.1, -0.05, 0.01, 0.0, 0.0]), fisheye=False)
    assert not np.allclose(fe, bc, atol=0.1)
    with pytest.raises(ValueError, match="4 distortion coefficients"):
        project_points(np.zeros((1, 3)) + [0, 0, 1], np.zeros(3), np.zeros(3), np.eye(3) * 500, np.zeros(5), fisheye=True)


def test_fisheye_jacobian_fd_and_small_radius():
    K = np.array([[600, 0, 320], [0, 590, 240], [0, 0, 1]], dtype=np.float64)
    D = np.array([0.1, -0.05, 0.01, 0.002])
    pts = np.array([[0.0, 0.0, 2.0], [1e-10, 0.0, 2.0], [0.3, -0.2, 1.5], [-1.5, 1.0, 1.0]])
    rvec, tvec = np.array([0.02, -0.01, 0.03]), np.array([0.0, 0.0, 0.5])
    uv, J = cm.project_fisheye(pts, rvec, tvec, K, D, jacobian=True)
    assert np.all(np.isfinite(uv)) and np.all(np.isfinite(J))
    for col, vec in ((slice(8, 11), "r"), (slice(11, 14), "t")):
        for j in range(3):
            e = np.zeros(3)
            e[j] = 1e-6
            if vec == "r":
                fd = (cm.project_fisheye(pts, rvec + e, tvec, K, D)[0] - cm.project_fisheye(pts, rvec - e, tvec, K, D)[0]) / 2e-6
            else:
                fd = (cm.project_fisheye(pts, rvec, tvec + e, K, D)[0] - cm.project_fisheye(pts, rvec, tvec - e, K, D)[0]) / 2e-6
            assert np.allclose(J[:, col][:, j], fd.reshape(-1), rtol=1e-6, atol=1e-5)


def test_zero_residual_and_row_locality():
    """reference tests/test_reprojection_dispatch.py:57-131."""
    def cam(cid, tx):
        return CameraData(cam_id=cid, size=(640, 480), matrix=np.array([[500.0, 0, 320], [0, 500, 240], [0, 0, 1]]),
                          distortions=np.zeros(5), rotation=np.eye(3), translation=np.array([tx, 0.0, 0.0]))
    ca = CameraArray({0: cam(0, 0.0), 1: cam(1, 1.0)})
    par = BundleParameterization.from_camera_array(ca, n_points=3, refine_intrinsics=False)
    points = np.array([[0.0, 0.0, 5.0], [1.0, 0.0, 5.0], [-1.0, 1.0, 5.0]])
    uv, ci, oi = [], [], []
    for c in (0, 1):
        uv.append(cm.project_pinhole(points, np.zeros(3), ca[c].translation, ca[c].matrix, ca[c].distortions)[0])
        ci += [c] * 3
        oi += [0, 1, 2]
    uv = np.vstack(uv)
    ci, oi = np.array(ci, dtype=np.int16), np.array(oi, dtype=np.int32)
    x = par.pack(ca, points)
    r0 = joint_residuals(x, par, ci, uv, oi)
    np.testing.assert_allclose(r0, 0.0, atol=1e-10)
    xp = x.copy()
    xp[par.n_camera_params + 3] += 0.5  # point 1, x
    r1 = joint_residuals(xp, par, ci, uv, oi).reshape(-1, 2)
    changed = np.abs(r1).max(axis=1) > 1e-5
    assert np.array_equal(changed, oi == 1)


def test_rodrigues_round_trip_and_derivative():
    rng = np.random.default_rng(0)
    for _ in range(50):
        r = rng.normal(0, 1.0, 3)
        R = cm.rodrigues(r)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-13) and np.isclose(np.linalg.det(R), 1.0)
        r2 = cm.rotation_to_rvec(R)
        assert np.allclose(cm.rodrigues(r2), R, atol=1e-12)
        D = cm.rodrigues_jacobian(r)
        for j in range(3):
            e = np.zeros(3)
            e[j] = 1e-6
            fd = (cm.rodrigues(r + e) - cm.rodrigues(r - e)) / 2e-6
            assert np.allclose(D[j], fd, atol=1e-8)
    assert np.allclose(cm.rodrigues(np.zeros(3)), np.eye(3))
    # angle ~ pi branch
    for axis in (np.array([1.0, 0, 0]), np.array([0.6, -0.8, 0.0]), np.array([1.0, 2.0, -3.0]) / np.sqrt(14)):
        R = cm.rodrigues(axis * (np.pi - 1e-9))
        assert np.allclose(cm.rodrigues(cm.rotation_to_rvec(R)), R, atol=1e-7)


def test_post_optimization_session_reprojects_subpixel(golden_dir):
    """Real calibrated session (BASELINE.json configs[0]): conventions check against OpenCV-produced data.

    The reference only asserts 0 < RMSE < 10 px here (tests/test_reprojection_report.py:60)."""
    d = golden_dir / "post_optimization"
    ca = CameraArray.from_toml(d / "camera_array.toml")
    xy = pd.read_csv(d / "xy_CHARUCO.csv")
    xyz = pd.read_csv(d / "xyz_CHARUCO.csv")
    key = {k: i for i, k in enumerate(zip(xyz.sync_index, xyz.object_id, xyz.keypoint_id))}
    obj = np.array([key.get(k, -1) for k in zip(xy.sync_index, xy.object_id, xy.keypoint_id)])
    posed = ca.posed_cam_id_to_index
    keep = (obj >= 0) & xy.cam_id.isin(list(posed)).to_numpy()
    cam_idx = np.array([posed[c] for c in xy.cam_id[keep]], dtype=np.int16)
    uv = xy.loc[keep, ["img_loc_x", "img_loc_y"]].to_numpy()
    world = xyz[["x_coord", "y_coord", "z_coord"]].to_numpy()[obj[keep]]
    err = reprojection_errors(ca, cam_idx, uv, world)
    rmse = float(np.sqrt(np.mean(np.sum(err**2, axis=1))))
    assert keep.sum() == 2175
    # 1.66 px at the stored state (1.59 px at its BA optimum, see test_trf_driver.py): a wrong sign,
    # distortion order or rotation convention would show up as tens to hundreds of pixels.
    assert 0 < rmse < 2.0, rmse


# ---- constraint rows of joint_residuals / joint_jacobian: the reference's own known answers (tests/test_constraints.py) ----
def _one_camera_two_points(points):
    from caliscope_amd.cameras import CameraArray, CameraData

    cam = CameraData(cam_id=0, size=(400, 400), matrix=np.array([[200.0, 0, 200], [0, 200, 200], [0, 0, 1]]), distortions=np.zeros(5),
                     rotation=np.eye(3), translation=np.array([0.0, 0.0, 5.0]))
    ca = CameraArray({0: cam})
    par = BundleParameterization.from_camera_array(ca, n_points=len(points), refine_intrinsics=False)
    return par, par.pack(ca, np.asarray(points, dtype=np.float64))


def test_constraint_rows_follow_the_reprojection_rows_and_known_values():
    """Reference tests/test_constraints.py:599-649 (rows appended, zero at the exact distance), :770-805 (a corner endpoint is
    one index four times: the row equals the single-index expression EXACTLY), :808-885 (centroid rows vanish at exact geometry)."""
    cam_idx, uv, obj = np.array([0, 0], dtype=np.int16), np.array([[200.0, 200.0], [240.0, 200.0]]), np.array([0, 1], dtype=np.int32)
    par, x = _one_camera_two_points([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0]])
    ga, gb = np.array([[0, 0, 0, 0]], dtype=np.int32), np.array([[1, 1, 1, 1]], dtype=np.int32)
    r_plain = joint_residuals(x, par, cam_idx, uv, obj)
    r_con = joint_residuals(x, par, cam_idx, uv, obj, ga, gb, np.array([1.0]), np.array([0.5]))
    assert len(r_plain) == 4 and len(r_con) == 5 and np.array_equal(r_con[:4], r_plain) and abs(r_con[4]) < 1e-10
    pts = np.array([[0.1, 0.2, 0.3], [1.4, -0.5, 0.7]])
    par, x = _one_camera_two_points(pts)
    r = joint_residuals(x, par, cam_idx, uv, obj, ga, gb, np.array([1.0]), np.array([0.5]))
    assert r[-1] == (np.linalg.norm(pts[0] - pts[1]) - 1.0) * 0.5  # exact, not approximate
    # centroid endpoints: two unit squares 2 m apart
    sq = np.array([[-0.5, 0.5, 0], [0.5, 0.5, 0], [0.5, -0.5, 0], [-0.5, -0.5, 0.0]])
    par, x = _one_camera_two_points(np.vstack([sq, sq + [2.0, 0, 0]]))
    ga, gb = np.array([[0, 1, 2, 3]], dtype=np.int32), np.array([[4, 5, 6, 7]], dtype=np.int32)
    r = joint_residuals(x, par, cam_idx, uv, obj, ga, gb, np.array([2.0]), np.array([3.0]))
    assert abs(r[-1]) < 1e-12
    r = joint_residuals(x, par, cam_idx, uv, obj, ga, gb, np.array([1.5]), np.array([3.0]))
    assert r[-1] == pytest.approx(1.5)  # (2.0 - 1.5) * 3


def test_constraint_rows_touch_at_most_24_columns():
    """Reference tests/test_constraints.py:1014-1036: 3 coordinate columns per distinct endpoint point — 6 for a corner row, 24 for
    a centroid row — and nothing in the camera block."""
    par, x = _one_camera_two_points(np.random.default_rng(0).normal(size=(8, 3)) + [0, 0, 1.0])
    cam_idx, uv, obj = np.zeros(2, dtype=np.int16), np.array([[200.0, 200.0], [240.0, 200.0]]), np.array([0, 1], dtype=np.int32)
    ga = np.array([[0, 0, 0, 0], [0, 1, 2, 3]], dtype=np.int32)
    gb = np.array([[1, 1, 1, 1], [4, 5, 6, 7]], dtype=np.int32)
    J = joint_jacobian(x, par, cam_idx, uv, obj, ga, gb, np.array([1.0, 1.0]), np.array([1.0, 1.0])).toarray()
    rows = J[4:]
    assert rows.shape[0] == 2 and np.count_nonzero(rows[:, : par.n_camera_params]) == 0
    assert np.count_nonzero(rows[0]) == 6 and np.count_nonzero(rows[1]) == 24
    # row of the corner constraint: +unit on point 0, -unit on point 1
    pts = x[par.n_camera_params:].reshape(-1, 3)
    unit = (pts[0] - pts[1]) / np.linalg.norm(pts[0] - pts[1])
    assert np.allclose(rows[0, par.n_camera_params : par.n_camera_params + 3], unit) and np.allclose(rows[0, par.n_camera_params + 3 : par.n_camera_params + 6], -unit)


def test_sparsity_pattern_covers_the_jacobian_and_nothing_else():
    """Reference tests/synthetic/test_intrinsic_recovery.py:37-80 (sparsity oracle, variable-width blocks): every zero of
    ``BundleParameterization.sparsity`` is a true zero of the Jacobian; here also the converse on generic data (each marked
    entry is structurally reachable), with a locked fisheye camera next to a free pinhole one and both kinds of
    constraint rows."""
    ca, points, image_coords, cam_idx, obj_idx = _mixed_arrays()
    par = BundleParameterization.from_camera_array(ca, n_points=len(points), refine_intrinsics=True)
    x0 = par.pack(ca, points)
    ga = np.array([[0, 0, 0, 0], [1, 2, 3, 4]], dtype=np.int32)
    gb = np.array([[6, 6, 6, 6], [8, 9, 10, 11]], dtype=np.int32)
    args = (par, cam_idx, image_coords, obj_idx, ga, gb, np.array([0.3, 0.2]), np.array([1.5, 2.5]))
    J = joint_jacobian(x0, *args).toarray()
    pattern = par.sparsity(cam_idx, obj_idx, 2, ga, gb)
    from scipy.sparse import issparse

    assert issparse(pattern) and pattern.format == "lil" and pattern.shape == J.shape
    S = pattern.toarray()
    assert set(np.unique(S)) == {0, 1}
    assert np.abs(J[S == 0]).max() == 0.0
    fd = _fd_jacobian(lambda x: joint_residuals(x, *args), x0)
    assert np.abs(fd[S == 0]).max() < 1e-4  # the reference's bound on false zeros
    n_obs = len(cam_idx)
    per_row = S.sum(axis=1)
    width = np.array([b.n_params for b in par.blocks])[cam_idx]
    assert np.array_equal(per_row[: 2 * n_obs], np.repeat(width + 3, 2))
    assert list(per_row[2 * n_obs:]) == [6, 24]  # corner endpoints mark one point each, centroid endpoints four
    # without constraint arguments only the reprojection rows exist
    assert par.sparsity(cam_idx, obj_idx, 0, None, None).shape == (2 * n_obs, par.n_params)
