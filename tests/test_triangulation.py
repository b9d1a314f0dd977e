"""Undistortion + DLT triangulation: oracle properties on CPU, device parity on the GPU."""

import numpy as np
import pandas as pd
import pytest

from caliscope_amd.cameras import CameraArray, CameraData
from caliscope_amd.point_data import STATIC_SYNC_INDEX, ImagePoints
from caliscope_amd.synthetic import WEBCAM_DIST, ring_camera_array
from oracle import triangulation as otri
from oracle.camera_model import project_fisheye, project_pinhole
from oracle.camera_model import rotation_to_rvec

FISHEYE_DIST = np.array([0.05, -0.01, 0.003, -0.001])


def _scene(n_cams=6, n_points=120, n_frames=3, fisheye_every=3, seed=5, noise_px=0.0):
    """Ring of cameras (every third one a fisheye), a few frames of a moving board's corners + two static markers."""
    rng = np.random.default_rng(seed)
    cams = ring_camera_array(n_cams)
    for c, cam in cams.cameras.items():
        if fisheye_every and c % fisheye_every == 0:
            cam.fisheye = True
            cam.distortions = FISHEYE_DIST.copy()
    rows, truth = [], {}
    for f in range(n_frames):
        pts = np.c_[rng.uniform(-0.4, 0.4, n_points), rng.uniform(-0.4, 0.4, n_points), rng.uniform(0.2, 1.0, n_points)]
        for k, X in enumerate(pts):
            truth[(f, 0, k)] = X
    statics = {(STATIC_SYNC_INDEX, 7, k): np.array([0.1 * k, -0.2, 0.5]) for k in range(2)}
    for (f, o, k), X in list(truth.items()) + [((f, 7, k), X) for f in range(n_frames) for (_, _, k), X in statics.items()]:
        seen = rng.choice(n_cams, size=rng.integers(1, n_cams + 1), replace=False)  # some points get a single view
        for c in seen:
            cam = cams.cameras[int(c)]
            rvec = rotation_to_rvec(cam.rotation)
            proj = project_fisheye if cam.fisheye else project_pinhole
            uv = proj(X[None, :], rvec, cam.translation, cam.matrix, cam.distortions)
            uv = (uv[0] if isinstance(uv, tuple) else uv).reshape(2) + rng.normal(0, noise_px, 2)
            rows.append(dict(sync_index=f, cam_id=int(c), object_id=o, keypoint_id=k, img_loc_x=uv[0], img_loc_y=uv[1], frame_time=0.1 * f))
    truth.update(statics)
    return cams, ImagePoints(pd.DataFrame(rows)), truth


def _oracle_world_points(cams, ip, static_ids, float32_io):
    df = ip.df
    und = np.zeros((len(df), 2))
    for c, cam in cams.cameras.items():
        m = (df["cam_id"] == c).to_numpy()
        if m.any():
            und[m] = otri.undistort_points(df.loc[m, ["img_loc_x", "img_loc_y"]].to_numpy(), cam.matrix, cam.distortions, cam.fisheye, float32_io=float32_io)
    sync = df["sync_index"].to_numpy().copy()
    sync[np.isin(df["object_id"].to_numpy(), list(static_ids))] = STATIC_SYNC_INDEX
    P = {c: otri.normalized_projection_matrix(cam.rotation, cam.translation) for c, cam in cams.cameras.items()}
    return otri.triangulate_image_points(P, sync, df["cam_id"].to_numpy(), df["object_id"].to_numpy(), df["keypoint_id"].to_numpy(), und)


def test_oracle_undistort_round_trip():
    """distort(undistort(p)) == p against the pinned projection functions (the reference has no golden for cv2's undistort)."""
    rng = np.random.default_rng(0)
    K = np.array([[1394.6, 0, 960.0], [0, 1394.6, 540.0], [0, 0, 1.0]])
    X = np.c_[rng.uniform(-0.5, 0.5, 300), rng.uniform(-0.3, 0.3, 300), rng.uniform(2, 4, 300)]
    for proj, und, dist in ((project_pinhole, otri.undistort_pinhole, np.array(WEBCAM_DIST)), (project_fisheye, otri.undistort_fisheye, FISHEYE_DIST)):
        uv = proj(X, np.zeros(3), np.zeros(3), K, dist)
        uv = uv[0] if isinstance(uv, tuple) else uv
        n = und(uv, K, dist, float32_io=False)
        assert np.abs(n - X[:, :2] / X[:, 2:]).max() < 1e-10  # OpenCV's 5 fixed-point iterations reach ~1e-11 here
        n32 = und(uv, K, dist, float32_io=True)
        assert np.abs(n32 - n).max() < 2e-6  # float32 in/out, as the reference calls cv2


def test_oracle_triangulation_recovers_truth():
    cams, ip, truth = _scene(noise_px=0.0)
    s, o, k, xyz = _oracle_world_points(cams, ip, {7}, float32_io=False)
    assert len(s) > 250 and (s == STATIC_SYNC_INDEX).sum() == 2  # single-view points are dropped
    err = max(np.abs(xyz[i] - truth[(int(s[i]), int(o[i]), int(k[i]))]).max() for i in range(len(s)))
    assert err < 1e-8


def test_exports_and_signature():
    from caliscope_amd import _lib

    assert "cba_triangulate" in _lib.SIGNATURES
    lib = _lib.load()
    assert hasattr(lib, "cba_triangulate")


@pytest.mark.gpu
@pytest.mark.parametrize("float32_io", [True, False])
def test_device_triangulation_matches_oracle(float32_io):
    from caliscope_amd.triangulation import triangulate

    cams, ip, truth = _scene(noise_px=0.3)
    wp = triangulate(ip, cams, static_object_ids=frozenset({7}), float32_io=float32_io).df
    s, o, k, xyz = _oracle_world_points(cams, ip, {7}, float32_io=float32_io)
    ref = pd.DataFrame({"sync_index": s, "object_id": o, "keypoint_id": k, "x": xyz[:, 0], "y": xyz[:, 1], "z": xyz[:, 2]})
    m = wp.merge(ref, on=["sync_index", "object_id", "keypoint_id"], how="outer", indicator=True)
    assert (m["_merge"] == "both").all() and len(m) == len(ref)  # same set of points (>= 2 views), none lost, none invented
    d = np.abs(m[["x_coord", "y_coord", "z_coord"]].to_numpy() - m[["x", "y", "z"]].to_numpy()).max()
    # eigenvector of A^T A (device) vs SVD of A (reference): 1e-9 m on metre-scale scenes with 0.3 px noise
    assert d < 1e-8, d
    stat = wp[wp["sync_index"] == STATIC_SYNC_INDEX]
    assert len(stat) == 2 and stat["frame_time"].isna().all()


@pytest.mark.gpu
def test_device_triangulate_image_points_signature():
    """The reference function's own contract: undistorted normalised input, dict of projection matrices."""
    from caliscope_amd.triangulation import triangulate_image_points, undistort_points

    cams, ip, truth = _scene(noise_px=0.0, fisheye_every=2)
    df = ip.df
    und = np.zeros((len(df), 2))
    for c, cam in cams.cameras.items():
        m = (df["cam_id"] == c).to_numpy()
        und[m] = undistort_points(cam, df.loc[m, ["img_loc_x", "img_loc_y"]].to_numpy(), output="normalized")
        ref = otri.undistort_points(df.loc[m, ["img_loc_x", "img_loc_y"]].to_numpy(), cam.matrix, cam.distortions, cam.fisheye)
        assert np.abs(und[m] - ref).max() < 1e-12  # same algorithm, same float32 rounding points
    s, o, k, xyz = triangulate_image_points(cams.normalized_projection_matrices, df["sync_index"].to_numpy(), df["cam_id"].to_numpy(),
                                            df["object_id"].to_numpy(), df["keypoint_id"].to_numpy(), und)
    keep = o == 0
    err = max(np.abs(xyz[i] - truth[(int(s[i]), 0, int(k[i]))]).max() for i in np.flatnonzero(keep))
    assert err < 5e-6  # float32-rounded normalised coordinates: ~1e-7 relative, metres
    px = undistort_points(cams.cameras[1], df.loc[df["cam_id"] == 1, ["img_loc_x", "img_loc_y"]].to_numpy(), output="pixels")
    assert px.shape[1] == 2 and np.isfinite(px).all()


@pytest.mark.gpu
def test_real_session_triangulation_and_unmatched_tracking(golden_dir):
    """The reference's session through ImagePoints.triangulate (reference tests/test_capture_volume.py:30-76: structure, no NaN, metre
    scale) against the oracle, then the volume built from it: every observation of a triangulated point is matched, the rest
    are counted per camera (tests/test_reprojection_report.py:92-123)."""
    from caliscope_amd.cameras import CameraArray
    from caliscope_amd.capture_volume import CaptureVolume
    from caliscope_amd.point_data import ImagePoints

    d = golden_dir / "post_optimization"
    cams = CameraArray.from_toml(d / "camera_array.toml")
    ip = ImagePoints.from_csv(d / "xy_CHARUCO.csv")
    wp = ip.triangulate(cams)
    df = wp.df
    assert len(df) > 0 and not df[["x_coord", "y_coord", "z_coord"]].isna().any().any() and df[["x_coord", "y_coord", "z_coord"]].abs().max().max() < 10.0
    s, o, k, xyz = _oracle_world_points(cams, ip, set(), float32_io=True)
    ref = pd.DataFrame({"sync_index": s, "object_id": o, "keypoint_id": k, "x": xyz[:, 0], "y": xyz[:, 1], "z": xyz[:, 2]})
    m = df.merge(ref, on=["sync_index", "object_id", "keypoint_id"], how="outer", indicator=True)
    assert (m["_merge"] == "both").all() and len(m) == len(ref)
    assert np.abs(m[["x_coord", "y_coord", "z_coord"]].to_numpy() - m[["x", "y", "z"]].to_numpy()).max() < 1e-7
    # the stored world points of the session are the same points after the reference's bundle adjustment: centimetres apart
    stored = pd.read_csv(d / "xyz_CHARUCO.csv")
    both = df.merge(stored, on=["sync_index", "object_id", "keypoint_id"], suffixes=("", "_ba"))
    assert len(both) > 0.9 * len(stored)
    assert np.median(np.linalg.norm(both[["x_coord", "y_coord", "z_coord"]].to_numpy() - both[["x_coord_ba", "y_coord_ba", "z_coord_ba"]].to_numpy(), axis=1)) < 0.05
    vol = CaptureVolume(cams, ip, wp)
    rep = vol.reprojection_report
    img = vol.image_points.df
    for cam_id in cams.cameras:
        here = (img["cam_id"] == cam_id).to_numpy()
        assert rep.unmatched_by_camera[cam_id] == int(here.sum() - (here & (vol.img_to_obj_map >= 0)).sum())
    assert rep.n_unmatched_observations == len(img) - rep.n_observations_matched and 0 < rep.overall_rmse < 10.0
    opt = vol.optimize()
    assert opt.optimization_status.converged and opt.reprojection_report.overall_rmse < rep.overall_rmse
