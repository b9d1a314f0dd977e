"""Scenes for the scenario tests (tests/test_scenarios.py): the situations the reference's synthetic suite puts
``CaptureVolume.optimize`` in (tests/synthetic/test_intrinsic_recovery.py, test_outlier_robustness.py,
test_robust_loss.py, test_multistage_flow.py, test_large_ring.py), restated with this repo's generators.  The
reference bootstraps the initial poses with OpenCV's PnP (out of scope here); these scenes start from
perturbed ground truth instead, which leaves the optimum — what the assertions are about — unchanged."""

import numpy as np
import pandas as pd

from caliscope_amd.cameras import CameraArray, CameraData, matrix_to_rvec, rvec_to_matrix
from caliscope_amd.capture_volume import CaptureVolume
from caliscope_amd.constraints import ConstraintSet
from caliscope_amd.point_data import ImagePoints, WorldPoints
from caliscope_amd.synthetic import WEBCAM_SIZE, project_pinhole_bc5, ring_camera_array


def board_grid(rows, cols, spacing):
    grid = np.array([[c * spacing, r * spacing, 0.0] for r in range(rows) for c in range(cols)])
    return grid - grid.mean(axis=0)


def moving_board_volume(n_cams=4, radius=1.2, n_frames=40, rows=6, cols=9, spacing=0.04, start=(0.55, -0.55, 0.15),
                        end=(-0.55, 0.55, 0.85), tumble=2.0, stationary=False, f_scale=1.0, k1_delta=0.0, k2_delta=0.0,
                        noise_px=0.5, outliers=0.0, seed=42, constraints=False, perturb_poses=True):
    """A board carried (and tumbled) along a line through a camera ring: with the default path every camera sees it from
    about 0.5 m to 1.9 m and out to the image borders, which makes focal length observable next to the poses (the
    reference's ``intrinsic_perturbation_scene``); ``stationary`` keeps a small board in one far place (its negative
    control).  Returns ``(volume, truth)``; the volume's cameras carry the perturbed intrinsics and poses."""
    rng = np.random.default_rng(seed)
    cams = ring_camera_array(n_cams, radius=radius, target=(0.0, 0.0, 0.5))
    grid = board_grid(rows, cols, spacing)
    n_per = len(grid)
    axis = rng.normal(0, 1, 3)
    axis /= np.linalg.norm(axis)
    pts = []
    for f in range(n_frames):
        s = 0.0 if stationary or n_frames == 1 else f / (n_frames - 1)
        centre = (1 - s) * np.asarray(start) + s * np.asarray(end)
        R = rvec_to_matrix(axis * (0.3 + (0.0 if stationary else tumble * 2 * np.pi * s)))
        pts.append(grid @ R.T + centre)
    pts = np.vstack(pts)
    w, h = WEBCAM_SIZE
    cam_idx, uv, obj = [], [], []
    for c, cam in sorted(cams.cameras.items()):
        K = cam.matrix
        p, z = project_pinhole_bc5(pts, cam.rotation, cam.translation, K[0, 0], K[1, 1], K[0, 2], K[1, 2], cam.distortions)
        ok = (z > 0.1) & (p[:, 0] >= 0) & (p[:, 0] < w) & (p[:, 1] >= 0) & (p[:, 1] < h)
        cam_idx.append(np.full(ok.sum(), c))
        uv.append(p[ok])
        obj.append(np.flatnonzero(ok))
    cam_idx, uv, obj = np.concatenate(cam_idx).astype(np.int32), np.vstack(uv), np.concatenate(obj).astype(np.int32)
    uv = uv + rng.normal(0, noise_px, uv.shape)
    # keep points seen by at least two cameras
    seen = np.bincount(obj, minlength=len(pts))
    keep_pt = seen >= 2
    keep = keep_pt[obj]
    cam_idx, uv, obj = cam_idx[keep], uv[keep], obj[keep]
    outlier_rows = np.zeros(0, dtype=np.int64)
    if outliers > 0:
        orng = np.random.default_rng(seed + 2)
        outlier_rows = np.sort(orng.choice(len(obj), size=round(outliers * len(obj)), replace=False))
        mag, th = orng.uniform(10.0, 50.0, len(outlier_rows)), orng.uniform(0, 2 * np.pi, len(outlier_rows))
        uv[outlier_rows] += np.stack([mag * np.cos(th), mag * np.sin(th)], axis=1)
    init = {}
    for c, cam in cams.cameras.items():
        K, dist = cam.matrix.copy(), cam.distortions.copy()
        K[0, 0] *= f_scale
        K[1, 1] *= f_scale
        dist[0] += k1_delta
        dist[1] += k2_delta
        rvec = matrix_to_rvec(cam.rotation) + (rng.normal(0, 0.01, 3) if perturb_poses else 0)
        init[c] = CameraData(cam_id=c, size=cam.size, matrix=K, distortions=dist, rotation=rvec_to_matrix(rvec),
                             translation=cam.translation + (rng.normal(0, 0.01, 3) if perturb_poses else 0))
    pts0 = pts + rng.normal(0, 0.005, pts.shape)
    ids = np.flatnonzero(keep_pt)
    world = pd.DataFrame({"sync_index": ids // n_per, "object_id": 0, "keypoint_id": ids % n_per, "x_coord": pts0[ids, 0],
                          "y_coord": pts0[ids, 1], "z_coord": pts0[ids, 2], "frame_time": (ids // n_per) * 0.1})
    img = pd.DataFrame({"sync_index": obj // n_per, "cam_id": cam_idx, "object_id": 0, "keypoint_id": obj % n_per,
                        "img_loc_x": uv[:, 0], "img_loc_y": uv[:, 1]})
    cs = None
    if constraints:
        raw = np.array([[c * spacing, r * spacing, 0.0] for r in range(rows) for c in range(cols)], dtype=np.float32)
        cs = ConstraintSet.from_grid(raw, spacing)
    vol = CaptureVolume(CameraArray(init), ImagePoints(img), WorldPoints(world), cs)
    truth = dict(cameras=cams, points=pts[ids], outlier_rows=outlier_rows, n_per=n_per)
    return vol, truth


def pose_errors(volume, truth):
    """Worst camera position error (m) and rotation error (deg) against the ground truth after a similarity alignment of
    camera centres + points (what the reference's ``align_to_ground_truth`` / ``pose_error`` report,
    tests/synthetic/assertions.py:125-168)."""
    from caliscope_amd.bundle_parameterization import BundleParameterization
    from tests.helpers import camera_centres_and_rotations, umeyama

    par = BundleParameterization.from_camera_array(volume.camera_array, n_points=len(truth["points"]), refine_intrinsics=False)
    x_got = par.pack(volume.camera_array, volume.world_points.points)
    x_true = par.pack(truth["cameras"], truth["points"])
    ca, Ra = camera_centres_and_rotations(par, x_got)
    cb, Rb = camera_centres_and_rotations(par, x_true)
    pa, pb = x_got[par.n_camera_params:].reshape(-1, 3), x_true[par.n_camera_params:].reshape(-1, 3)
    s, R, t = umeyama(np.vstack([ca, pa]), np.vstack([cb, pb]))
    trans = float(np.linalg.norm(s * ca @ R.T + t - cb, axis=1).max())
    ang = 0.0
    for A, B in zip(Ra, Rb):
        rel = (A @ R.T) @ B.T
        w = np.array([rel[2, 1] - rel[1, 2], rel[0, 2] - rel[2, 0], rel[1, 0] - rel[0, 1]])
        ang = max(ang, float(np.arctan2(0.5 * np.linalg.norm(w), 0.5 * (np.trace(rel) - 1.0))))
    return trans, float(np.degrees(ang))


def chain_volume(n_cams=6, spacing=1.0, n_points=600, noise_px=0.5, seed=42):
    """Cameras in a row looking the same way; every point is seen by one pair of neighbours only, so the co-visibility
    graph is a chain (tridiagonal coverage, the reference's ``chain_scene``): errors accumulate towards the ends and the
    reduced camera system is block-tridiagonal."""
    from caliscope_amd.synthetic import WEBCAM_DIST, WEBCAM_FOCAL

    rng = np.random.default_rng(seed)
    w, h = WEBCAM_SIZE
    K = np.array([[WEBCAM_FOCAL, 0, w / 2.0], [0, WEBCAM_FOCAL, h / 2.0], [0, 0, 1.0]])
    R = np.array([[1.0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]])  # camera z = world y (looking along +y), image y = world -z
    cams = {c: CameraData(cam_id=c, size=WEBCAM_SIZE, matrix=K.copy(), distortions=np.array(WEBCAM_DIST), rotation=R.copy(),
                          translation=-R @ np.array([c * spacing, 0.0, 0.0])) for c in range(n_cams)}
    truth_cams = CameraArray(cams)
    per_pair = n_points // (n_cams - 1)
    pts, cam_idx, uv, obj = [], [], [], []
    for pair in range(n_cams - 1):
        got = 0
        while got < per_pair:
            X = np.c_[rng.uniform(pair * spacing - 0.3, (pair + 1) * spacing + 0.3, 64), rng.uniform(2.0, 4.0, 64), rng.uniform(-0.7, 0.7, 64)]
            ok = np.ones(len(X), dtype=bool)
            proj = []
            for c in (pair, pair + 1):
                cam = cams[c]
                p, z = project_pinhole_bc5(X, cam.rotation, cam.translation, K[0, 0], K[1, 1], K[0, 2], K[1, 2], cam.distortions)
                ok &= (z > 0.1) & (p[:, 0] >= 0) & (p[:, 0] < w) & (p[:, 1] >= 0) & (p[:, 1] < h)
                proj.append(p)
            for i in np.flatnonzero(ok)[: per_pair - got]:
                for c, p in zip((pair, pair + 1), proj):
                    cam_idx.append(c)
                    uv.append(p[i])
                    obj.append(len(pts))
                pts.append(X[i])
                got += 1
    pts, uv = np.array(pts), np.array(uv) + rng.normal(0, noise_px, (len(uv), 2))
    init = CameraArray({c: CameraData(cam_id=c, size=cam.size, matrix=cam.matrix.copy(), distortions=cam.distortions.copy(),
                                      rotation=rvec_to_matrix(matrix_to_rvec(cam.rotation) + rng.normal(0, 0.01, 3)),
                                      translation=cam.translation + rng.normal(0, 0.02, 3)) for c, cam in cams.items()})
    vol = CaptureVolume.from_arrays(init, np.array(cam_idx, dtype=np.int32), uv, np.array(obj, dtype=np.int32),
                                    pts + rng.normal(0, 0.01, pts.shape))
    return vol, dict(cameras=truth_cams, points=pts)


def two_sided_board_session(n_cams=6, radius=2.0, n_frames=30, rows=4, cols=6, spacing=0.05, thickness=0.006, noise_px=0.5, seed=42,
                            fused=False):
    """A thick board filmed from all around: a camera on the +z side of the board sees the back face (object 1, corners at
    board z = +thickness), a camera on the -z side the front face (object 0, z = 0) — never both in one frame (the reference's
    ``two_sided_charuco_scene``).  ``fused`` relabels everything as object 0 at z = 0, the pre-thickness treatment.
    Returns ``(image_points, cameras_init, constraints, truth)`` as ``calibrate_extrinsics`` takes them."""
    rng = np.random.default_rng(seed)
    cams = ring_camera_array(n_cams, radius=radius, target=(0.0, 0.0, 0.5))
    grid = np.array([[c * spacing, r * spacing, 0.0] for r in range(rows) for c in range(cols)])
    centre_off = grid.mean(axis=0)
    w, h = WEBCAM_SIZE
    rows_out, truth_pts = [], {}
    for f in range(n_frames):
        s = f / max(n_frames - 1, 1)
        # the board stands upright and turns about the vertical axis while drifting through the volume
        yaw = 2 * np.pi * 1.5 * s + 0.3
        R = rvec_to_matrix(np.array([0.0, 0.0, yaw])) @ rvec_to_matrix(np.array([np.pi / 2 + 0.2 * np.sin(5 * s), 0.0, 0.0]))
        centre = np.array([0.3 * np.cos(3 * s), 0.3 * np.sin(2 * s), 0.5 + 0.1 * np.sin(4 * s)])
        normal = R[:, 2]
        faces = {0: (grid - centre_off) @ R.T + centre, 1: (grid - centre_off + [0, 0, thickness]) @ R.T + centre}
        for c, cam in sorted(cams.cameras.items()):
            cam_centre = -cam.rotation.T @ cam.translation
            view = cam_centre - centre
            cosang = float(normal @ view / np.linalg.norm(view))
            if abs(cosang) < 0.25:
                continue  # grazing view: the tracker would not detect the board
            face = 1 if cosang > 0 else 0
            X = faces[face]
            K = cam.matrix
            p, z = project_pinhole_bc5(X, cam.rotation, cam.translation, K[0, 0], K[1, 1], K[0, 2], K[1, 2], cam.distortions)
            ok = (z > 0.1) & (p[:, 0] >= 0) & (p[:, 0] < w) & (p[:, 1] >= 0) & (p[:, 1] < h)
            p = p + rng.normal(0, noise_px, p.shape)
            for k in np.flatnonzero(ok):
                rows_out.append(dict(sync_index=f, cam_id=c, object_id=0 if fused else face, keypoint_id=int(k), img_loc_x=p[k, 0], img_loc_y=p[k, 1],
                                     obj_loc_x=grid[k, 0], obj_loc_y=grid[k, 1], obj_loc_z=0.0 if fused or face == 0 else thickness))
                truth_pts[(f, face, int(k))] = X[k]
    init = CameraArray({c: CameraData(cam_id=c, size=cam.size, matrix=cam.matrix.copy(), distortions=cam.distortions.copy(),
                                      rotation=rvec_to_matrix(matrix_to_rvec(cam.rotation) + rng.normal(0, 0.01, 3)),
                                      translation=cam.translation + rng.normal(0, 0.01, 3)) for c, cam in cams.cameras.items()})
    cs = ConstraintSet.from_grid(grid.astype(np.float32), spacing, thickness_m=0.0 if fused else thickness)
    return ImagePoints(pd.DataFrame(rows_out)), init, cs, dict(cameras=cams, points=truth_pts)


def keyed_errors(volume, truth):
    """Worst camera position error (m), rotation error (deg) and world-point RMSE (m) against a ground truth keyed by
    ``(sync_index, object_id, keypoint_id)``, after a similarity alignment on cameras + matched points."""
    from caliscope_amd.bundle_parameterization import BundleParameterization
    from tests.helpers import camera_centres_and_rotations, umeyama

    df = volume.world_points.df
    keys = list(zip(df["sync_index"].tolist(), df["object_id"].tolist(), df["keypoint_id"].tolist()))
    has = np.array([k in truth["points"] for k in keys])
    got_pts = df[["x_coord", "y_coord", "z_coord"]].to_numpy()[has]
    true_pts = np.array([truth["points"][k] for k, ok in zip(keys, has) if ok])
    par = BundleParameterization.from_camera_array(volume.camera_array, n_points=1, refine_intrinsics=False)
    ca, Ra = camera_centres_and_rotations(par, par.pack(volume.camera_array, np.zeros((1, 3))))
    cb, Rb = camera_centres_and_rotations(par, par.pack(truth["cameras"], np.zeros((1, 3))))
    s, R, t = umeyama(np.vstack([ca, got_pts]), np.vstack([cb, true_pts]))
    trans = float(np.linalg.norm(s * ca @ R.T + t - cb, axis=1).max())
    ang = 0.0
    for A, B in zip(Ra, Rb):
        rel = (A @ R.T) @ B.T
        wv = np.array([rel[2, 1] - rel[1, 2], rel[0, 2] - rel[2, 0], rel[1, 0] - rel[0, 1]])
        ang = max(ang, float(np.arctan2(0.5 * np.linalg.norm(wv), 0.5 * (np.trace(rel) - 1.0))))
    rmse = float(np.sqrt(np.mean(np.sum((s * got_pts @ R.T + t - true_pts) ** 2, axis=1))))
    return trans, float(np.degrees(ang)), rmse
