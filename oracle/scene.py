"""Restatement of the reference's pinned synthetic scene.  TEST INFRASTRUCTURE.

Reproduces ``default_ring_scene(pixel_noise_sigma=0.5, random_seed=42)``
(reference ``synthetic/scene_factories.py:20-46``) so that the oracle's pinhole
projection can be checked against the reference's golden vector
``tests/fixtures/synthetic/default_ring_baseline/image_points_noisy.csv`` — which was produced
by the real ``cv2.projectPoints`` — to 1e-10 px (the tolerance the reference itself uses,
``tests/synthetic/primitives/test_scene.py:641-664``).

Pieces restated (file:line in the reference):
  ring cameras        synthetic/camera_synthesizer.py:134-199, 278-318 (t = -R @ position, :304)
  look-at pose        synthetic/se3_pose.py:107-157   (rows = right, down, forward)
  WEBCAM lens         synthetic/camera_synthesizer.py:23-27, 92-97
  planar grid         synthetic/calibration_object.py:57-103
  orbital trajectory  synthetic/trajectory.py:76-147   (origin frame made identity)
  projection + noise  synthetic/synthetic_scene.py:151-230 (noise drawn BEFORE masking, :184-188)
"""

from __future__ import annotations

import numpy as np

from oracle.camera_model import project_pinhole, rotation_to_rvec

WEBCAM_F = 1394.6
WEBCAM_DIST = np.array([0.115, -0.219, 0.0012, 0.0086, 0.113])
WEBCAM_SIZE = (1920, 1080)


def look_at_rotation(position, target, up=(0.0, 0.0, 1.0)) -> np.ndarray:
    fwd = np.asarray(target, float) - np.asarray(position, float)
    fwd = fwd / np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, float))
    right = right / np.linalg.norm(right)
    down = np.cross(fwd, right)
    down = down / np.linalg.norm(down)
    return np.vstack([right, down, fwd])


def ring_cameras(n: int, radius: float, height: float, angular_offset_deg: float = 0.0):
    """List of (R, t, K, dist, size) for an inward-facing ring."""
    w, h = WEBCAM_SIZE
    K = np.array([[WEBCAM_F, 0, w / 2.0], [0, WEBCAM_F, h / 2.0], [0, 0, 1]], dtype=np.float64)
    cams = []
    for i in range(n):
        ang = 2 * np.pi * i / n + np.radians(angular_offset_deg)
        pos = np.array([radius * np.cos(ang), radius * np.sin(ang), height])
        R = look_at_rotation(pos, np.array([0.0, 0.0, height]))
        cams.append((R, -R @ pos, K.copy(), WEBCAM_DIST.copy(), WEBCAM_SIZE))
    return cams


def planar_grid(rows: int, cols: int, spacing: float) -> np.ndarray:
    pts = np.zeros((rows * cols, 3))
    for r in range(rows):
        for c in range(cols):
            pts[r * cols + c] = (c * spacing, r * spacing, 0.0)
    return pts


def _pose_matrix(R, t):
    m = np.eye(4)
    m[:3, :3] = R
    m[:3, 3] = t
    return m


def orbital_poses(n_frames: int, radius: float, arc_extent_deg: float = 360.0, height: float = 0.0,
                  tumble_rate: float = 0.0, origin_frame: int = 0):
    """List of 4x4 object->world poses; the origin frame's pose is the identity."""
    arc = np.radians(arc_extent_deg)
    raw = []
    for i in range(n_frames):
        t = i / n_frames if arc_extent_deg >= 360.0 else i / max(n_frames - 1, 1)
        ang = arc * t
        tumble = 2 * np.pi * tumble_rate * t
        c, s = np.cos(tumble), np.sin(tumble)
        Kz = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 0.0]])
        R = np.eye(3) + s * Kz + (1 - c) * (Kz @ Kz)
        raw.append(_pose_matrix(R, np.array([radius * np.cos(ang), radius * np.sin(ang), height])))
    o = raw[origin_frame]
    o_inv = _pose_matrix(o[:3, :3].T, -o[:3, :3].T @ o[:3, 3])
    # reference: origin_inv.compose(p) == p.matrix @ origin_inv.matrix  (se3_pose.py:171-182)
    return [p @ o_inv for p in raw]


def default_ring_scene_rows(pixel_noise_sigma: float = 0.5, random_seed: int = 42):
    """Rows (sync_index, cam_id, object_id, keypoint_id, img_loc_x, img_loc_y) + GT structures."""
    cams = ring_cameras(4, radius=2.0, height=0.5)
    grid = planar_grid(5, 7, 0.05)
    poses = orbital_poses(20, radius=0.2, arc_extent_deg=360.0, tumble_rate=1.0)
    rng = np.random.default_rng(random_seed)
    rows = []
    world_by_frame = []
    for frame, pose in enumerate(poses):
        world = grid @ pose[:3, :3].T + pose[:3, 3]
        world_by_frame.append(world)
        for cam_id, (R, t, K, dist, (w, h)) in enumerate(cams):
            uv, _ = project_pinhole(world, rotation_to_rvec(R), t, K, dist)
            uv = uv + rng.normal(0, pixel_noise_sigma, uv.shape)
            depth = (world @ R.T + t)[:, 2]
            for kp in range(len(grid)):
                if depth[kp] > 0 and 0 <= uv[kp, 0] < w and 0 <= uv[kp, 1] < h:
                    rows.append((frame, cam_id, 0, kp, float(uv[kp, 0]), float(uv[kp, 1])))
    return rows, cams, np.asarray(world_by_frame)
