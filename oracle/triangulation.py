"""CPU restatement of the step before the BA path: undistort + batched DLT triangulation.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  What the reference does with OpenCV and numpy:

* ``CameraData.undistort_points(points, output="normalized")`` (``cameras/camera_array.py:135-174``): the points are cast
  to **float32**, ``cv2.undistortPoints`` (pinhole + Brown-Conrady) or ``cv2.fisheye.undistortPoints`` is called with
  ``P = I``; the result comes back as float32.  OpenCV is absent here, so the two functions are restated from their
  published algorithm (OpenCV 4.x ``cvUndistortPointsInternal``: 5 fixed-point iterations of
  ``x <- (x0 - delta(x)) / cdist(x)``; ``fisheye::undistortPoints``: Newton on ``theta_d = theta (1 + k1 theta^2 + ...)``,
  at most 10 steps, stop below 1e-8, ``theta_d`` clipped to [-pi/2, pi/2]).  **Parity unpinned**: the reference has no
  golden vector for either; they are validated by the round trip ``distort(undistort(p)) == p`` against the pinned
  projection functions of ``oracle/camera_model.py``.
* ``CameraData.normalized_projection_matrix`` = ``[R | t]`` (3 x 4, identity intrinsics).
* ``triangulate_image_points`` (``core/point_data.py:121-229``): per 3-D point the 2k x 4 DLT matrix with rows
  ``x P[2] - P[0]``, ``y P[2] - P[1]`` in ascending camera order, ``np.linalg.svd``, last right-singular vector,
  dehomogenised.  Points seen by fewer than two cameras are dropped.
"""

from __future__ import annotations

import numpy as np


def undistort_pinhole(points_px, K, dist, *, float32_io: bool = True, iterations: int = 5) -> np.ndarray:
    """cv2.undistortPoints(points, K, dist, P=I) for the 5-coefficient Brown-Conrady model."""
    pts = np.asarray(points_px, dtype=np.float32 if float32_io else np.float64).astype(np.float64).reshape(-1, 2)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    k1, k2, p1, p2, k3 = (list(np.asarray(dist, dtype=np.float64).ravel()) + [0.0] * 5)[:5]
    x0 = (pts[:, 0] - cx) / fx
    y0 = (pts[:, 1] - cy) / fy
    x, y = x0.copy(), y0.copy()
    for _ in range(iterations):
        r2 = x * x + y * y
        icdist = 1.0 / (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2)
        dx = 2.0 * p1 * x * y + p2 * (r2 + 2.0 * x * x)
        dy = p1 * (r2 + 2.0 * y * y) + 2.0 * p2 * x * y
        x = (x0 - dx) * icdist
        y = (y0 - dy) * icdist
    out = np.stack([x, y], axis=1)
    return out.astype(np.float32).astype(np.float64) if float32_io else out


def undistort_fisheye(points_px, K, dist, *, float32_io: bool = True) -> np.ndarray:
    """cv2.fisheye.undistortPoints(points, K, D, P=I) for the 4-coefficient equidistant model."""
    pts = np.asarray(points_px, dtype=np.float32 if float32_io else np.float64).astype(np.float64).reshape(-1, 2)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    k = (list(np.asarray(dist, dtype=np.float64).ravel()) + [0.0] * 4)[:4]
    px = (pts[:, 0] - cx) / fx
    py = (pts[:, 1] - cy) / fy
    theta_d = np.clip(np.sqrt(px * px + py * py), -np.pi / 2, np.pi / 2)
    theta = theta_d.copy()
    active = theta_d > 1e-8
    for _ in range(10):
        t2 = theta * theta
        t4, t6, t8 = t2 * t2, t2 * t2 * t2, t2 * t2 * t2 * t2
        fix = (theta * (1 + k[0] * t2 + k[1] * t4 + k[2] * t6 + k[3] * t8) - theta_d) / (
            1 + 3 * k[0] * t2 + 5 * k[1] * t4 + 7 * k[2] * t6 + 9 * k[3] * t8
        )
        theta = np.where(active, theta - fix, theta)
        active = active & (np.abs(fix) >= 1e-8)
    scale = np.where(theta_d > 1e-8, np.tan(theta) / np.where(theta_d > 1e-8, theta_d, 1.0), 1.0)
    out = np.stack([px * scale, py * scale], axis=1)
    return out.astype(np.float32).astype(np.float64) if float32_io else out


def undistort_points(points_px, K, dist, fisheye: bool, *, float32_io: bool = True) -> np.ndarray:
    return (undistort_fisheye if fisheye else undistort_pinhole)(points_px, K, dist, float32_io=float32_io)


def normalized_projection_matrix(R, t) -> np.ndarray:
    return np.hstack([np.asarray(R, dtype=np.float64), np.asarray(t, dtype=np.float64).reshape(3, 1)])


def triangulate_point(P_list, xy) -> np.ndarray:
    """One point: rows ``x P[2] - P[0]``, ``y P[2] - P[1]`` per view, smallest right-singular vector."""
    rows = []
    for P, (x, y) in zip(P_list, xy):
        rows.append(x * P[2] - P[0])
        rows.append(y * P[2] - P[1])
    _, _, vh = np.linalg.svd(np.asarray(rows), full_matrices=False)
    xyzw = vh[-1]
    return xyzw[:3] / xyzw[3]


def triangulate_image_points(projection_matrices, sync_indices, camera_ids, object_ids, keypoint_ids, img_xy):
    """Same contract as the reference function: returns (sync_indices, object_ids, keypoint_ids, xyz) of the points seen
    by at least two cameras; row order = ascending (sync_index, object_id, keypoint_id) (the reference returns them
    grouped by camera set; callers treat the result as a keyed table)."""
    sync_indices, camera_ids = np.asarray(sync_indices), np.asarray(camera_ids)
    object_ids, keypoint_ids = np.asarray(object_ids), np.asarray(keypoint_ids)
    img_xy = np.asarray(img_xy, dtype=np.float64).reshape(-1, 2)
    if len(keypoint_ids) < 2:
        e = np.array([], dtype=np.int64)
        return e, e.copy(), e.copy(), np.zeros((0, 3))
    order = np.lexsort((camera_ids, keypoint_ids, object_ids, sync_indices))
    s, o, k, c, xy = sync_indices[order], object_ids[order], keypoint_ids[order], camera_ids[order], img_xy[order]
    brk = np.flatnonzero((np.diff(s) != 0) | (np.diff(o) != 0) | (np.diff(k) != 0)) + 1
    starts = np.concatenate([[0], brk, [len(s)]])
    out = []
    for a, b in zip(starts[:-1], starts[1:]):
        if b - a < 2:
            continue
        out.append((s[a], o[a], k[a], triangulate_point([projection_matrices[int(ci)] for ci in c[a:b]], xy[a:b])))
    if not out:
        e = np.array([], dtype=np.int64)
        return e, e.copy(), e.copy(), np.zeros((0, 3))
    return (np.array([r[0] for r in out], dtype=np.int64), np.array([r[1] for r in out], dtype=np.int64),
            np.array([r[2] for r in out], dtype=np.int64), np.vstack([r[3] for r in out]))
