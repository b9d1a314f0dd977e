"""numpy restatement of the three OpenCV functions on the hot path.  TEST INFRASTRUCTURE.

The reference calls ``cv2.Rodrigues``, ``cv2.projectPoints`` and
``cv2.fisheye.projectPoints`` (reference ``core/reprojection.py:28,31,172,175,186``;
``cameras/camera_array.py:121,132``).  OpenCV is absent from this image, so the
published camera model is written out here.  The
Jacobian arrays use OpenCV's column layout so that :mod:`oracle.reprojection`
can slice them exactly as the reference slices cv2's output
(``core/reprojection.py:122-125,179-183``).

Conventions (``cameras/camera_array.py:38-39,115-133``): ``X_cam = R @ X_world + t``.
"""

from __future__ import annotations

import numpy as np

_EPS = np.finfo(np.float64).eps


def _skew(v: np.ndarray) -> np.ndarray:
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def rodrigues(rvec: np.ndarray) -> np.ndarray:
    """Axis-angle vector -> 3x3 rotation (``cv2.Rodrigues(rvec)[0]``; Appendix A.1)."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    theta = float(np.linalg.norm(r))
    if theta < _EPS:
        return np.eye(3)
    k = r / theta
    c, s = np.cos(theta), np.sin(theta)
    return c * np.eye(3) + (1.0 - c) * np.outer(k, k) + s * _skew(k)


def rodrigues_jacobian(rvec: np.ndarray) -> np.ndarray:
    """dR/dr as an array ``D[j] = dR/dr_j`` (3,3,3).

    Closed form (Gallego & Yezzi 2015): ``dR/dr_j = (r_j [r]x + [r x (I-R) e_j]x) R / theta^2``;
    at theta -> 0 the generators ``[e_j]x``.  cv2 returns the same numbers as a 3x9 block;
    only its action on a point enters the projection Jacobian.
    """
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    theta2 = float(r @ r)
    out = np.zeros((3, 3, 3))
    if theta2 < _EPS * _EPS:
        for j in range(3):
            e = np.zeros(3)
            e[j] = 1.0
            out[j] = _skew(e)
        return out
    R = rodrigues(r)
    rx = _skew(r)
    I = np.eye(3)
    for j in range(3):
        e = np.zeros(3)
        e[j] = 1.0
        out[j] = (r[j] * rx + _skew(np.cross(r, (I - R) @ e))) @ R / theta2
    return out


def rotation_to_rvec(R: np.ndarray) -> np.ndarray:
    """3x3 rotation -> axis-angle (``cv2.Rodrigues(R)[0]``; ``camera_array.py:121``).

    Appendix A.1 matrix->vector branch structure.  OpenCV first re-orthonormalises
    R by SVD; that is a no-op for valid rotations and is reproduced for safety.
    """
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt(v @ v * 0.25)
    c = np.clip((np.trace(R) - 1.0) * 0.5, -1.0, 1.0)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = (R[0, 0] + 1) * 0.5
        rx = np.sqrt(max(t, 0.0))
        t = (R[1, 1] + 1) * 0.5
        ry = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        t = (R[2, 2] + 1) * 0.5
        rz = np.sqrt(max(t, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(rx) < abs(ry) and abs(rx) < abs(rz) and (R[1, 2] > 0) != (ry * rz > 0):
            rz = -rz
        out = np.array([rx, ry, rz])
        out *= theta / np.linalg.norm(out)
        return out
    return v * (0.5 / s) * theta


def _camera_frame(points_world: np.ndarray, rvec: np.ndarray, tvec: np.ndarray):
    X = np.asarray(points_world, dtype=np.float64).reshape(-1, 3)
    R = rodrigues(rvec)
    t = np.asarray(tvec, dtype=np.float64).reshape(3)
    Y = X @ R.T  # R X
    return X, R, Y, Y + t


def _extrinsic_chain(X, R, dRdr, dpdXc):
    """Chain d(pixel)/dX_cam (n,2,3) to rvec / tvec / world-point derivatives."""
    # dXc/dr_j = dR/dr_j @ X
    dXc_dr = np.einsum("jab,nb->naj", dRdr, X)  # (n,3,3): [n, a, j]
    d_rvec = np.einsum("nia,naj->nij", dpdXc, dXc_dr)
    d_tvec = dpdXc
    return d_rvec, d_tvec


def project_pinhole(points_world, rvec, tvec, K, dist, jacobian: bool = False):
    """``cv2.projectPoints`` for the 5-coefficient Brown-Conrady model (Appendix A.2).

    Returns pixels (n,2) and, if requested, the (2n, 15) Jacobian in cv2's column
    layout ``[rvec 0:3 | tvec 3:6 | fx 6 | fy 7 | cx 8 | cy 9 | k1 k2 p1 p2 k3 10:15]``
    with rows interleaved (u0, v0, u1, v1, ...).
    """
    K = np.asarray(K, dtype=np.float64)
    d = np.asarray(dist, dtype=np.float64).ravel()
    if d.shape[0] != 5:
        raise ValueError(f"pinhole model expects 5 distortion coefficients, got {d.shape[0]}")
    k1, k2, p1, p2, k3 = d
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    X, R, _, Xc = _camera_frame(points_world, rvec, tvec)
    iz = 1.0 / Xc[:, 2]
    x = Xc[:, 0] * iz
    y = Xc[:, 1] * iz
    r2 = x * x + y * y
    r4 = r2 * r2
    r6 = r4 * r2
    cdist = 1.0 + k1 * r2 + k2 * r4 + k3 * r6
    a1 = 2.0 * x * y
    a2 = r2 + 2.0 * x * x
    a3 = r2 + 2.0 * y * y
    xd = x * cdist + p1 * a1 + p2 * a2
    yd = y * cdist + p1 * a3 + p2 * a1
    uv = np.stack([fx * xd + cx, fy * yd + cy], axis=1)
    if not jacobian:
        return uv, None

    n = X.shape[0]
    dcd = k1 + 2.0 * k2 * r2 + 3.0 * k3 * r4  # d cdist / d r2
    dxd_dx = cdist + 2.0 * x * x * dcd + 2.0 * p1 * y + 6.0 * p2 * x
    dxd_dy = 2.0 * x * y * dcd + 2.0 * p1 * x + 2.0 * p2 * y
    dyd_dx = 2.0 * x * y * dcd + 2.0 * p1 * x + 2.0 * p2 * y
    dyd_dy = cdist + 2.0 * y * y * dcd + 6.0 * p1 * y + 2.0 * p2 * x
    # d(x,y)/dXc
    dxy = np.zeros((n, 2, 3))
    dxy[:, 0, 0] = iz
    dxy[:, 0, 2] = -x * iz
    dxy[:, 1, 1] = iz
    dxy[:, 1, 2] = -y * iz
    dd = np.zeros((n, 2, 2))
    dd[:, 0, 0] = fx * dxd_dx
    dd[:, 0, 1] = fx * dxd_dy
    dd[:, 1, 0] = fy * dyd_dx
    dd[:, 1, 1] = fy * dyd_dy
    dpdXc = np.einsum("nij,njk->nik", dd, dxy)
    d_rvec, d_tvec = _extrinsic_chain(X, R, rodrigues_jacobian(rvec), dpdXc)

    J = np.zeros((n, 2, 15))
    J[:, :, 0:3] = d_rvec
    J[:, :, 3:6] = d_tvec
    J[:, 0, 6] = xd
    J[:, 1, 7] = yd
    J[:, 0, 8] = 1.0
    J[:, 1, 9] = 1.0
    J[:, 0, 10] = fx * x * r2
    J[:, 1, 10] = fy * y * r2
    J[:, 0, 11] = fx * x * r4
    J[:, 1, 11] = fy * y * r4
    J[:, 0, 12] = fx * a1
    J[:, 1, 12] = fy * a3
    J[:, 0, 13] = fx * a2
    J[:, 1, 13] = fy * a1
    J[:, 0, 14] = fx * x * r6
    J[:, 1, 14] = fy * y * r6
    return uv, J.reshape(2 * n, 15)


def project_fisheye(points_world, rvec, tvec, K, dist, jacobian: bool = False):
    """``cv2.fisheye.projectPoints`` (equidistant, 4 coefficients, alpha=0; Appendix A.3).

    Jacobian columns follow cv2: ``[fx 0, fy 1, cx 2, cy 3, k1..k4 4:8, rvec 8:11, tvec 11:14, alpha 14]``.
    """
    K = np.asarray(K, dtype=np.float64)
    d = np.asarray(dist, dtype=np.float64).ravel()
    if d.shape[0] != 4:
        raise ValueError(f"Fisheye projection requires 4 distortion coefficients, got {d.shape[0]}")
    k1, k2, k3, k4 = d
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    X, R, _, Xc = _camera_frame(points_world, rvec, tvec)
    iz = 1.0 / Xc[:, 2]
    x = Xc[:, 0] * iz
    y = Xc[:, 1] * iz
    r2 = x * x + y * y
    r = np.sqrt(r2)
    th = np.arctan(r)
    th2 = th * th
    th3, th5, th7, th9 = th * th2, th * th2**2, th * th2**3, th * th2**4
    thd = th + k1 * th3 + k2 * th5 + k3 * th7 + k4 * th9
    big = r > 1e-8
    inv_r = np.where(big, 1.0 / np.where(big, r, 1.0), 1.0)
    cdist = np.where(big, thd * inv_r, 1.0)
    xd = x * cdist
    yd = y * cdist
    uv = np.stack([fx * xd + cx, fy * yd + cy], axis=1)
    if not jacobian:
        return uv, None

    n = X.shape[0]
    dthd_dth = 1.0 + 3.0 * k1 * th2 + 5.0 * k2 * th2**2 + 7.0 * k3 * th2**3 + 9.0 * k4 * th2**4
    dth_dr = 1.0 / (1.0 + r2)
    # d cdist / dr  (zero where r is tiny, as cv2 does)
    dcd_dr = np.where(big, (dthd_dth * dth_dr * r - thd) * inv_r * inv_r, 0.0)
    dr_dx = np.where(big, x * inv_r, 0.0)
    dr_dy = np.where(big, y * inv_r, 0.0)
    dxd_dx = cdist + x * dcd_dr * dr_dx
    dxd_dy = x * dcd_dr * dr_dy
    dyd_dx = y * dcd_dr * dr_dx
    dyd_dy = cdist + y * dcd_dr * dr_dy
    dxy = np.zeros((n, 2, 3))
    dxy[:, 0, 0] = iz
    dxy[:, 0, 2] = -x * iz
    dxy[:, 1, 1] = iz
    dxy[:, 1, 2] = -y * iz
    dd = np.zeros((n, 2, 2))
    dd[:, 0, 0] = fx * dxd_dx
    dd[:, 0, 1] = fx * dxd_dy
    dd[:, 1, 0] = fy * dyd_dx
    dd[:, 1, 1] = fy * dyd_dy
    dpdXc = np.einsum("nij,njk->nik", dd, dxy)
    d_rvec, d_tvec = _extrinsic_chain(X, R, rodrigues_jacobian(rvec), dpdXc)

    J = np.zeros((n, 2, 15))
    J[:, 0, 0] = xd
    J[:, 1, 1] = yd
    J[:, 0, 2] = 1.0
    J[:, 1, 3] = 1.0
    for j, thp in enumerate((th3, th5, th7, th9)):
        dk = np.where(big, thp * inv_r, 0.0)
        J[:, 0, 4 + j] = fx * x * dk
        J[:, 1, 4 + j] = fy * y * dk
    J[:, :, 8:11] = d_rvec
    J[:, :, 11:14] = d_tvec
    J[:, 0, 14] = fx * yd  # d/d alpha (unused; alpha is fixed at 0)
    return uv, J.reshape(2 * n, 15)
