"""Shared test helpers: gauge (similarity) alignment and problem builders.

BA has a 7-DoF gauge freedom that the reference never fixes (reference
tests/synthetic/test_alignment_gauge.py:19-44), so converged poses are compared after a similarity
alignment — the Umeyama algorithm the reference itself uses (core/alignment.py:84-150; applied in
tests/synthetic/assertions.py:125-168).
"""
from __future__ import annotations

import numpy as np

from caliscope_amd.bundle_parameterization import BundleParameterization
from caliscope_amd.cameras import rvec_to_matrix
from caliscope_amd.synthetic import make_scene


def umeyama(src: np.ndarray, dst: np.ndarray):
    """Least-squares similarity (s, R, t) with dst ~ s R src + t."""
    mu_s, mu_d = src.mean(0), dst.mean(0)
    a, b = src - mu_s, dst - mu_d
    cov = b.T @ a / len(src)
    U, sv, Vt = np.linalg.svd(cov)
    sign = np.ones(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        sign[2] = -1
    R = U @ np.diag(sign) @ Vt
    s = float((sv * sign).sum() / (a * a).sum() * len(src))
    return s, R, mu_d - s * R @ mu_s


def camera_centres_and_rotations(par, x):
    cen, rot = [], []
    for off in par.camera_param_offsets:
        R = rvec_to_matrix(x[off : off + 3])
        rot.append(R)
        cen.append(-R.T @ x[off + 3 : off + 6])
    return np.array(cen), np.array(rot)


def aligned_difference(par, x_a, x_b):
    """Align solution a onto b (camera centres + points); return (max rel position diff, max rotation angle [rad], scale)."""
    ca, Ra = camera_centres_and_rotations(par, x_a)
    cb, Rb = camera_centres_and_rotations(par, x_b)
    pa = x_a[par.n_camera_params :].reshape(-1, 3)
    pb = x_b[par.n_camera_params :].reshape(-1, 3)
    s, R, t = umeyama(np.vstack([ca, pa]), np.vstack([cb, pb]))
    ca2 = s * ca @ R.T + t
    pa2 = s * pa @ R.T + t
    extent = np.abs(np.vstack([cb, pb])).max()
    pos = max(np.abs(ca2 - cb).max(), np.abs(pa2 - pb).max()) / extent
    ang = 0.0
    for A, B in zip(Ra, Rb):
        # world->cam rotation of a in b's gauge: A R^T
        rel = (A @ R.T) @ B.T
        w = np.array([rel[2, 1] - rel[1, 2], rel[0, 2] - rel[2, 0], rel[1, 0] - rel[0, 1]])
        # atan2 form: arccos((tr-1)/2) alone has a sqrt(eps) ~ 1.5e-8 rad noise floor near zero
        ang = max(ang, float(np.arctan2(0.5 * np.linalg.norm(w), 0.5 * (np.trace(rel) - 1.0))))
    return pos, ang, s


def small_problem(n_cams=6, n_points=300, k=6, refine=False, loss="linear", outliers=0.0, seed=42):
    sc = make_scene(n_cams=n_cams, n_points=n_points, n_obs=n_points * k, refine=refine, loss=loss, outliers=outliers, seed=seed)
    par = BundleParameterization.from_camera_array(sc.cameras_init, n_points=n_points, refine_intrinsics=refine)
    x0 = par.pack(sc.cameras_init, sc.points_init)
    return sc, par, x0
