"""The reference's solver call, verbatim, on the oracle callables.  TEST INFRASTRUCTURE / CPU baseline.

``scipy.optimize.least_squares(joint_residuals, x0, args=(...8 items...), jac=joint_jacobian,
x_scale="jac", loss=..., f_scale=..., ftol=..., max_nfev=..., method="trf", bounds=...)`` exactly as
``core/capture_volume.py:387-411`` issues it (scipy 1.15.3 = the reference's pinned version).
"""

from __future__ import annotations

import numpy as np
from scipy.optimize import least_squares

from oracle.residuals import joint_jacobian, joint_residuals, reprojection_errors  # noqa: F401


def optimize_scipy(parameterization, camera_indices, image_coords, obj_indices, x0, *, ftol=1e-8, xtol=1e-8,
                   gtol=1e-8, max_nfev=None, loss="linear", f_scale=1.0, verbose=0, tr_options=None, constraints=None, tr_solver=None):
    """``constraints`` = (groups_a, groups_b, distances, weights) or None: the four trailing ``args`` of the reference call.
    ``tr_solver`` None is the reference's call (scipy then picks 'lsmr' for the sparse Jacobian: inexact steps that crawl
    along weakly determined directions); tests that need the fully converged point pass 'exact' (dense SVD steps)."""
    kw = {}
    jac = joint_jacobian
    if tr_solver is not None:
        kw["tr_solver"] = tr_solver
        if tr_solver == "exact":  # scipy accepts 'exact' with a dense Jacobian only
            jac = lambda *a: joint_jacobian(*a).toarray()  # noqa: E731
    if tr_options is not None:
        kw["tr_options"] = tr_options
    return least_squares(
        joint_residuals,
        x0,
        args=(parameterization, camera_indices, image_coords, obj_indices, *(constraints if constraints is not None else (None, None, None, None))),
        jac=jac,
        verbose=verbose,
        x_scale="jac",
        loss=loss,
        f_scale=f_scale,
        ftol=ftol,
        xtol=xtol,
        gtol=gtol,
        max_nfev=max_nfev,
        method="trf",
        bounds=parameterization.bounds(),
        **kw,
    )


def rms_reprojection_px(parameterization, camera_indices, image_coords, obj_indices, x) -> float:
    """sqrt(mean(ex^2 + ey^2)) in pixels at parameter vector x (reference capture_volume.py:183,197)."""
    r = joint_residuals(x, parameterization, camera_indices, image_coords, obj_indices).reshape(-1, 2)
    fx0 = np.array([b.fx_initial for b in parameterization.blocks])[camera_indices]
    e = r * fx0[:, None]
    return float(np.sqrt(np.mean(np.sum(e * e, axis=1))))
