"""numpy implementation of the BAEngine protocol on top of the oracle.  TEST INFRASTRUCTURE.

Purpose: run the trust-region loop (``oracle/trf_driver.py``, the Python restatement of ``csrc/cba_solve.cpp``) on CPU so that its
logic can be compared with scipy's TRF without a GPU, and so that the point-sharded multi-rank
protocol (what is all-reduced and when) can be exercised with ``gloo`` at world_size 2.  It also
is the per-primitive oracle the HIP engine is compared with on the GPU (normal-equation blocks,
Newton step, trial cost).

It follows the same protocol as the HIP engine: camera parameters are replicated on every rank,
world points (with ALL their observations) are owned by exactly one rank, and only

  * the camera blocks ``U``, ``g_c``;  * the reduced camera system ``S``, ``b``;  * scalar sums

cross ranks (``allreduce``).  Robust-loss scaling uses scipy's own helpers
(``scipy.optimize._lsq``: ``construct_loss_function``, ``scale_for_robust_loss_function``), i.e. the
reference's exact arithmetic for that part.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np
from scipy.optimize._lsq.common import scale_for_robust_loss_function
from scipy.optimize._lsq.least_squares import construct_loss_function

from oracle.residuals import joint_jacobian, joint_residuals


@dataclass
class _Lin:
    g_norm_inf: float
    gh_sq: float
    jg_sq: float
    x_scaled_norm: float
    x_norm: float


@dataclass
class _Step:
    ok: bool
    p_sq: float
    gh_dot_p: float
    w_sq: float


@dataclass
class _Trial:
    cost: float
    step_norm: float
    finite: bool


class _LocalParam:
    """Duck-typed parameterization restricted to a subset of the world points."""

    def __init__(self, base, n_local_points):
        self._b = base
        self.blocks = base.blocks
        self.camera_param_offsets = base.camera_param_offsets
        self.n_camera_params = base.n_camera_params
        self.n_points = n_local_points

    def trial_projection_inputs(self, x, i):
        return self._b.trial_projection_inputs(x, i)


class OracleEngine:
    def __init__(self, parameterization, camera_indices, image_coords, obj_indices, loss="linear", f_scale=1.0,
                 owned_points=None, allreduce=None, constraints=None):
        """``owned_points``: sorted global ids of the points this rank owns (None = all)."""
        self.par = parameterization
        self.ncp = parameterization.n_camera_params
        P = parameterization.n_points
        self.P_global = P
        # owned_points=None: the arrays describe a problem that is already local to this rank (the product's
        # shard_problem did the re-indexing); current_x() then returns the local vector as the HIP engine does.
        self._gather_on_read = owned_points is not None
        self.owned = np.arange(P) if owned_points is None else np.asarray(owned_points, dtype=np.int64)
        g2l = -np.ones(P, dtype=np.int64)
        g2l[self.owned] = np.arange(self.owned.size)
        keep = g2l[np.asarray(obj_indices)] >= 0
        self.cam_idx = np.asarray(camera_indices)[keep]
        self.uv = np.asarray(image_coords, dtype=np.float64)[keep]
        self.obj = g2l[np.asarray(obj_indices)[keep]].astype(np.int32)
        self.lpar = _LocalParam(parameterization, self.owned.size)
        self.n_params = parameterization.n_camera_params + 3 * P
        self.loss = loss
        self.f_scale = f_scale
        # rigid-distance constraint rows (groups_a, groups_b, distances, weights), single-rank problems only
        self.constraints = tuple(constraints) if constraints is not None and constraints[0] is not None else None
        if self.constraints is not None and owned_points is not None:
            raise ValueError("constraint rows are not supported in sharded solves")
        n_con = 0 if self.constraints is None else len(self.constraints[0])
        self._loss_fn = construct_loss_function(2 * self.cam_idx.size + n_con, loss, f_scale)
        self._allreduce = allreduce if allreduce is not None else (lambda a: a)
        self.scale_inv = None
        self.x = None

    # ---- helpers -------------------------------------------------------------------------------
    def _sum(self, value: float) -> float:
        return float(self._allreduce(np.array([value], dtype=np.float64))[0])

    def _split(self, v):
        return v[: self.ncp], v[self.ncp :]

    def _norm_sq(self, v) -> float:
        c, p = self._split(v)
        return float(c @ c) + self._sum(float(p @ p))

    def _dot(self, a, b) -> float:
        ac, ap = self._split(a)
        bc, bp = self._split(b)
        return float(ac @ bc) + self._sum(float(ap @ bp))

    def _local_x(self, x_full):
        pts = x_full[self.ncp :].reshape(-1, 3)[self.owned].reshape(-1)
        return np.concatenate([x_full[: self.ncp], pts])

    def _con_args(self):
        return () if self.constraints is None else self.constraints

    def _residuals(self, xl):
        return joint_residuals(xl, self.lpar, self.cam_idx, self.uv, self.obj, *self._con_args())

    def _cost(self, f) -> float:
        if not np.all(np.isfinite(f)):
            return float("nan")
        local = 0.5 * float(f @ f) if self._loss_fn is None else float(self._loss_fn(f, cost_only=True))
        return self._sum(local)

    # ---- protocol ------------------------------------------------------------------------------
    def begin(self, x0):
        self.x_full_template = np.array(x0, dtype=np.float64)
        self.x = self._local_x(self.x_full_template)
        self.f_raw = self._residuals(self.x)
        self.scale_inv = None
        return self._cost(self.f_raw)

    def linearize(self):
        ncp = self.ncp
        f = self.f_raw.copy()
        J = joint_jacobian(self.x, self.lpar, self.cam_idx, self.uv, self.obj, *self._con_args()).tocsr()
        if self._loss_fn is not None:
            rho = self._loss_fn(f)
            J, f = scale_for_robust_loss_function(J, f, rho)
        self.J, self.f = J, f
        g = np.asarray(J.T @ f).ravel()
        g[:ncp] = self._allreduce(g[:ncp].copy())
        H = (J.T @ J).tocsr()
        U = H[:ncp, :ncp].toarray()
        U = self._allreduce(U)
        self.U, self.g = U, g
        self.W = H[:ncp, ncp:].tocsr()
        nloc = self.owned.size
        Hpp = H[ncp:, ncp:].tocsr()
        V = np.zeros((nloc, 3, 3))
        coo = Hpp.tocoo()
        blockdiag = coo.row // 3 == coo.col // 3  # all of it without constraint rows
        V[coo.row[blockdiag] // 3, coo.row[blockdiag] % 3, coo.col[blockdiag] % 3] = coo.data[blockdiag]
        assert self.constraints is not None or np.all(blockdiag)
        self.V = V
        self.H = H if self.constraints is not None else None  # constraint rows couple points: direct solve (newton_step)
        col_sq = np.concatenate([np.diag(U), V[:, [0, 1, 2], [0, 1, 2]].reshape(-1)])
        scale_inv = np.sqrt(col_sq)
        if self.scale_inv is None:
            scale_inv[scale_inv == 0] = 1.0
        else:
            scale_inv = np.maximum(scale_inv, self.scale_inv)
        self.scale_inv = scale_inv
        d = 1.0 / scale_inv
        g_h = d * g
        jg = J @ (d * g_h)
        return _Lin(
            g_norm_inf=float(self._allreduce_max(np.max(np.abs(g)) if g.size else 0.0)),
            gh_sq=self._norm_sq(g_h),
            jg_sq=self._sum(float(jg @ jg)),
            x_scaled_norm=float(np.sqrt(self._norm_sq(self.x * scale_inv))),
            x_norm=float(np.sqrt(self._norm_sq(self.x))),
        )

    def _allreduce_max(self, v: float) -> float:
        # max via sum is not possible; the protocol uses a dedicated max-reduction.  With the default
        # identity all-reduce this is the local value; the gloo harness overrides ``allreduce_max``.
        fn = getattr(self, "allreduce_max", None)
        return float(fn(v)) if fn is not None else float(v)

    def _finish_step(self, s):
        self.s = s
        p = s * self.scale_inv
        g_h = self.g / self.scale_inv
        gh_sq = self._norm_sq(g_h)
        ghp = self._dot(g_h, p)
        w = p - (ghp / gh_sq) * g_h
        return _Step(True, self._norm_sq(p), ghp, self._norm_sq(w))

    def newton_step(self, lam):
        ncp = self.ncp
        D2 = self.scale_inv**2
        if self.H is not None:
            # with constraint rows H_pp is not block diagonal: the same Schur complement with a sparse LU of the damped
            # H_pp in place of the 3 x 3 inverses (the device path's Woodbury correction is checked against this)
            import scipy.sparse as sp
            from scipy.sparse.linalg import splu

            g_c, g_p = self.g[:ncp], self.g[ncp:]
            Hpp = (self.H[ncp:, ncp:] + lam * sp.diags(D2[ncp:])).tolil()
            has_obs = np.repeat(np.bincount(self.obj, minlength=self.owned.size) > 0, 3) | (np.asarray(self.H[ncp:, ncp:].diagonal()) > 0)
            for k in np.flatnonzero(~has_obs):
                Hpp[k, k] = 1.0
            try:
                lu = splu(Hpp.tocsc())
            except RuntimeError:
                return _Step(False, 0.0, 0.0, 0.0)
            Wt = self.W.T.toarray()                      # (3 P_loc, ncp)
            X = lu.solve(Wt)                             # H_pp^-1 W^T
            S_loc = -(Wt.T @ X)
            b_loc = X.T @ g_p
            red = self._allreduce(np.concatenate([S_loc.reshape(-1), b_loc]))
            S = self.U + lam * np.diag(D2[:ncp]) + red[: ncp * ncp].reshape(ncp, ncp)
            rhs = -g_c + red[ncp * ncp :]
            try:
                L = np.linalg.cholesky(0.5 * (S + S.T))
            except np.linalg.LinAlgError:
                return _Step(False, 0.0, 0.0, 0.0)
            dc = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
            dp = -lu.solve(g_p + Wt @ dc)
            dp[~has_obs] = 0.0
            s = np.concatenate([dc, dp])
            if not np.all(np.isfinite(s)):
                return _Step(False, 0.0, 0.0, 0.0)
            return self._finish_step(s)
        g_c, g_p = self.g[:ncp], self.g[ncp:].reshape(-1, 3)
        Vd = self.V + lam * np.einsum("pi,ij->pij", D2[ncp:].reshape(-1, 3), np.eye(3))
        has_obs = np.bincount(self.obj, minlength=self.owned.size) > 0
        Vd[~has_obs] = np.eye(3)
        try:
            Vinv = np.linalg.inv(Vd)
        except np.linalg.LinAlgError:
            return _Step(False, 0.0, 0.0, 0.0)
        if not np.all(np.isfinite(Vinv)):
            return _Step(False, 0.0, 0.0, 0.0)
        Wd = self.W.toarray().reshape(ncp, -1, 3)  # (ncp, P_loc, 3)
        WVi = np.einsum("cpi,pij->cpj", Wd, Vinv)
        S_loc = -np.einsum("cpj,epj->ce", WVi, Wd)
        b_loc = np.einsum("cpj,pj->c", WVi, g_p)
        red = self._allreduce(np.concatenate([S_loc.reshape(-1), b_loc]))
        S = self.U + lam * np.diag(D2[:ncp]) + red[: ncp * ncp].reshape(ncp, ncp)
        rhs = -g_c + red[ncp * ncp :]
        try:
            L = np.linalg.cholesky(S)
        except np.linalg.LinAlgError:
            return _Step(False, 0.0, 0.0, 0.0)
        dc = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
        dp = -np.einsum("pij,pj->pi", Vinv, g_p + np.einsum("cpi,c->pi", Wd, dc))
        dp[~has_obs] = 0.0
        return self._finish_step(np.concatenate([dc, dp.reshape(-1)]))

    def subspace_gram(self, a1, b1, a2, b2):
        d2g = self.g / self.scale_inv**2
        j1 = self.J @ (a1 * d2g + b1 * self.s)
        j2 = self.J @ (a2 * d2g + b2 * self.s)
        return self._sum(float(j1 @ j1)), self._sum(float(j1 @ j2)), self._sum(float(j2 @ j2))

    def trial(self, alpha, beta):
        d = 1.0 / self.scale_inv
        step = alpha * d * d * self.g + beta * self.s
        self.x_new = self.x + step
        self.f_new = self._residuals(self.x_new)
        cost = self._cost(self.f_new)
        return _Trial(cost, float(np.sqrt(self._norm_sq(step))), bool(np.isfinite(cost)))

    def camera_params(self, which=0):
        return (self.x_new if which == 1 else self.x)[: self.ncp].copy()

    def accept(self):
        self.x = self.x_new
        self.f_raw = self.f_new

    def solve(self, x0, *, ftol=1e-8, xtol=1e-8, gtol=1e-8, max_nfev=None, verbose=0, lb=None, ub=None, fetch_x=True):
        """The engine-level ``solve()`` the product calls (``HipEngine.solve`` = ``cba_solve``): here the Python restatement of the driver
        (oracle/trf_driver.py) on this engine's primitives.  Bounds on the camera block: infeasible trial points are rejected (the CPU tests'
        bounded cases converge inside their boxes; the Coleman-Li variant lives in ``cba_solve`` and is tested on the device and on the
        dense CPU build of the ABI)."""
        from oracle.trf_driver import trf_solve

        feasible = None
        if lb is not None and ub is not None and (np.any(np.isfinite(lb)) or np.any(np.isfinite(ub))):
            lbc, ubc = np.asarray(lb, dtype=np.float64), np.asarray(ub, dtype=np.float64)

            def feasible(cam_params):
                return bool(np.all(cam_params > lbc) and np.all(cam_params < ubc))

        return trf_solve(self, x0, ftol=ftol, xtol=xtol, gtol=gtol, max_nfev=max_nfev, verbose=verbose, feasible=feasible)

    def current_x(self):
        """Full-layout x; non-owned points are filled by an all-reduce of zero-padded vectors."""
        if not self._gather_on_read:
            return self.x.copy()
        out = np.zeros(self.n_params)
        pts = np.zeros((self.P_global, 3))
        pts[self.owned] = self.x[self.ncp :].reshape(-1, 3)
        out[self.ncp :] = self._allreduce(pts.reshape(-1).copy())
        out[: self.ncp] = self.x[: self.ncp]
        return out
