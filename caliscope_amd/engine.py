"""The contract between the trust-region driver (host, Python) and a BA engine (device).

The driver — ``cba_solve`` in the library (``csrc/cba_solve.cpp``), the ONE driver of the package — never sees a vector: every
O(n) / O(N_obs) operation lives behind this interface and only scalars cross it.  The production implementation is
:class:`caliscope_amd.hip_engine.HipEngine` (hand-written HIP kernels behind the C ABI of ``include/caliscope_ba.h``); an engine
is anything with these primitives and a ``solve()``.  The CPU tests plug in a numpy engine built on the oracle
(``oracle/engine.py``) whose ``solve()`` runs a Python restatement of the same loop (``oracle/trf_driver.py``): test
infrastructure, never imported from this package.

Notation.  ``J`` and ``f`` are the robust-loss-scaled Jacobian and
residuals at the current point ``x``; ``g = J^T f``; ``D = scale_inv`` are the Jacobi column norms
with scipy's monotone-max rule (``x_scale='jac'``), ``d = 1/D``; hatted quantities live in the
scaled space: ``g_h = d*g``, ``J_h = J diag(d)``.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Protocol

import numpy as np

LOSS_CODES = {"linear": 0, "huber": 1, "soft_l1": 2, "cauchy": 3, "arctan": 4}

# scipy's termination codes (least_squares.py:241 ff.), named as OptimizationStatus.termination_reason wants them (capture_volume.py:60-67)
STATUS_REASONS = {
    -1: "improper_input",
    0: "max_evaluations",
    1: "converged_gtol",
    2: "converged_ftol",
    3: "converged_xtol",
    4: "converged_small_step",
}


@dataclass
class TrfResult:
    """What an engine's ``solve()`` returns (``cba_result`` + the solution)."""

    x: np.ndarray
    cost: float
    optimality: float
    nfev: int
    njev: int
    status: int
    n_iterations: int = 0
    seconds: float = 0.0            # wall time of the solve (cba_result.t_total_s)
    rejected_seconds: float = 0.0   # of it: the separately evaluated trial points that were rejected (cba_result.t_rejected_s) ...
    rejected_timed: int = 0         # ... and their number
    trace: list = field(default_factory=list)  # per outer iteration: dict(cost, g_norm, Delta, lam, nfev)

    @property
    def message(self) -> str:
        return STATUS_REASONS.get(self.status, f"unknown_{self.status}")

    @property
    def success(self) -> bool:
        return self.status > 0


@dataclass
class BAProblem:
    """Everything ``least_squares`` receives at the reference seam (``capture_volume.py:387-411``)."""

    parameterization: "object"  # BundleParameterization
    camera_indices: np.ndarray  # (N,) int
    image_coords: np.ndarray  # (N, 2) float64
    obj_indices: np.ndarray  # (N,) int
    loss: str = "linear"
    f_scale: float = 1.0
    # rigid-distance constraint rows (reference capture_volume.py:446-531): (n_con, 4) endpoint groups of world-point
    # rows, target distances, row weights; None = no constraint rows
    constraint_groups_a: np.ndarray | None = None
    constraint_groups_b: np.ndarray | None = None
    constraint_distances: np.ndarray | None = None
    constraint_weights: np.ndarray | None = None

    def __post_init__(self):
        self.camera_indices = np.ascontiguousarray(self.camera_indices, dtype=np.int32)
        self.image_coords = np.ascontiguousarray(self.image_coords, dtype=np.float64).reshape(-1, 2)
        self.obj_indices = np.ascontiguousarray(self.obj_indices, dtype=np.int32)
        n = self.camera_indices.shape[0]
        if self.image_coords.shape[0] != n or self.obj_indices.shape[0] != n:
            raise ValueError("camera_indices, image_coords and obj_indices must have the same length")
        if self.loss not in LOSS_CODES:
            raise ValueError(f"`loss` must be one of {sorted(LOSS_CODES)}, got {self.loss!r}")
        if n:
            if self.camera_indices.min() < 0 or self.camera_indices.max() >= len(self.parameterization.blocks):
                raise ValueError("camera index out of range")
            if self.obj_indices.min() < 0 or self.obj_indices.max() >= self.parameterization.n_points:
                raise ValueError("world-point index out of range")
        if self.constraint_groups_a is not None:
            self.constraint_groups_a = np.ascontiguousarray(self.constraint_groups_a, dtype=np.int32).reshape(-1, 4)
            self.constraint_groups_b = np.ascontiguousarray(self.constraint_groups_b, dtype=np.int32).reshape(-1, 4)
            self.constraint_distances = np.ascontiguousarray(self.constraint_distances, dtype=np.float64).ravel()
            self.constraint_weights = np.ascontiguousarray(self.constraint_weights, dtype=np.float64).ravel()
            m = self.constraint_groups_a.shape[0]
            if not (self.constraint_groups_b.shape[0] == self.constraint_distances.size == self.constraint_weights.size == m):
                raise ValueError("constraint arrays must have the same length")
            P = self.parameterization.n_points
            if m and (min(self.constraint_groups_a.min(), self.constraint_groups_b.min()) < 0
                      or max(self.constraint_groups_a.max(), self.constraint_groups_b.max()) >= P):
                raise ValueError("constraint point index out of range")

    @property
    def n_constraints(self) -> int:
        return 0 if self.constraint_groups_a is None else int(self.constraint_groups_a.shape[0])

    def constraint_args(self):
        """The four trailing ``args`` of the reference's least_squares call (all None without constraints)."""
        return (self.constraint_groups_a, self.constraint_groups_b, self.constraint_distances, self.constraint_weights)

    @property
    def n_obs(self) -> int:
        return int(self.camera_indices.shape[0])

    @property
    def n_params(self) -> int:
        return int(self.parameterization.n_camera_params) + 3 * int(self.parameterization.n_points)


@dataclass
class Linearization:
    g_norm_inf: float  # ||g||_inf                       (first-order optimality, gtol test)
    gh_sq: float  # ||g_h||^2
    jg_sq: float  # ||J_h g_h||^2                    (regularisation term of the TR sub-problem)
    x_scaled_norm: float  # ||x * D||                         (initial trust radius)
    x_norm: float  # ||x||                             (xtol test)


@dataclass
class NewtonStep:
    ok: bool  # False: the damped system could not be factorised
    p_sq: float  # ||p||^2      with p the scaled damped Gauss-Newton step,  (J_h^T J_h + lam I) p = -g_h
    gh_dot_p: float  # <g_h, p>
    w_sq: float  # ||p - (<g_h,p>/||g_h||^2) g_h||^2   (for a stable orthonormal basis of span{g_h, p})


@dataclass
class Trial:
    cost: float  # 0.5 * sum rho(f(x + step)) — nan/inf if any residual is not finite
    step_norm: float  # ||step||  (unscaled)
    finite: bool


class BAEngine(Protocol):
    n_params: int

    def begin(self, x0: np.ndarray) -> float:
        """Load x0, evaluate residuals; return the initial cost."""

    def linearize(self) -> Linearization:
        """Build the normal-equation blocks at the current x and refresh the Jacobi scaling."""

    def newton_step(self, lam: float) -> NewtonStep:
        """Solve ``(J^T J + lam D^2) s = -g`` by Schur complement; keep ``s`` on the device."""

    def subspace_gram(self, a1: float, b1: float, a2: float, b2: float) -> tuple[float, float, float]:
        """``(|J v1|^2, <J v1, J v2>, |J v2|^2)`` for ``v_i = a_i * d^2 * g + b_i * s`` (one pass over the observations)."""

    def trial(self, alpha: float, beta: float) -> Trial:
        """Form ``x_new = x + alpha * d^2 * g + beta * s`` and evaluate the cost there."""

    def camera_params(self, which: int = 0) -> np.ndarray:
        """Camera part of x (``which=0``) or of the last trial point (``which=1``)."""

    def accept(self) -> None:
        """Make the last trial point the current point."""

    def current_x(self) -> np.ndarray:
        """Download the current point in the reference's parameter layout."""
