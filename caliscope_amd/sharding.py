"""Point sharding for multi-GPU bundle adjustment (one process per GPU).

The BA normal equations couple observations only through the camera blocks, so the natural partition is by
world point: every point, with ALL of its observations, is owned by exactly one rank; camera
parameters are replicated.  Then V_p, g_p, W_p, the point back-substitution and the trial-point update are
local, and one iteration exchanges only

  * the camera blocks U_c, g_c            (after the build pass),
  * the reduced camera system (S, b)       (after the Schur pass — the one real exchange step),
  * a handful of scalar sums               (norms, costs),

as sum all-reduces (RCCL over xGMI inside libcaliscope_ba.so; ``gloo`` in the CPU tests).  The reference has
no distributed path at all; this module is new host logic.
"""

from __future__ import annotations

import dataclasses

import numpy as np

from caliscope_amd.engine import BAProblem


def partition_points(obj_indices: np.ndarray, n_points: int, world: int) -> list[np.ndarray]:
    """Contiguous point ranges balanced by observation count; returns the sorted point ids of every rank."""
    if world < 1:
        raise ValueError("world must be >= 1")
    counts = np.bincount(np.asarray(obj_indices, dtype=np.int64), minlength=n_points).astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(counts)])
    total = int(cum[-1])
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        # first point index whose cumulative count reaches the target, never going backwards
        cut = int(np.searchsorted(cum, target, side="left"))
        bounds.append(min(max(cut, bounds[-1]), n_points))
    bounds.append(n_points)
    return [np.arange(bounds[r], bounds[r + 1], dtype=np.int64) for r in range(world)]


@dataclasses.dataclass
class Shard:
    problem: BAProblem  # local problem: all cameras, owned points re-indexed 0..P_local-1
    owned_points: np.ndarray  # global ids of the owned points (sorted)
    n_points_global: int

    def local_x(self, x_global: np.ndarray) -> np.ndarray:
        ncp = self.problem.parameterization.n_camera_params
        pts = x_global[ncp:].reshape(-1, 3)[self.owned_points]
        return np.concatenate([x_global[:ncp], pts.reshape(-1)])

    def scatter_points(self, x_local: np.ndarray) -> np.ndarray:
        """(P_global, 3) array with this rank's points filled in and zeros elsewhere (sum over ranks = all points)."""
        ncp = self.problem.parameterization.n_camera_params
        out = np.zeros((self.n_points_global, 3))
        out[self.owned_points] = x_local[ncp:].reshape(-1, 3)
        return out


def shard_problem(problem: BAProblem, rank: int, world: int) -> Shard:
    par = problem.parameterization
    owned = partition_points(problem.obj_indices, par.n_points, world)[rank]
    g2l = -np.ones(par.n_points, dtype=np.int64)
    g2l[owned] = np.arange(owned.size)
    local_obj = g2l[problem.obj_indices]
    keep = local_obj >= 0
    if owned.size == 0 or not keep.any():
        raise ValueError(f"rank {rank} of {world} owns no observations; use fewer ranks for this problem")
    local_par = dataclasses.replace(par, n_points=int(owned.size))
    local = BAProblem(
        local_par, problem.camera_indices[keep], problem.image_coords[keep], local_obj[keep].astype(np.int32),
        loss=problem.loss, f_scale=problem.f_scale,
    )
    return Shard(local, owned, par.n_points)
