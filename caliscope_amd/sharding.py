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


def _blocked_cuts(n_points: int, groups) -> np.ndarray:
    """blocked[c] is True when cutting between points c - 1 and c would split a connected component of the constraint
    graph (rigid-distance rows couple the points they name: a component must live on one rank)."""
    blocked = np.zeros(n_points + 1, dtype=bool)
    if groups is None:
        return blocked
    rows = np.concatenate([np.asarray(g, dtype=np.int64).reshape(-1, 4) for g in groups], axis=1)  # (n_con, 8)
    if rows.size == 0:
        return blocked
    # components by union-find over the rows, then one (min, max) point interval per component
    parent = np.arange(n_points)

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for row in rows:
        r0 = find(int(row[0]))
        for q in row[1:]:
            r = find(int(q))
            if r != r0:
                parent[r] = r0
    used = np.unique(rows)
    roots = np.array([find(int(q)) for q in used])
    order = np.argsort(roots, kind="stable")
    roots, used = roots[order], used[order]
    first = np.flatnonzero(np.r_[True, roots[1:] != roots[:-1]])
    lo = np.minimum.reduceat(used, first)
    hi = np.maximum.reduceat(used, first)
    delta = np.zeros(n_points + 2, dtype=np.int64)  # cuts lo < c <= hi are blocked
    np.add.at(delta, lo + 1, 1)
    np.add.at(delta, hi + 1, -1)
    blocked[:] = np.cumsum(delta)[: n_points + 1] > 0
    return blocked


def partition_points(obj_indices: np.ndarray, n_points: int, world: int, constraint_groups=None) -> list[np.ndarray]:
    """Contiguous point ranges balanced by observation count; returns the sorted point ids of every rank.
    ``constraint_groups`` = (groups_a, groups_b): cuts are moved forward so that no constraint component is split."""
    if world < 1:
        raise ValueError("world must be >= 1")
    blocked = _blocked_cuts(n_points, constraint_groups)
    counts = np.bincount(np.asarray(obj_indices, dtype=np.int64), minlength=n_points).astype(np.int64)
    cum = np.concatenate([[0], np.cumsum(counts)])
    total = int(cum[-1])
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        # first point index whose cumulative count reaches the target, never going backwards
        cut = min(max(int(np.searchsorted(cum, target, side="left")), bounds[-1]), n_points)
        while cut < n_points and blocked[cut]:
            cut += 1
        bounds.append(cut)
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
    has_con = problem.n_constraints > 0
    groups = (problem.constraint_groups_a, problem.constraint_groups_b) if has_con else None
    owned = partition_points(problem.obj_indices, par.n_points, world, groups)[rank]
    g2l = -np.ones(par.n_points, dtype=np.int64)
    g2l[owned] = np.arange(owned.size)
    local_obj = g2l[problem.obj_indices]
    keep = local_obj >= 0
    if owned.size == 0 or not keep.any():
        raise ValueError(f"rank {rank} of {world} owns no observations; use fewer ranks for this problem")
    local_par = dataclasses.replace(par, n_points=int(owned.size))
    con = {}
    if has_con:  # a component is never split (partition_points), so a row is local as a whole or not at all
        la, lb = g2l[problem.constraint_groups_a], g2l[problem.constraint_groups_b]
        mine = la[:, 0] >= 0
        assert np.all((la >= 0).all(axis=1) == mine) and np.all((lb >= 0).all(axis=1) == mine)
        if mine.any():
            con = dict(constraint_groups_a=la[mine].astype(np.int32), constraint_groups_b=lb[mine].astype(np.int32),
                       constraint_distances=problem.constraint_distances[mine], constraint_weights=problem.constraint_weights[mine])
    local = BAProblem(
        local_par, problem.camera_indices[keep], problem.image_coords[keep], local_obj[keep].astype(np.int32),
        loss=problem.loss, f_scale=problem.f_scale, **con,
    )
    return Shard(local, owned, par.n_points)
