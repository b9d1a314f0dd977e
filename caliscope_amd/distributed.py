"""Sharded solve driver: one process per GPU, launched e.g. by ``python -m torch.distributed.run``.

``torch.distributed`` is used here only as host-side plumbing (rendezvous, broadcasting RCCL's unique id,
gathering the solution) — the data-plane collectives run inside libcaliscope_ba.so on the engine's HIP stream
(``cba_comm_init``).  Everything is optional: with ``world == 1`` nothing distributed is touched.
"""

from __future__ import annotations

import os

import numpy as np

from caliscope_amd.engine import BAProblem
from caliscope_amd.sharding import Shard, shard_problem
from caliscope_amd.trf import TrfResult, trf_solve


class TorchControlPlane:
    """Host-side collectives over an initialised torch.distributed process group (any backend for CPU tensors)."""

    def __init__(self, group=None):
        import torch
        import torch.distributed as dist

        self._torch, self._dist, self._group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def broadcast_bytes(self, payload: bytes | None, n: int) -> bytes:
        t = self._torch.zeros(n, dtype=self._torch.uint8)
        if self.rank == 0:
            t[: len(payload)] = self._torch.frombuffer(bytearray(payload), dtype=self._torch.uint8)
        self._dist.broadcast(t, src=0, group=self._group)
        return bytes(t.numpy().tobytes())

    def allreduce_sum(self, a: np.ndarray) -> np.ndarray:
        t = self._torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self._group)
        return t.numpy()

    def allreduce_max(self, v: float) -> float:
        t = self._torch.tensor([float(v)], dtype=self._torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX, group=self._group)
        return float(t[0])

    def barrier(self):
        self._dist.barrier(group=self._group)


def make_sharded_hip_engine(problem: BAProblem, control, device_id: int = -1):
    """Shard ``problem`` for this rank, create its HIP engine and join the RCCL communicator."""
    from caliscope_amd.hip_engine import HipEngine

    shard = shard_problem(problem, control.rank, control.world)
    engine = HipEngine(shard.problem, device_id=device_id)
    if control.world > 1:
        uid = engine.comm_unique_id() if control.rank == 0 else None
        uid = control.broadcast_bytes(uid, 128)
        engine.comm_init(uid, control.rank, control.world)
    return engine, shard


def gather_solution(shard: Shard, x_local: np.ndarray, control) -> np.ndarray:
    """Full parameter vector in the reference layout on every rank."""
    ncp = shard.problem.parameterization.n_camera_params
    pts = shard.scatter_points(x_local)
    if control.world > 1:
        pts = control.allreduce_sum(pts.reshape(-1)).reshape(-1, 3)
    return np.concatenate([x_local[:ncp], pts.reshape(-1)])


def solve_sharded(problem: BAProblem, x0: np.ndarray, control, *, device_id: int = -1, engine_factory=None, **tol) -> TrfResult:
    """Solve ``problem`` with its points sharded over ``control.world`` ranks; every rank returns the full x.

    ``engine_factory(shard, control)`` is a test hook (the CPU tests plug the numpy oracle engine in)."""
    if engine_factory is None:
        engine, shard = make_sharded_hip_engine(problem, control, device_id)
    else:
        shard = shard_problem(problem, control.rank, control.world)
        engine = engine_factory(shard, control)
    try:
        par = shard.problem.parameterization
        ncp = par.n_camera_params
        lb, ub = par.bounds()
        bounded = bool(np.any(np.isfinite(lb[:ncp])) or np.any(np.isfinite(ub[:ncp])))
        x_local = shard.local_x(np.asarray(x0, dtype=np.float64))
        if hasattr(engine, "solve") and os.environ.get("CBA_HOST_LOOP", "native") != "python":
            # the library's driver (cba_solve) on every rank: the scalars that steer it are identical everywhere
            res = engine.solve(x_local, lb=np.ascontiguousarray(lb[:ncp]) if bounded else None,
                               ub=np.ascontiguousarray(ub[:ncp]) if bounded else None, **tol)
        else:
            feasible = (lambda c: bool(np.all(c > lb[:ncp]) and np.all(c < ub[:ncp]))) if bounded else None
            res = trf_solve(engine, x_local, feasible=feasible, **tol)
        res.x = gather_solution(shard, res.x, control)
    finally:
        close = getattr(engine, "close", None)
        if close is not None:
            close()
    return res
