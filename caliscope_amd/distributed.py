"""Sharded solves: the world points partitioned over several GPUs, cameras replicated.

Two ways to run one, both on the same library protocol (``cba_solve`` on every rank, the camera blocks, the reduced
camera system and the scalar sums all-reduced inside libcaliscope_ba.so on the engines' HIP streams):

* :func:`solve_multi_device` — ONE process, one host thread per GPU.  This is what a caller of
  ``CaptureVolume.optimize()`` gets (the reference's solve is a single in-process call, ``core/capture_volume.py:322-334``):
  ``caliscope_amd.least_squares.least_squares`` routes here when ``devices=[...]`` or ``CALISCOPE_HIP_DEVICES=0,1,...``
  names more than one device.  The data plane is RCCL over xGMI (threads share a unique id) or, with ``backend="direct"``,
  the library's peer-to-peer group exchange; the latter also accepts the same device several times, which is how a
  1-GPU box executes the multi-rank protocol in the tests.
* :func:`solve_sharded` — one process per GPU (``python -m torch.distributed.run`` or any launcher that sets
  ``RANK / WORLD_SIZE``), RCCL between the processes; :class:`SocketControlPlane` is the host-side plumbing
  (rendezvous, RCCL's unique id, gathering the solution) over plain TCP on the loopback interface.

Nothing here imports torch; with ``world == 1`` nothing distributed is touched.
"""

from __future__ import annotations

import os
import socket
import struct
import threading
import time

import numpy as np

from caliscope_amd.engine import BAProblem
from caliscope_amd.sharding import Shard, shard_problem
from caliscope_amd.engine import TrfResult


# ------------------------------------------------------------------------------------------------------------------
# host-side control planes (never on the data path)
class SoloControlPlane:
    rank, world = 0, 1

    def barrier(self):
        pass

    def broadcast_bytes(self, payload, n):
        return bytes(payload)

    def allreduce_sum(self, a):
        return np.asarray(a, dtype=np.float64)

    def allreduce_max(self, v):
        return float(v)

    def close(self):
        pass


def _send_msg(sock, payload: bytes) -> None:
    sock.sendall(struct.pack("<q", len(payload)) + payload)


def _recv_exact(sock, n: int) -> bytes:
    chunks, got = [], 0
    while got < n:
        b = sock.recv(min(n - got, 1 << 20))
        if not b:
            raise ConnectionError("control-plane peer closed the connection")
        chunks.append(b)
        got += len(b)
    return b"".join(chunks)


def _recv_msg(sock) -> bytes:
    (n,) = struct.unpack("<q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


class SocketControlPlane:
    """Host-side collectives of a multi-process solve over TCP (star: rank 0 serves).  Single node: everything binds
    and connects on 127.0.0.1 unless ``host`` says otherwise."""

    def __init__(self, rank: int, world: int, port: int, host: str = "127.0.0.1", timeout: float = 120.0):
        self.rank, self.world = int(rank), int(world)
        self._peers: list[socket.socket] = []
        self._up: socket.socket | None = None
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket()
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((host, port))
            srv.listen(self.world)
            srv.settimeout(timeout)
            by_rank = {}
            while len(by_rank) < self.world - 1:
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(None)
                (r,) = struct.unpack("<i", _recv_exact(conn, 4))
                by_rank[r] = conn
            srv.close()
            self._peers = [by_rank[r] for r in range(1, self.world)]
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((host, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(None)
            s.sendall(struct.pack("<i", self.rank))
            self._up = s

    @classmethod
    def from_env(cls, timeout: float = 120.0) -> "SocketControlPlane":
        """``RANK`` / ``WORLD_SIZE`` as a launcher sets them.  The port is ``CBA_CONTROL_PORT`` when given; otherwise
        rank 0 binds a free port and leaves it in a rendezvous file keyed by the launcher (parent pid + ``MASTER_PORT``),
        which the other ranks of the same launch poll — ``MASTER_PORT`` itself belongs to the launcher's own store."""
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        if world == 1:
            return cls(0, 1, 0)
        explicit = os.environ.get("CBA_CONTROL_PORT")
        if explicit:
            return cls(rank, world, int(explicit), timeout=timeout)
        path = f"/tmp/cba_rdzv_{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}"
        if rank == 0:
            with socket.socket() as probe:
                probe.bind(("127.0.0.1", 0))
                port = probe.getsockname()[1]
            tmp = f"{path}.{os.getpid()}"
            with open(tmp, "w") as f:
                f.write(str(port))
            os.replace(tmp, path)
            try:
                return cls(rank, world, port, timeout=timeout)
            finally:
                try:
                    os.unlink(path)
                except OSError:
                    pass
        deadline = time.time() + timeout
        while True:
            try:
                with open(path) as f:
                    port = int(f.read().strip())
                break
            except (OSError, ValueError):
                if time.time() > deadline:
                    raise TimeoutError(f"rank {rank}: no rendezvous file {path}")
                time.sleep(0.02)
        return cls(rank, world, port, timeout=timeout)

    def _reduce(self, mine: bytes, combine) -> bytes:
        if self.world == 1:
            return mine
        if self.rank == 0:
            parts = [mine] + [_recv_msg(p) for p in self._peers]
            out = combine(parts)
            for p in self._peers:
                _send_msg(p, out)
            return out
        _send_msg(self._up, mine)
        return _recv_msg(self._up)

    def broadcast_bytes(self, payload: bytes | None, n: int) -> bytes:
        mine = bytes(payload) if self.rank == 0 else b""
        return self._reduce(mine, lambda parts: parts[0])[:n]

    def allreduce_sum(self, a: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.float64)
        out = self._reduce(a.tobytes(), lambda parts: np.sum([np.frombuffer(p, dtype=np.float64) for p in parts], axis=0).tobytes())
        return np.frombuffer(out, dtype=np.float64).reshape(a.shape).copy()

    def allreduce_max(self, v: float) -> float:
        out = self._reduce(struct.pack("<d", float(v)), lambda parts: struct.pack("<d", max(struct.unpack("<d", p)[0] for p in parts)))
        return struct.unpack("<d", out)[0]

    def barrier(self) -> None:
        self._reduce(b"", lambda parts: b"")

    def close(self) -> None:
        for s in self._peers + ([self._up] if self._up is not None else []):
            try:
                s.close()
            except OSError:
                pass
        self._peers, self._up = [], None


class _ThreadGroupState:
    def __init__(self, world: int):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world


class ThreadControlPlane:
    """Host-side collectives among the member threads of :func:`solve_multi_device`."""

    def __init__(self, state: _ThreadGroupState, rank: int):
        self._s, self.rank, self.world = state, rank, state.world

    def _exchange(self, mine):
        self._s.slots[self.rank] = mine
        self._s.barrier.wait()
        parts = list(self._s.slots)
        self._s.barrier.wait()
        return parts

    def broadcast_bytes(self, payload, n):
        return bytes(self._exchange(payload)[0])[:n]

    def allreduce_sum(self, a):
        return np.sum(self._exchange(np.asarray(a, dtype=np.float64)), axis=0)

    def allreduce_max(self, v):
        return float(max(self._exchange(float(v))))

    def barrier(self):
        self._s.barrier.wait()

    def abort(self):
        self._s.barrier.abort()

    def close(self):
        pass


# ------------------------------------------------------------------------------------------------------------------
def make_sharded_hip_engine(problem: BAProblem, control, device_id: int = -1, group=None, on_engine=None):
    """Shard ``problem`` for this rank, create its HIP engine and join the communicator: the device group ``group``
    (``caliscope_amd.hip_engine.DeviceGroup``, in-process) or RCCL (unique id passed over ``control``)."""
    from caliscope_amd.hip_engine import HipEngine

    shard = shard_problem(problem, control.rank, control.world)
    engine = HipEngine(shard.problem, device_id=device_id)
    if on_engine is not None:
        # before the communicator is created: a peer that fails from here on marks this handle aborted, so that comm_init (if it has not begun) and
        # every later collective fail instead of waiting for it.  (A rank already INSIDE ncclCommInitRank when a peer dies before calling it is not
        # released by this — RCCL's blocking rendezvous has no abort; the peers' host-side barrier abort covers the ranks that have not entered it.)
        on_engine(engine)
    try:
        if control.world > 1:
            if group is not None:
                engine.group_join(group, control.rank)
            else:
                uid = engine.comm_unique_id() if control.rank == 0 else None
                uid = control.broadcast_bytes(uid, 128)
                engine.comm_init(uid, control.rank, control.world)
    except Exception:
        engine.close()
        raise
    return engine, shard


def gather_solution(shard: Shard, x_local: np.ndarray, control) -> np.ndarray:
    """Full parameter vector in the reference layout on every rank."""
    ncp = shard.problem.parameterization.n_camera_params
    pts = shard.scatter_points(x_local)
    if control.world > 1:
        pts = control.allreduce_sum(pts.reshape(-1)).reshape(-1, 3)
    return np.concatenate([x_local[:ncp], pts.reshape(-1)])


def solve_sharded(problem: BAProblem, x0: np.ndarray, control, *, device_id: int = -1, engine_factory=None, group=None, cam_bounds=None,
                  on_engine=None, **tol) -> TrfResult:
    """Solve ``problem`` with its points sharded over ``control.world`` ranks; every rank returns the full x.

    ``engine_factory(shard, control)`` is a test hook (the CPU tests plug the numpy oracle engine in); ``cam_bounds`` =
    (lb, ub) of the camera block when the caller's bounds are not ``parameterization.bounds()``; ``on_engine(engine)`` is told
    the rank's engine as soon as it exists (``solve_multi_device`` keeps it to abort this rank's communicator when a peer fails)."""
    if engine_factory is None:
        engine, shard = make_sharded_hip_engine(problem, control, device_id, group, on_engine=on_engine)
    else:
        shard = shard_problem(problem, control.rank, control.world)
        engine = engine_factory(shard, control)
        if on_engine is not None:
            on_engine(engine)
    try:
        par = shard.problem.parameterization
        ncp = par.n_camera_params
        lb, ub = par.bounds() if cam_bounds is None else (np.asarray(cam_bounds[0], dtype=np.float64), np.asarray(cam_bounds[1], dtype=np.float64))
        bounded = bool(np.any(np.isfinite(lb[:ncp])) or np.any(np.isfinite(ub[:ncp])))
        x_local = shard.local_x(np.asarray(x0, dtype=np.float64))
        if tol.get("max_nfev") is None:
            # scipy's default 100 n refers to the whole problem; the shards have different sizes and must stop together
            tol["max_nfev"] = 100 * int(problem.n_params)
        # the engine's solve() on every rank (the library's cba_solve for the device engine): the scalars that steer it are identical everywhere
        res = engine.solve(x_local, lb=np.ascontiguousarray(lb[:ncp]) if bounded else None,
                           ub=np.ascontiguousarray(ub[:ncp]) if bounded else None, **tol)
        res.x = gather_solution(shard, res.x, control)
    finally:
        close = getattr(engine, "close", None)
        if close is not None:
            close()
    return res


def devices_from_env() -> list[int] | None:
    """``CALISCOPE_HIP_DEVICES=0,1,2,3``: the devices ``least_squares`` shards a solve over (None: one device)."""
    raw = os.environ.get("CALISCOPE_HIP_DEVICES", "").strip()
    if not raw:
        return None
    try:
        return [int(t) for t in raw.split(",") if t.strip() != ""]
    except ValueError as exc:
        raise ValueError(f"CALISCOPE_HIP_DEVICES={raw!r}: expected a comma-separated list of device ordinals") from exc


def solve_multi_device(problem: BAProblem, x0: np.ndarray, devices, *, backend: str = "auto", **tol) -> TrfResult:
    """One process, ``len(devices)`` GPUs: a host thread per device runs the sharded solve of its rank.

    ``backend``: ``"rccl"`` (RCCL communicator shared by the threads), ``"direct"`` (the library's peer-to-peer group
    exchange; required when a device is named more than once) or ``"auto"`` (``CBA_XCHG`` if set, else RCCL for distinct
    devices, direct otherwise).  Returns rank 0's result; all ranks end with identical bits."""
    devices = [int(d) for d in devices]
    world = len(devices)
    if world < 1:
        raise ValueError("solve_multi_device: no devices given")
    if world == 1:
        return solve_sharded(problem, x0, SoloControlPlane(), device_id=devices[0], **tol)
    if backend == "auto":
        backend = os.environ.get("CBA_XCHG", "rccl" if len(set(devices)) == world else "direct")
    if backend not in ("rccl", "direct"):
        raise ValueError(f"backend must be 'rccl', 'direct' or 'auto', got {backend!r}")
    if backend == "rccl" and len(set(devices)) != world:
        raise ValueError("RCCL needs distinct devices; use backend='direct' to place several ranks on one device")
    from caliscope_amd.hip_engine import DeviceGroup

    group = DeviceGroup(world) if backend == "direct" else None
    state = _ThreadGroupState(world)
    results: list = [None] * world
    errors: list = [None] * world
    engines: list = [None] * world
    engines_lock = threading.Lock()

    def member(rank):
        ctl = ThreadControlPlane(state, rank)

        def keep(engine):
            with engines_lock:
                engines[rank] = engine

        try:
            results[rank] = solve_sharded(problem, x0, ctl, device_id=devices[rank], group=group, on_engine=keep, **dict(tol))
        except BaseException as exc:  # noqa: BLE001 - re-raised in the caller's thread
            errors[rank] = (time.monotonic(), exc)
            ctl.abort()  # peers waiting in a host-side exchange fail instead of hanging
            if group is not None:
                group.abort()  # ... and so do peers spinning in the library's group barrier
            else:
                # RCCL: a peer may already sit in an all-reduce (or in the stream wait behind it) that this rank will never join;
                # ncclCommAbort on its communicator makes that call return an error instead of hanging least_squares for ever
                with engines_lock:
                    peers = [e for r, e in enumerate(engines) if r != rank and e is not None]
                for e in peers:
                    abort = getattr(e, "comm_abort", None)
                    if abort is not None:
                        try:
                            abort()
                        except Exception:  # noqa: BLE001 - best effort on the failure path
                            pass
        finally:
            with engines_lock:
                engines[rank] = None  # closed by solve_sharded: never abort a destroyed handle

    threads = [threading.Thread(target=member, args=(r,), name=f"cba-rank{r}", daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if group is not None:
        group.close()
    failed = sorted((e for e in errors if e is not None), key=lambda te: te[0])
    if failed:
        raise failed[0][1]  # the rank that failed first; the others only report that the group was aborted
    return results[0]
