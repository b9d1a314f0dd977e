"""Point sharding and the multi-rank protocol on CPU: world_size 2 — two processes over the stdlib TCP control plane and
two threads over the in-process one — with the numpy oracle engine standing in for the HIP engine (same BAEngine protocol,
same set of all-reduced quantities).  tests/test_multi_device.py runs the same protocol on the device."""
import os
import socket

import numpy as np
import pytest

from caliscope_amd.engine import BAProblem
from caliscope_amd.sharding import partition_points, shard_problem
from tests.helpers import small_problem


def test_partition_is_contiguous_complete_and_balanced():
    rng = np.random.default_rng(0)
    counts = rng.integers(0, 15, 1000)
    obj = np.repeat(np.arange(1000), counts)
    for world in (1, 2, 3, 4, 8):
        parts = partition_points(obj, 1000, world)
        assert len(parts) == world
        assert np.array_equal(np.concatenate(parts), np.arange(1000))
        loads = [counts[p].sum() for p in parts]
        assert max(loads) - min(loads) <= 2 * counts.max()
    assert all(p.size == 0 for p in partition_points(np.array([], dtype=int), 0, 2))


def test_shard_problem_reindexes_points_and_keeps_all_cameras():
    sc, par, x0 = small_problem(n_cams=5, n_points=101, k=4)
    prob = BAProblem(par, sc.camera_indices, sc.image_coords, sc.obj_indices)
    shards = [shard_problem(prob, r, 3) for r in range(3)]
    assert sum(s.problem.n_obs for s in shards) == prob.n_obs
    assert np.array_equal(np.sort(np.concatenate([s.owned_points for s in shards])), np.arange(101))
    for s in shards:
        lp = s.problem.parameterization
        assert lp.n_camera_params == par.n_camera_params and lp.n_points == s.owned_points.size
        assert s.problem.obj_indices.max() == lp.n_points - 1
        xl = s.local_x(x0)
        assert np.array_equal(xl[: par.n_camera_params], x0[: par.n_camera_params])
        assert np.array_equal(s.scatter_points(xl)[s.owned_points], x0[par.n_camera_params :].reshape(-1, 3)[s.owned_points])
    total = sum(s.scatter_points(s.local_x(x0)) for s in shards)
    assert np.array_equal(total.reshape(-1), x0[par.n_camera_params :])
    with pytest.raises(ValueError, match="owns no observations"):
        shard_problem(BAProblem(par, sc.camera_indices[:8], sc.image_coords[:8], np.zeros(8, dtype=np.int32)), 1, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(target, world, *args):
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=target, args=(r, world) + args) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]


def _worker(rank, world, port, loss, out_dir):
    from caliscope_amd.distributed import SocketControlPlane, solve_sharded
    from oracle.engine import OracleEngine

    ctl = SocketControlPlane(rank, world, port)
    try:
        sc, par, x0 = small_problem(n_cams=6, n_points=240, k=6, loss=loss, outliers=0.05 if loss != "linear" else 0.0)
        fs = sc.f_scale_1px() * 2.0 if loss != "linear" else 1.0
        prob = BAProblem(par, sc.camera_indices, sc.image_coords, sc.obj_indices, loss=loss, f_scale=fs)

        def factory(shard, control):
            sp = shard.problem
            eng = OracleEngine(sp.parameterization, sp.camera_indices, sp.image_coords, sp.obj_indices, loss=loss,
                               f_scale=fs, allreduce=control.allreduce_sum)
            eng.allreduce_max = control.allreduce_max
            # rank > 0 must not count the replicated camera entries twice: the oracle engine splits every
            # vector into a camera part (added locally) and a point part (all-reduced), see OracleEngine._norm_sq
            return eng

        res = solve_sharded(prob, x0, ctl, engine_factory=factory, ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=200)
        np.save(os.path.join(out_dir, f"x_{rank}.npy"), res.x)
        np.save(os.path.join(out_dir, f"meta_{rank}.npy"), np.array([res.cost, res.nfev, res.status], dtype=np.float64))
    finally:
        ctl.close()


@pytest.mark.parametrize("loss", ["linear", "soft_l1"])
def test_two_rank_socket_solve_matches_single_rank(tmp_path, loss):
    from oracle.trf_driver import trf_solve
    from oracle.engine import OracleEngine

    world = 2
    _spawn(_worker, world, _free_port(), loss, str(tmp_path))
    sc, par, x0 = small_problem(n_cams=6, n_points=240, k=6, loss=loss, outliers=0.05 if loss != "linear" else 0.0)
    fs = sc.f_scale_1px() * 2.0 if loss != "linear" else 1.0
    single = trf_solve(OracleEngine(par, sc.camera_indices, sc.image_coords, sc.obj_indices, loss=loss, f_scale=fs), x0,
                       ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=200)
    xs = [np.load(tmp_path / f"x_{r}.npy") for r in range(world)]
    metas = [np.load(tmp_path / f"meta_{r}.npy") for r in range(world)]
    assert np.array_equal(xs[0], xs[1]), "ranks must agree bit for bit (replicated dense solve)"
    assert np.array_equal(metas[0], metas[1])
    assert abs(metas[0][0] - single.cost) <= 1e-10 * single.cost
    assert int(metas[0][1]) == single.nfev and int(metas[0][2]) == single.status
    # the summation order differs between 1 and 2 ranks; along the gauge directions (damped only by lam ~ 1e-13)
    # that rounding noise shows up at 1e-6 raw, so compare after similarity alignment like everywhere else
    from tests.helpers import aligned_difference

    pos, ang, _ = aligned_difference(par, xs[0], single.x)
    assert pos < 1e-8 and ang < 1e-8, (pos, ang)


def test_partition_keeps_constraint_components_together():
    from tests.constrained_scene import board_scene

    sc = board_scene(n_frames=9)
    ga, gb, dist, w = sc["constraints"]
    n_points, n_per = len(sc["points_true"]), sc["n_per"]
    for world in (2, 3, 4):
        parts = partition_points(sc["obj"], n_points, world, (ga, gb))
        assert np.array_equal(np.concatenate(parts), np.arange(n_points))
        for part in parts:
            assert part.size % n_per == 0 and part[0] % n_per == 0  # whole boards only
        prob = BAProblem(sc["par"], sc["cam"], sc["uv"], sc["obj"], constraint_groups_a=ga, constraint_groups_b=gb,
                         constraint_distances=dist, constraint_weights=w)
        shards = [shard_problem(prob, r, world) for r in range(world)]
        assert sum(s.problem.n_constraints for s in shards) == len(dist)
        for s in shards:
            pts = sc["x0"][sc["par"].n_camera_params:].reshape(-1, 3)[s.owned_points]
            la, lb = s.problem.constraint_groups_a, s.problem.constraint_groups_b
            assert la.max() < s.owned_points.size and la.min() >= 0
            # local rows measure the same distances as the global ones
            d_loc = np.linalg.norm(pts[la].mean(axis=1) - pts[lb].mean(axis=1), axis=1)
            assert np.all(np.abs(d_loc - s.problem.constraint_distances) < 0.05)
    # without constraints the cuts may fall inside a board
    free = partition_points(sc["obj"], n_points, 4)
    assert any(part[0] % n_per for part in free[1:])


def _con_worker(rank, world, port, out_dir):
    from caliscope_amd.distributed import SocketControlPlane, solve_sharded
    from oracle.engine import OracleEngine
    from tests.constrained_scene import board_scene

    ctl = SocketControlPlane(rank, world, port)
    try:
        sc = board_scene(n_frames=9)
        ga, gb, d, w = sc["constraints"]
        prob = BAProblem(sc["par"], sc["cam"], sc["uv"], sc["obj"], constraint_groups_a=ga, constraint_groups_b=gb,
                         constraint_distances=d, constraint_weights=w)

        def factory(shard, control):
            sp = shard.problem
            eng = OracleEngine(sp.parameterization, sp.camera_indices, sp.image_coords, sp.obj_indices, allreduce=control.allreduce_sum,
                               constraints=sp.constraint_args() if sp.n_constraints else None)
            eng.allreduce_max = control.allreduce_max
            return eng

        res = solve_sharded(prob, sc["x0"], ctl, engine_factory=factory, ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=100)
        np.save(os.path.join(out_dir, f"x_{rank}.npy"), res.x)
        np.save(os.path.join(out_dir, f"meta_{rank}.npy"), np.array([res.cost, res.nfev, res.status], dtype=np.float64))
    finally:
        ctl.close()


def test_two_rank_socket_solve_with_constraint_rows(tmp_path):
    """Constraint components stay on one rank; their share of the reduced camera system rides in the same all-reduce."""
    from oracle.trf_driver import trf_solve
    from oracle.engine import OracleEngine
    from tests.constrained_scene import board_scene
    from tests.helpers import aligned_difference

    world = 2
    _spawn(_con_worker, world, _free_port(), str(tmp_path))
    sc = board_scene(n_frames=9)
    single = trf_solve(OracleEngine(sc["par"], sc["cam"], sc["uv"], sc["obj"], constraints=sc["constraints"]), sc["x0"],
                       ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=100)
    xs = [np.load(tmp_path / f"x_{r}.npy") for r in range(world)]
    metas = [np.load(tmp_path / f"meta_{r}.npy") for r in range(world)]
    assert np.array_equal(xs[0], xs[1]) and np.array_equal(metas[0], metas[1])
    assert abs(metas[0][0] - single.cost) <= 1e-10 * single.cost and int(metas[0][2]) == single.status
    pos, ang, scale = aligned_difference(sc["par"], xs[0], single.x)
    assert pos < 1e-8 and ang < 1e-8 and abs(scale - 1) < 1e-8


def test_three_rank_thread_control_plane_matches_single_rank():
    """The in-process control plane of solve_multi_device (one thread per rank) on the numpy engine."""
    import threading

    from caliscope_amd.distributed import ThreadControlPlane, _ThreadGroupState, solve_sharded
    from oracle.trf_driver import trf_solve
    from oracle.engine import OracleEngine
    from tests.helpers import aligned_difference

    sc, par, x0 = small_problem(n_cams=6, n_points=240, k=6)
    prob = BAProblem(par, sc.camera_indices, sc.image_coords, sc.obj_indices)
    world = 3
    state = _ThreadGroupState(world)
    out = [None] * world

    def factory(shard, control):
        sp = shard.problem
        eng = OracleEngine(sp.parameterization, sp.camera_indices, sp.image_coords, sp.obj_indices, allreduce=control.allreduce_sum)
        eng.allreduce_max = control.allreduce_max
        return eng

    def member(r):
        out[r] = solve_sharded(prob, x0, ThreadControlPlane(state, r), engine_factory=factory, ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=200)

    threads = [threading.Thread(target=member, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert all(o is not None for o in out)
    single = trf_solve(OracleEngine(par, sc.camera_indices, sc.image_coords, sc.obj_indices), x0, ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=200)
    for o in out[1:]:
        assert np.array_equal(o.x, out[0].x) and o.nfev == out[0].nfev and o.cost == out[0].cost
    assert abs(out[0].cost - single.cost) <= 1e-10 * single.cost and out[0].nfev == single.nfev
    pos, ang, _ = aligned_difference(par, out[0].x, single.x)
    assert pos < 1e-8 and ang < 1e-8


def test_socket_control_plane_rendezvous_from_env(tmp_path):
    """``SocketControlPlane.from_env``: the ranks of one launch meet through the rendezvous file keyed by their parent."""
    _spawn(_env_worker, 3, str(tmp_path))
    got = [np.load(tmp_path / f"s_{r}.npy") for r in range(3)]
    for g in got:
        assert np.array_equal(g, np.array([0.0 + 1.0 + 2.0, 2.0, 7.0]))


def _env_worker(rank, world, out_dir):
    from caliscope_amd.distributed import SocketControlPlane

    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_PORT="29555")
    os.environ.pop("CBA_CONTROL_PORT", None)
    ctl = SocketControlPlane.from_env(timeout=60)
    try:
        s = ctl.allreduce_sum(np.array([float(rank)]))[0]
        m = ctl.allreduce_max(float(rank))
        b = ctl.broadcast_bytes(bytes([7]) * 4 if rank == 0 else None, 4)
        ctl.barrier()
        np.save(os.path.join(out_dir, f"s_{rank}.npy"), np.array([s, m, float(b[0])]))
    finally:
        ctl.close()
