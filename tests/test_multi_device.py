"""Sharded solves on the device with MORE THAN ONE rank of real code (run with ``-m gpu``).

A test box has one GPU, and RCCL refuses two ranks on one device, so the ranks here are handles on the same device
joined by the library's in-process device group (``cba_group_*``, caliscope_amd.distributed.solve_multi_device with
``backend="direct"``): every rank runs ``cba_solve`` on its shard of the points and the camera blocks, the reduced camera
system and the scalar sums are exchanged by the group's rank-ordered sums — the same call sites and the same
4-collectives-per-iteration pattern that run over RCCL/xGMI on a multi-GPU node.  Reference seam: the single in-process
``least_squares`` call of ``core/capture_volume.py:387-411``."""
import numpy as np
import pytest

from caliscope_amd.engine import BAProblem
from caliscope_amd.least_squares import least_squares
from tests.helpers import aligned_difference, small_problem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _built():
    from caliscope_amd import build
    from caliscope_amd.hip_engine import require_device

    build.build(verbose=False)
    require_device()


def _problem(**kw):
    loss = kw.get("loss", "linear")
    sc, par, x0 = small_problem(**kw)
    fs = sc.f_scale_1px() * 2.0 if loss != "linear" else 1.0
    return par, BAProblem(par, sc.camera_indices, sc.image_coords, sc.obj_indices, loss=loss, f_scale=fs), x0, fs, sc


def _single(par, prob, x0, fs, **kw):
    return least_squares(None, x0, jac=None, bounds=par.bounds(), x_scale="jac", loss=prob.loss, f_scale=fs, method="trf",
                         args=(par, prob.camera_indices, prob.image_coords, prob.obj_indices), **kw)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", [dict(n_cams=8, n_points=900, k=8), dict(n_cams=24, n_points=1500, k=10),
                                  dict(n_cams=8, n_points=800, k=8, loss="huber", outliers=0.05)])
def test_ranks_on_one_device_match_the_single_rank_solve(world, case):
    from caliscope_amd.distributed import solve_multi_device

    par, prob, x0, fs, _ = _problem(**case)
    tol = dict(ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=200)
    one = _single(par, prob, x0, fs, **tol)
    many = solve_multi_device(prob, x0, [0] * world, backend="direct", **tol)
    assert many.status == one.status
    assert abs(many.cost - one.cost) <= 1e-9 * one.cost
    assert abs(many.nfev - one.nfev) <= 5  # the sums are formed in another order: at 1e-12 tolerances the last tiny steps differ
    pos, ang, scale = aligned_difference(par, many.x, one.x)
    assert pos < 1e-7 and ang < 1e-7 and abs(scale - 1) < 1e-4, (pos, ang, scale)  # the overall scale is a gauge direction: it drifts with rounding


def test_least_squares_shards_when_devices_are_named(monkeypatch):
    """The drop-in seam itself: ``devices=`` / ``CALISCOPE_HIP_DEVICES`` route the reference's call to the sharded solve."""
    par, prob, x0, fs, _ = _problem(n_cams=8, n_points=700, k=8)
    one = _single(par, prob, x0, fs)
    monkeypatch.setenv("CBA_XCHG", "direct")
    two = _single(par, prob, x0, fs, devices=[0, 0])
    monkeypatch.setenv("CALISCOPE_HIP_DEVICES", "0,0,0")
    three = _single(par, prob, x0, fs)
    for res in (two, three):
        assert res.status == one.status and abs(res.cost - one.cost) <= 1e-8 * one.cost
        pos, ang, _ = aligned_difference(par, res.x, one.x)
        assert pos < 1e-6 and ang < 1e-6


def test_free_intrinsics_sharded(monkeypatch):
    """Bounded variant (Coleman-Li scaling of the camera block) on two ranks: the primitives route, three exchanges per step."""
    from caliscope_amd.distributed import solve_multi_device

    par, prob, x0, fs, _ = _problem(n_cams=6, n_points=600, k=6, refine=True)
    lb, ub = par.bounds()
    tol = dict(ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=200)
    one = _single(par, prob, x0, fs, **tol)
    two = solve_multi_device(prob, x0, [0, 0], backend="direct", **tol)
    assert abs(two.cost - one.cost) <= 1e-8 * one.cost
    ncp = par.n_camera_params
    assert np.all(two.x[:ncp] > lb[:ncp]) and np.all(two.x[:ncp] < ub[:ncp])
    np.testing.assert_allclose(two.x[:ncp].reshape(-1, 9)[:, 6:], one.x[:ncp].reshape(-1, 9)[:, 6:], rtol=0, atol=1e-6)


def test_constraint_rows_sharded():
    """Rigid-distance rows: components stay on one rank, the Woodbury correction lands in that rank's share of S | b."""
    from caliscope_amd.distributed import solve_multi_device
    from tests.constrained_scene import board_scene

    sc = board_scene(n_frames=12)
    ga, gb, d, w = sc["constraints"]
    par = sc["par"]
    prob = BAProblem(par, sc["cam"], sc["uv"], sc["obj"], constraint_groups_a=ga, constraint_groups_b=gb, constraint_distances=d,
                     constraint_weights=w)
    tol = dict(ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=100)
    one = least_squares(None, sc["x0"], jac=None, bounds=par.bounds(), x_scale="jac", method="trf",
                        args=(par, sc["cam"], sc["uv"], sc["obj"], ga, gb, d, w), **tol)
    two = solve_multi_device(prob, sc["x0"], [0, 0], backend="direct", **tol)
    assert abs(two.cost - one.cost) <= 1e-8 * one.cost
    pos, ang, scale = aligned_difference(par, two.x, one.x)
    assert pos < 1e-6 and ang < 1e-6 and abs(scale - 1) < 1e-4


def test_a_failing_rank_releases_the_others():
    """A member that fails outside the library (here: a shard without observations) must not leave its peers spinning."""
    from caliscope_amd.distributed import solve_multi_device

    par, prob, x0, fs, sc = _problem(n_cams=4, n_points=40, k=4)
    keep = sc.obj_indices < 3  # three observed points: the fourth rank owns nothing
    tiny = BAProblem(par, sc.camera_indices[keep], sc.image_coords[keep], sc.obj_indices[keep])
    with pytest.raises(ValueError, match="owns no observations"):
        solve_multi_device(tiny, x0, [0, 0, 0, 0], backend="direct", max_nfev=5)


def test_rccl_refuses_repeated_devices():
    from caliscope_amd.distributed import solve_multi_device

    par, prob, x0, fs, _ = _problem(n_cams=4, n_points=60, k=4)
    with pytest.raises(ValueError, match="distinct devices"):
        solve_multi_device(prob, x0, [0, 0], backend="rccl")
