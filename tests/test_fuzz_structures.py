"""Randomised observation structures against the numpy engine: ragged visibility (points seen once, twice, by every
camera), rows repeated (the same camera observing a point again), heavy points beyond one chunk, points nobody sees,
shuffled input order, free intrinsics, robust losses, optional constraint rows — one evaluation, linearisation and
damped step each."""
import numpy as np
import pytest

from caliscope_amd.bundle_parameterization import BundleParameterization
from caliscope_amd.engine import BAProblem
from caliscope_amd.synthetic import make_scene

pytestmark = pytest.mark.gpu


def _random_problem(seed):
    rng = np.random.default_rng(seed)
    n_cams = int(rng.integers(3, 12))
    n_points = int(rng.integers(40, 400))
    refine = bool(rng.integers(0, 2))
    loss = ["linear", "linear", "soft_l1", "cauchy"][int(rng.integers(0, 4))]
    sc = make_scene(n_cams=n_cams, n_points=n_points, n_obs=n_points * min(n_cams, 3), refine=refine, seed=seed, loss=loss)
    # rebuild the observation list with a random structure: take each point's true projections in all cameras it was
    # generated for, then drop / repeat rows at random
    cam, uv, obj = sc.camera_indices, sc.image_coords, sc.obj_indices
    keep = rng.random(len(cam)) < 0.8
    cam, uv, obj = cam[keep], uv[keep], obj[keep]
    rep = rng.integers(0, len(cam), size=len(cam) // 5)  # repeated rows, jittered
    cam = np.concatenate([cam, cam[rep]]); obj = np.concatenate([obj, obj[rep]]); uv = np.vstack([uv, uv[rep] + rng.normal(0, 0.5, (len(rep), 2))])
    heavy = rng.integers(0, n_points, size=int(rng.integers(0, 3)))  # a few points observed hundreds of times
    for h in heavy:
        rows = np.flatnonzero(obj == h)
        if rows.size == 0:
            continue
        extra = rng.choice(rows, size=int(rng.integers(60, 600)))
        cam = np.concatenate([cam, cam[extra]]); obj = np.concatenate([obj, obj[extra]])
        uv = np.vstack([uv, uv[extra] + rng.normal(0, 0.5, (extra.size, 2))])
    order = rng.permutation(len(cam))
    cam, uv, obj = cam[order].astype(np.int32), np.ascontiguousarray(uv[order]), obj[order].astype(np.int32)
    seen = np.bincount(cam, minlength=n_cams)
    if seen.min() < 8:  # every camera needs a few rows for a well-posed block
        return None
    par = BundleParameterization.from_camera_array(sc.cameras_init, n_points=n_points, refine_intrinsics=refine)
    x0 = par.pack(sc.cameras_init, sc.points_init)
    con = None
    if rng.random() < 0.4:  # distance rows between random pairs of observed points, a few sharing points (components)
        observed = np.flatnonzero(np.bincount(obj, minlength=n_points) > 0)
        m = int(rng.integers(3, 25))
        a, b = rng.choice(observed, m), rng.choice(observed, m)
        ok = a != b
        a, b = a[ok], b[ok]
        if a.size:
            pts = sc.points_init
            ga, gb = np.repeat(a[:, None], 4, axis=1).astype(np.int32), np.repeat(b[:, None], 4, axis=1).astype(np.int32)
            dist = np.linalg.norm(pts[a] - pts[b], axis=1) * rng.uniform(0.97, 1.03, a.size)
            con = (ga, gb, dist, np.full(a.size, (1.0 / 1394.6) / 0.002))
    fs = sc.f_scale_1px() * 3.0 if loss != "linear" else 1.0
    return par, cam, uv, obj, x0, loss, fs, con


@pytest.mark.parametrize("seed", range(24))
def test_random_structure(seed):
    from caliscope_amd.hip_engine import HipEngine
    from oracle.engine import OracleEngine

    prob = _random_problem(1000 + seed)
    if prob is None:
        pytest.skip("degenerate draw")
    par, cam, uv, obj, x0, loss, fs, con = prob
    kw = {} if con is None else dict(constraint_groups_a=con[0], constraint_groups_b=con[1], constraint_distances=con[2], constraint_weights=con[3])
    hip = HipEngine(BAProblem(par, cam, uv, obj, loss=loss, f_scale=fs, **kw))
    ora = OracleEngine(par, cam, uv, obj, loss=loss, f_scale=fs, constraints=con)
    try:
        c_h, c_o = hip.begin(x0), ora.begin(x0)
        assert abs(c_h - c_o) <= 1e-12 * c_o
        hip.linearize(); ora.linearize()
        assert np.abs(hip.get_vector(2) - ora.g).max() <= 1e-10 * np.abs(ora.g).max()
        big = ora.scale_inv > 1e-4 * np.median(ora.scale_inv)  # columns at the sqrt(EPS) floor of a robust loss are rounding noise
        assert np.abs(hip.get_vector(4) - ora.scale_inv)[big].max() <= 1e-9 * np.abs(ora.scale_inv).max()
        sh, so = hip.newton_step(1e-2), ora.newton_step(1e-2)
        assert sh.ok and so.ok
        assert np.abs(hip.get_vector(3) - ora.s)[big].max() <= 1e-6 * np.abs(ora.s[big]).max()
        th, to = hip.trial(-1e-4, 0.7), ora.trial(-1e-4, 0.7)
        assert th.finite == to.finite and abs(th.step_norm - to.step_norm) <= 1e-9 * to.step_norm
        # a robust loss leaves sqrt(EPS)-floor columns: the gradient part of this step is then ~1e14 long and the cost at
        # such a point is decided by the last bits of the step (note in tests/test_gpu_parity.py)
        if to.step_norm < 1e6:
            assert abs(th.cost - to.cost) <= 1e-9 * to.cost
    finally:
        hip.close()
