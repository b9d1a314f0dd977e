"""The per-observation arithmetic the HIP kernels inline (caliscope_amd/csrc/ba_math.h), compiled for the
host by g++ and compared with the numpy oracle — runs without a GPU."""
import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest
from scipy.optimize._lsq.least_squares import construct_loss_function

from caliscope_amd.bundle_parameterization import BundleParameterization
from oracle import camera_model as cm
from oracle.residuals import joint_jacobian, joint_residuals
from tests.helpers import small_problem
from tests.test_oracle_pins import _mixed_arrays

ROOT = Path(__file__).resolve().parent.parent
D = ctypes.POINTER(ctypes.c_double)


def _p(a):
    return a.ctypes.data_as(D)


@pytest.fixture(scope="module")
def mh(tmp_path_factory):
    out = tmp_path_factory.mktemp("mh") / "libmath_harness.so"
    subprocess.run(
        ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", f"-I{ROOT / 'caliscope_amd' / 'csrc'}",
         str(ROOT / "tests" / "native" / "math_harness.cpp"), "-o", str(out)], check=True)
    lib = ctypes.CDLL(str(out))
    lib.mh_chol3_solve.restype = ctypes.c_int
    lib.mh_robust.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_double, D]
    return lib


def _native_blocks(mh, par, x, cam_idx, uv, obj_idx):
    t = par.device_tables()
    ncp = par.n_camera_params
    pts = x[ncp:].reshape(-1, 3)
    n = len(cam_idx)
    E, A, B = np.zeros((n, 2)), np.zeros((n, 2, 9)), np.zeros((n, 2, 3))
    for i in range(n):
        c = int(cam_idx[i])
        off = par.camera_param_offsets[c]
        xc = np.zeros(9)
        xc[: t["cam_n_params"][c]] = x[off : off + t["cam_n_params"][c]]
        cc = np.ascontiguousarray(t["cam_const"][c])
        X = np.ascontiguousarray(pts[obj_idx[i]])
        o = np.ascontiguousarray(uv[i])
        mh.mh_project_full(_p(xc), _p(cc), int(t["cam_model"][c]), int(t["cam_n_params"][c]), _p(X), _p(o),
                           _p(E[i]), _p(A[i].reshape(-1)), _p(B[i].reshape(-1)))
        Ef, Af, Bf = np.zeros(2), np.zeros((2, 9)), np.zeros((2, 3))  # the factored form of the same blocks (round 6)
        mh.mh_project_factors(_p(xc), _p(cc), int(t["cam_model"][c]), int(t["cam_n_params"][c]), _p(X), _p(o), _p(Ef), _p(Af.reshape(-1)), _p(Bf.reshape(-1)))
        assert np.allclose(Ef, E[i], rtol=0, atol=1e-15) and np.allclose(Bf, B[i], rtol=1e-13, atol=1e-15)
        assert np.allclose(Af, A[i], rtol=1e-12, atol=1e-14 * max(1.0, np.abs(A[i]).max()))
        e2 = np.zeros(2)
        mh.mh_project_residual(_p(xc), _p(cc), int(t["cam_model"][c]), int(t["cam_n_params"][c]), _p(X), _p(o), _p(e2))
        assert np.allclose(e2, E[i], rtol=0, atol=1e-15)
    return E, A, B


def _check_against_oracle(mh, par, x, cam_idx, uv, obj_idx):
    E, A, B = _native_blocks(mh, par, x, cam_idx, uv, obj_idx)
    r = joint_residuals(x, par, cam_idx, uv, obj_idx).reshape(-1, 2)
    J = joint_jacobian(x, par, cam_idx, uv, obj_idx).toarray()
    scale = max(1.0, np.abs(r).max())
    assert np.abs(E - r).max() < 1e-12 * scale
    ncp = par.n_camera_params
    jscale = np.abs(J).max()
    for i in range(len(cam_idx)):
        c = int(cam_idx[i])
        off, nb = par.camera_param_offsets[c], par.blocks[c].n_params
        Ja = J[2 * i : 2 * i + 2, off : off + nb]
        Jb = J[2 * i : 2 * i + 2, ncp + 3 * obj_idx[i] : ncp + 3 * obj_idx[i] + 3]
        assert np.abs(A[i][:, :nb] - Ja).max() < 1e-11 * jscale, (i, c)
        assert np.abs(B[i] - Jb).max() < 1e-11 * jscale


@pytest.mark.parametrize("refine", [False, True])
def test_blocks_match_oracle_pinhole(mh, refine):
    sc, par, x0 = small_problem(n_cams=5, n_points=60, k=4, refine=refine)
    _check_against_oracle(mh, par, x0, sc.camera_indices, sc.image_coords, sc.obj_indices)


def test_blocks_match_oracle_mixed_fisheye(mh):
    ca, points, uv, cam_idx, obj_idx = _mixed_arrays()
    par = BundleParameterization.from_camera_array(ca, n_points=len(points), refine_intrinsics=True)
    _check_against_oracle(mh, par, par.pack(ca, points), cam_idx.astype(np.int32), uv, obj_idx)


def test_camera_table_rotation_small_and_large_angles(mh):
    cc = np.array([800.0, 790.0, 320.0, 240.0, 0.1, -0.05, 0.001, 0.002, 0.01, 0, 0, 0])
    for r in ([0, 0, 0], [1e-9, -2e-9, 1e-9], [1e-5, 2e-5, -3e-5], [0.3, -0.2, 0.5], [2.0, -1.5, 1.0], [0, 0, 3.1]):
        xc = np.array([*r, 0.1, 0.2, 0.3, 1.05, 0.12, -0.2], dtype=np.float64)
        out = np.zeros(48)
        mh.mh_cam_table(_p(xc), _p(cc), 0, 9, _p(out))
        assert np.allclose(out[:9].reshape(3, 3), cm.rodrigues(np.array(r, float)), atol=1e-15)
        assert np.isclose(out[21], 1.05 * 800.0) and np.isclose(out[22], 1.05 * 790.0)
        assert np.allclose(out[25:30], [0.12, -0.2, 0.001, 0.002, 0.01])
        # J_l: d(R X)/dr = -[R X]x J_l  vs the oracle's dR/dr
        Jl = out[12:21].reshape(3, 3)
        X = np.array([0.3, -0.4, 1.2])
        Y = cm.rodrigues(np.array(r, float)) @ X
        skew = np.array([[0, -Y[2], Y[1]], [Y[2], 0, -Y[0]], [-Y[1], Y[0], 0]])
        D3 = cm.rodrigues_jacobian(np.array(r, float))
        ref = np.stack([D3[j] @ X for j in range(3)], axis=1)
        # the oracle's closed form divides by theta^2 and loses digits below ~1e-6 rad; the kernel uses a series there
        assert np.allclose(-skew @ Jl, ref, atol=1e-12 if np.linalg.norm(r) > 1e-3 or not np.any(r) else 1e-6)


@pytest.mark.parametrize("loss", ["linear", "huber", "soft_l1", "cauchy", "arctan"])
def test_robust_scaling_matches_scipy(mh, loss):
    fs = 0.0017
    r = np.concatenate([np.linspace(-0.02, 0.02, 41), [0.0, 1e-9, fs, -fs, 0.5]])
    if loss == "linear":
        rho0, js, rsc = r * r, np.ones_like(r), r
    else:
        fn = construct_loss_function(len(r), loss, fs)
        rho = fn(r.copy())
        rho0 = rho[0].copy()
        js = np.sqrt(np.maximum(rho[1] + 2 * rho[2] * r * r, np.finfo(float).eps))
        rsc = r * rho[1] / js
    for i, ri in enumerate(r):
        out = np.zeros(4)
        mh.mh_robust({"linear": 0, "huber": 1, "soft_l1": 2, "cauchy": 3, "arctan": 4}[loss], fs, float(ri), _p(out))
        assert np.isclose(out[0], rho0[i], rtol=1e-12, atol=1e-300)
        assert np.isclose(out[1], js[i], rtol=1e-12)
        assert np.isclose(out[2], rsc[i], rtol=1e-12, atol=1e-300)
        assert np.isclose(out[3], rho0[i], rtol=1e-12, atol=1e-300)


def test_chol3(mh):
    rng = np.random.default_rng(0)
    for _ in range(100):
        M = rng.normal(size=(5, 3))
        V = M.T @ M + 1e-3 * np.eye(3)
        b = rng.normal(size=3)
        v6 = np.array([V[0, 0], V[0, 1], V[0, 2], V[1, 1], V[1, 2], V[2, 2]])
        x = np.zeros(3)
        assert mh.mh_chol3_solve(_p(v6), _p(b), _p(x)) == 1
        assert np.allclose(x, np.linalg.solve(V, b), rtol=1e-9)
    sing = np.array([1.0, 1.0, 0.0, 1.0, 0.0, 1.0])  # rank deficient
    assert mh.mh_chol3_solve(_p(sing), _p(np.ones(3)), _p(np.zeros(3))) == 0
