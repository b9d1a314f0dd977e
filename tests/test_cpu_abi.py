"""The whole boundary without a GPU: tests/native/cpu_library.cpp is a CPU TEST BUILD of include/caliscope_ba.h (same symbols,
same structs, dense arithmetic from csrc/ba_math.h, csrc/cba_solve.cpp compiled in).  Pointing CALISCOPE_BA_LIB at it lets the
CPU suite drive ctypes marshalling -> HipEngine -> least_squares seam -> CaptureVolume.optimize() end to end.

It is test infrastructure: the product never looks for it (caliscope_amd._lib loads the HIP library unless the variable is set),
and each case runs in its OWN interpreter so the library this process has loaded — the real one — is never swapped."""
import json
import os
import re
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def cpu_lib(tmp_path_factory):
    out = tmp_path_factory.mktemp("cpuabi") / "libcaliscope_ba_cpu.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", "-I", str(ROOT / "include"),
                    str(ROOT / "tests" / "native" / "cpu_library.cpp"), str(ROOT / "caliscope_amd" / "csrc" / "cba_solve.cpp"),
                    "-o", str(out)], check=True)
    return out


def _run(cpu_lib, body: str, timeout=240):
    """Run `body` (which prints one JSON object as its last line) in a fresh interpreter bound to the CPU build."""
    env = dict(os.environ, CALISCOPE_BA_LIB=str(cpu_lib), PYTHONPATH=str(ROOT))
    proc = subprocess.run([sys.executable, "-c", textwrap.dedent(body)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert proc.returncode == 0, proc.stderr[-3000:]
    return json.loads(proc.stdout.strip().splitlines()[-1])


def test_cpu_build_exports_the_whole_header(cpu_lib):
    out = _run(cpu_lib, """
        import json, re
        from pathlib import Path
        from caliscope_amd import _lib
        lib = _lib.load()                      # binds every entry of SIGNATURES or raises
        header = Path("include/caliscope_ba.h").read_text()
        declared = set(re.findall(r"\\b(cba_[a-z_0-9]+)\\s*\\(", header)) - {"cba_problem"}
        missing = sorted(n for n in declared if not hasattr(lib, n))
        print(json.dumps(dict(missing=missing, version=lib.cba_version(), devices=lib.cba_device_count(),
                              timers=[lib.cba_timer_name(i).decode() for i in range(lib.cba_timer_count())])))
    """)
    assert out["missing"] == [] and out["version"] == 100 and out["devices"] == 1
    assert len(out["timers"]) == 13 and out["timers"][2] == "build" and out["timers"][12] == "exchange"


@pytest.mark.parametrize("case", ["locked", "refine", "huber", "fisheye_mixed"])
def test_engine_hooks_match_the_oracle(cpu_lib, case):
    """Residual rows, cost, U/V blocks and gradient through HipEngine — the comparison tests/test_gpu_parity.py makes on the device."""
    out = _run(cpu_lib, f"""
        import json
        import numpy as np
        from scipy.optimize._lsq.common import scale_for_robust_loss_function
        from scipy.optimize._lsq.least_squares import construct_loss_function
        from caliscope_amd.bundle_parameterization import BundleParameterization
        from caliscope_amd.engine import BAProblem
        from caliscope_amd.hip_engine import HipEngine
        from oracle.residuals import joint_jacobian, joint_residuals
        from tests.helpers import small_problem
        case = {case!r}
        loss, fs = "linear", 1.0
        if case == "fisheye_mixed":
            from tests.test_oracle_pins import _mixed_arrays
            ca, points, uv, cam, obj = _mixed_arrays()
            par = BundleParameterization.from_camera_array(ca, n_points=len(points), refine_intrinsics=True)
            x0 = par.pack(ca, points)
        else:
            loss = "huber" if case == "huber" else "linear"
            sc, par, x0 = small_problem(n_cams=4, n_points=60, k=4, refine=(case == "refine"), loss=loss, outliers=0.05 if case == "huber" else 0.0)
            cam, uv, obj = sc.camera_indices, sc.image_coords, sc.obj_indices
            fs = sc.f_scale_1px() * 2.0 if case == "huber" else 1.0
        eng = HipEngine(BAProblem(par, cam, uv, obj, loss=loss, f_scale=fs))
        r, cost = eng.residuals(x0)
        f = joint_residuals(x0, par, cam, uv, obj)
        dr = float(np.abs(r - f).max())
        J = joint_jacobian(x0, par, cam, uv, obj).tocsr()
        if loss == "linear":
            cost_ref = 0.5 * float(f @ f)
        else:
            fn = construct_loss_function(len(f), loss, fs)
            cost_ref = float(fn(f, cost_only=True))
            J, f = scale_for_robust_loss_function(J, f, fn(f))
        H = (J.T @ J).toarray()
        g = np.asarray(J.T @ f).ravel()
        U, V, gc, gp = eng.normal_blocks(x0)
        ncp, P = par.n_camera_params, par.n_points
        dU = max(np.abs(U[c, :b.n_params, :b.n_params] - H[o:o + b.n_params, o:o + b.n_params]).max()
                 for c, (b, o) in enumerate(zip(par.blocks, par.camera_param_offsets)))
        Vref = np.stack([H[ncp + 3 * np.arange(P) + a, ncp + 3 * np.arange(P) + b] for a, b in ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2))], axis=1)
        print(json.dumps(dict(dr=dr, dcost=float(abs(cost - cost_ref) / cost_ref), dU=float(dU / np.abs(H).max()), dV=float(np.abs(V - Vref).max() / np.abs(H).max()),
                              dg=float(np.abs(np.concatenate([gc, gp.reshape(-1)]) - g).max() / np.abs(g).max()))))
    """)
    assert out["dr"] < 1e-11 and out["dcost"] < 1e-12 and out["dU"] < 1e-11 and out["dV"] < 1e-11 and out["dg"] < 1e-11, out


def test_least_squares_seam_solves_like_scipy(cpu_lib):
    out = _run(cpu_lib, """
        import json
        import numpy as np
        from caliscope_amd.least_squares import least_squares
        from oracle.solver import optimize_scipy, rms_reprojection_px
        from tests.helpers import aligned_difference, small_problem
        res = {}
        for name, kw in dict(locked=dict(), refine=dict(refine=True), huber=dict(loss="huber", outliers=0.05)).items():
            loss = kw.get("loss", "linear")
            sc, par, x0 = small_problem(n_cams=4, n_points=80, k=4, **kw)
            fs = sc.f_scale_1px() * 2.0 if loss != "linear" else 1.0
            a = (par, sc.camera_indices, sc.image_coords, sc.obj_indices)
            ref = optimize_scipy(*a, x0, loss=loss, f_scale=fs)
            got = least_squares(None, x0, args=a + (None, None, None, None), x_scale="jac", loss=loss, f_scale=fs, bounds=par.bounds(), method="trf")
            pos, ang, _ = aligned_difference(par, got.x, ref.x)
            res[name] = dict(status=int(got.status), nfev=int(got.nfev), ref_nfev=int(ref.nfev), rel_cost=float((got.cost - ref.cost) / ref.cost),
                             d_rms=float(rms_reprojection_px(*a, got.x) - rms_reprojection_px(*a, ref.x)), pos=float(pos), ang=float(ang))
        print(json.dumps(res))
    """)
    for name, r in out.items():
        assert r["status"] > 0, (name, r)
        if name == "huber":  # scipy's inexact lsmr steps wander for hundreds of evaluations here; the exact step must end at least as low
            assert r["rel_cost"] < 1e-6 and abs(r["d_rms"]) < 1e-2 and r["nfev"] <= r["ref_nfev"], (name, r)
            continue
        assert abs(r["nfev"] - r["ref_nfev"]) <= 3, (name, r)
        assert abs(r["rel_cost"]) < 1e-6 and abs(r["d_rms"]) < 1e-4, (name, r)
        assert r["pos"] < 1e-4 and r["ang"] < 1e-4, (name, r)


def test_capture_volume_optimize_real_session_on_the_cpu_build(cpu_lib):
    """BASELINE.json configs[0] — the reference's 4-camera ChArUco session (tests/test_capture_volume.py:354-415 there) — through
    CaptureVolume.optimize() -> filter -> optimize(), with the C ABI underneath."""
    out = _run(cpu_lib, """
        import json
        from pathlib import Path
        from caliscope_amd.bundle_parameterization import BundleParameterization
        from caliscope_amd.cameras import CameraArray
        from caliscope_amd.capture_volume import CaptureVolume
        from caliscope_amd.point_data import ImagePoints, WorldPoints
        from oracle.solver import optimize_scipy, rms_reprojection_px
        d = Path("tests/golden/post_optimization")
        cv = CaptureVolume(CameraArray.from_toml(d / "camera_array.toml"), ImagePoints.from_csv(d / "xy_CHARUCO.csv"), WorldPoints.from_csv(d / "xyz_CHARUCO.csv"))
        r0 = cv.reprojection_report.overall_rmse
        opt = cv.optimize()
        st = opt.optimization_status
        _, cam, uv, obj = cv._matched_arrays()
        par = BundleParameterization.from_camera_array(cv.camera_array, n_points=len(cv.world_points), refine_intrinsics=False)
        ref = optimize_scipy(par, cam, uv, obj, par.pack(cv.camera_array, cv.world_points.points))
        filt = opt.filter_by_percentile_error(50.0)
        print(json.dumps(dict(r0=r0, r1=opt.reprojection_report.overall_rmse, ref=rms_reprojection_px(par, cam, uv, obj, ref.x), converged=bool(st.converged),
                              cost=st.final_cost, ref_cost=float(ref.cost), r2=filt.reprojection_report.overall_rmse, r3=filt.optimize().reprojection_report.overall_rmse)))
    """, timeout=600)
    assert out["converged"] and abs(out["r0"] - 1.6625073265) < 1e-6
    assert abs(out["r1"] - out["ref"]) < 1e-4 and out["r1"] < out["r0"]
    assert abs(out["cost"] - out["ref_cost"]) <= 1e-6 * out["ref_cost"]
    assert out["r3"] <= out["r2"] < out["r1"]


def test_constraint_rows_through_the_cpu_build(cpu_lib):
    """cba_set_constraints: residual rows, one damped step and the converged solve against the oracle / scipy (tests/test_constraints.py on the device)."""
    out = _run(cpu_lib, """
        import json
        import numpy as np
        from caliscope_amd.engine import BAProblem
        from caliscope_amd.hip_engine import HipEngine
        from caliscope_amd.least_squares import least_squares
        from oracle.engine import OracleEngine
        from oracle.residuals import joint_jacobian, joint_residuals
        from oracle.solver import optimize_scipy
        from tests.constrained_scene import board_scene
        from tests.helpers import aligned_difference
        sc = board_scene()
        par, x0 = sc["par"], sc["x0"]
        ga, gb, dist, w = sc["constraints"]
        eng = HipEngine(BAProblem(par, sc["cam"], sc["uv"], sc["obj"], constraint_groups_a=ga, constraint_groups_b=gb, constraint_distances=dist, constraint_weights=w))
        r, _ = eng.residuals(x0)
        dr = float(np.abs(r - joint_residuals(x0, par, sc["cam"], sc["uv"], sc["obj"], ga, gb, dist, w)).max())
        ora = OracleEngine(par, sc["cam"], sc["uv"], sc["obj"], constraints=sc["constraints"])
        eng.begin(x0); ora.begin(x0); eng.linearize(); ora.linearize()
        a, b = eng.newton_step(1e-3), ora.newton_step(1e-3)
        ds = float(np.abs(eng.get_vector(3) - ora.s).max() / np.abs(ora.s).max())
        S, rhs = eng.reduced_system()
        dS = float(np.abs(S - S.T).max() / np.abs(S).max())
        sc_step = np.linalg.solve(S, rhs)
        dsc = float(np.abs(sc_step - eng.get_vector(3)[: par.n_camera_params]).max() / np.abs(sc_step).max())
        ref = optimize_scipy(par, sc["cam"], sc["uv"], sc["obj"], x0, constraints=sc["constraints"])
        res = least_squares(joint_residuals, x0, args=(par, sc["cam"], sc["uv"], sc["obj"], ga, gb, dist, w), jac=joint_jacobian, x_scale="jac", method="trf", bounds=par.bounds())
        pos, ang, scale = aligned_difference(par, res.x, ref.x)
        print(json.dumps(dict(dr=dr, ds=ds, dS=dS, dsc=dsc, ok=bool(a.ok and b.ok), status=int(res.status), rel_cost=float(abs(res.cost - ref.cost) / ref.cost), pos=float(pos), ang=float(ang),
                              scale=float(scale))))
    """)
    assert out["ok"] and out["dr"] < 1e-12 and out["ds"] < 1e-8 and out["dS"] < 1e-12 and out["dsc"] < 1e-8, out
    assert out["status"] > 0 and out["rel_cost"] < 1e-8 and out["pos"] < 1e-6 and out["ang"] < 1e-6 and abs(out["scale"] - 1) < 1e-6, out


def test_engine_cache_and_set_loss_on_the_cpu_build(cpu_lib):
    """Two least_squares calls on the same observations: the second reuses the first call's handle and only changes the loss."""
    out = _run(cpu_lib, """
        import json
        import numpy as np
        from caliscope_amd import engine_cache
        from caliscope_amd.least_squares import least_squares
        from oracle.solver import optimize_scipy
        from tests.helpers import small_problem
        sc, par, x0 = small_problem(n_cams=4, n_points=80, k=4, loss="huber", outliers=0.05)
        a = (par, sc.camera_indices, sc.image_coords, sc.obj_indices)
        fs = sc.f_scale_1px() * 2.0
        kw = dict(x_scale="jac", bounds=par.bounds(), method="trf")
        first = least_squares(None, x0, args=a + (None,) * 4, **kw)
        robust = least_squares(None, first.x, args=a + (None,) * 4, loss="huber", f_scale=fs, **kw)
        ref = optimize_scipy(*a, first.x, loss="huber", f_scale=fs)
        keep = np.arange(len(sc.camera_indices)) % 5 != 0
        sub = least_squares(None, x0, args=(par, sc.camera_indices[keep], sc.image_coords[keep], sc.obj_indices[keep]) + (None,) * 4, **kw)
        print(json.dumps(dict(stats=engine_cache.stats, rel=float(abs(robust.cost - ref.cost) / ref.cost), ok=bool(first.status > 0 and robust.status > 0 and sub.status > 0))))
    """)
    assert out["ok"] and out["stats"] == {"hits": 1, "misses": 2} and out["rel"] < 1e-6, out


# ---- more than one rank: the in-process group of the CPU build (host threads, host all-reduce) -------------------------------
# The same Python that drives a multi-GPU node — caliscope_amd.distributed.solve_multi_device: shard by point, one thread per
# rank, HipEngine.group_join, csrc/cba_solve.cpp on every rank, gather — against cba_group_* of the CPU build.  SURVEY.md 8e.

_SHARD_PRELUDE = """
    import json
    import numpy as np
    from caliscope_amd.engine import BAProblem
    from caliscope_amd.least_squares import least_squares
    from caliscope_amd.distributed import solve_multi_device
    from tests.helpers import aligned_difference, small_problem

    def problem(**kw):
        loss = kw.get("loss", "linear")
        sc, par, x0 = small_problem(**kw)
        fs = sc.f_scale_1px() * 2.0 if loss != "linear" else 1.0
        return par, BAProblem(par, sc.camera_indices, sc.image_coords, sc.obj_indices, loss=loss, f_scale=fs), x0, fs, sc

    def single(par, prob, x0, fs, **kw):
        return least_squares(None, x0, jac=None, bounds=par.bounds(), x_scale="jac", loss=prob.loss, f_scale=fs, method="trf",
                             args=(par, prob.camera_indices, prob.image_coords, prob.obj_indices), **kw)
"""


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", [dict(n_cams=4, n_points=90, k=4), dict(n_cams=5, n_points=80, k=4, loss="huber", outliers=0.05)],
                         ids=["linear", "huber"])
def test_ranks_of_a_host_group_match_the_single_rank_solve(cpu_lib, world, case):
    out = _run(cpu_lib, _SHARD_PRELUDE + f"""
    par, prob, x0, fs, _ = problem(**{case!r})
    tol = dict(ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=200)
    one = single(par, prob, x0, fs, **tol)
    many = solve_multi_device(prob, x0, [0] * {world}, backend="direct", **tol)
    pos, ang, scale = aligned_difference(par, many.x, one.x)
    print(json.dumps(dict(status=[int(one.status), int(many.status)], rel=float(abs(many.cost - one.cost) / one.cost),
                          nfev=[int(one.nfev), int(many.nfev)], pos=float(pos), ang=float(ang), scale=float(scale))))
    """)
    assert out["status"][0] == out["status"][1] and out["rel"] <= 1e-9, out
    assert abs(out["nfev"][0] - out["nfev"][1]) <= 5, out  # the sums are formed in another order
    assert out["pos"] < 1e-7 and out["ang"] < 1e-7 and abs(out["scale"] - 1) < 1e-4, out


def test_the_seam_shards_over_a_host_group(cpu_lib):
    """``devices=`` and CALISCOPE_HIP_DEVICES at the least_squares seam (what CaptureVolume.optimize() calls)."""
    out = _run(cpu_lib, _SHARD_PRELUDE + """
    import os
    par, prob, x0, fs, _ = problem(n_cams=4, n_points=70, k=4)
    one = single(par, prob, x0, fs)
    os.environ["CBA_XCHG"] = "direct"
    two = single(par, prob, x0, fs, devices=[0, 0])
    os.environ["CALISCOPE_HIP_DEVICES"] = "0,0,0"
    three = single(par, prob, x0, fs)
    rows = []
    for res in (two, three):
        pos, ang, _ = aligned_difference(par, res.x, one.x)
        rows.append(dict(same=bool(res.status == one.status), rel=float(abs(res.cost - one.cost) / one.cost), pos=float(pos), ang=float(ang)))
    print(json.dumps(rows))
    """)
    for row in out:
        assert row["same"] and row["rel"] <= 1e-8 and row["pos"] < 1e-6 and row["ang"] < 1e-6, out


def test_bounded_and_constrained_solves_on_a_host_group(cpu_lib):
    """Free intrinsics (the Coleman-Li route of cba_solve: camera state, scaling, step and trial exchanged per primitive) and
    rigid-distance rows (components stay on one rank) on two ranks."""
    out = _run(cpu_lib, _SHARD_PRELUDE + """
    from tests.constrained_scene import board_scene
    tol = dict(ftol=1e-12, xtol=1e-12, gtol=1e-12, max_nfev=200)
    par, prob, x0, fs, _ = problem(n_cams=4, n_points=90, k=4, refine=True)
    lb, ub = par.bounds()
    one = single(par, prob, x0, fs, **tol)
    two = solve_multi_device(prob, x0, [0, 0], backend="direct", **tol)
    ncp = par.n_camera_params
    intr = float(np.abs(two.x[:ncp].reshape(-1, 9)[:, 6:] - one.x[:ncp].reshape(-1, 9)[:, 6:]).max())
    free = dict(rel=float(abs(two.cost - one.cost) / one.cost), inside=bool(np.all(two.x[:ncp] > lb[:ncp]) and np.all(two.x[:ncp] < ub[:ncp])), intr=intr)

    sc = board_scene(n_cams=4, n_frames=4)
    ga, gb, d, w = sc["constraints"]
    par = sc["par"]
    prob = BAProblem(par, sc["cam"], sc["uv"], sc["obj"], constraint_groups_a=ga, constraint_groups_b=gb, constraint_distances=d, constraint_weights=w)
    one = least_squares(None, sc["x0"], jac=None, bounds=par.bounds(), x_scale="jac", method="trf", args=(par, sc["cam"], sc["uv"], sc["obj"], ga, gb, d, w), **tol)
    two = solve_multi_device(prob, sc["x0"], [0, 0], backend="direct", **tol)
    pos, ang, scale = aligned_difference(par, two.x, one.x)
    print(json.dumps(dict(free=free, con=dict(rel=float(abs(two.cost - one.cost) / one.cost), pos=float(pos), ang=float(ang), scale=float(scale)))))
    """)
    assert out["free"]["rel"] <= 1e-8 and out["free"]["inside"] and out["free"]["intr"] < 1e-6, out
    assert out["con"]["rel"] <= 1e-8 and out["con"]["pos"] < 1e-6 and out["con"]["ang"] < 1e-6 and abs(out["con"]["scale"] - 1) < 1e-4, out


def test_a_failing_rank_releases_the_host_group(cpu_lib):
    """One rank fails before joining (its shard has no observations): the others must come back with an error, not wait."""
    out = _run(cpu_lib, _SHARD_PRELUDE + """
    par, prob, x0, fs, sc = problem(n_cams=4, n_points=40, k=4)
    keep = sc.obj_indices < 3
    tiny = BAProblem(par, sc.camera_indices[keep], sc.image_coords[keep], sc.obj_indices[keep])
    try:
        solve_multi_device(tiny, x0, [0, 0, 0, 0], backend="direct", max_nfev=5)
        msg = "no error"
    except ValueError as e:
        msg = str(e)
    print(json.dumps(dict(msg=msg)))
    """, timeout=60)
    assert "owns no observations" in out["msg"]


def test_unsupported_entries_fail_loudly(cpu_lib):
    out = _run(cpu_lib, """
        import json
        import numpy as np
        from caliscope_amd import _lib
        from caliscope_amd.exceptions import BackendError
        lib = _lib.load()
        rc = lib.cba_comm_init(None, None, 1, 2)
        msg = _lib.last_error(lib)
        import ctypes as C
        g = C.c_void_p()
        rc2 = lib.cba_group_create(17, C.byref(g))
        rc3 = lib.cba_triangulate(None, 0, None, None)
        print(json.dumps(dict(rc=rc, msg=msg, rc2=rc2, rc3=rc3)))
    """)
    # (cba_comm_init joins processes in this build — the launcher test below — so a null handle is a plain argument error)
    assert out["rc"] == -1 and "bad arguments" in out["msg"] and out["rc2"] == -1 and out["rc3"] == -4


# ---- bench.py --gpus N started plainly: the ranks run inside the process (the reference's solve is one in-process call) --------
def test_bench_starts_its_own_ranks_without_a_launcher(cpu_lib):
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment: bench.main() starts two rank threads, joins their
    engines (here: the device group of the CPU build, both ranks on "device" 0) and prints one contract line; the sharded run ends at
    the single-rank run's solution."""
    out = _run(cpu_lib, """
        import io, json, os, sys
        from contextlib import redirect_stdout
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            os.environ.pop(k, None)
        import bench
        lines = {}
        for name, argv in (("two", ["--gpus", "2", "--devices", "0,0", "--xchg", "direct"]), ("one", ["--gpus", "1"])):
            lines[name] = json.loads(bench.main(argv + ["--workload", "tiny", "--steps", "4", "--warmup", "1", "--no-cpu", "--also", ""]))
        try:
            bench.main(["--gpus", "3", "--workload", "tiny", "--no-cpu", "--also", ""])
            refused = ""
        except SystemExit as exc:
            refused = str(exc)
        print(json.dumps(dict(two=lines["two"], one=lines["one"], refused=refused)))
    """)
    two, one = out["two"], out["one"]
    for d in (two, one):
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                    "config", "rccl_ranks", "comm_ms_per_step", "setup_ms"):
            assert key in d, key
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["rccl_ranks"] == 0 and "2 ranks in one process" in two["config"]["parallelism"]
    assert two["config"]["n_obs_total"] == one["config"]["n_obs_total"] == 450 and "obs per rank" in two["config"]["parallelism"]
    assert one["n_gpus"] == 1 and abs(two["final_rms_px"] - one["final_rms_px"]) < 1e-6 and two["solve"]["status"] > 0
    assert abs(two["value"] - 450 / (two["ms_per_step"] * 1e-3)) < 1e-2 * two["value"]
    assert "1 HIP device" in out["refused"] and "--gpus 3" in out["refused"]  # fewer devices than ranks: a clear message, no hang


def test_bench_under_the_launcher_one_process_per_rank(cpu_lib):
    """The driver's multi-GPU command — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N ...` — with two ranks: RANK / WORLD_SIZE / LOCAL_RANK from the launcher, the host-side control plane finds its own port
    (MASTER_PORT belongs to the launcher's store), rank 0's communicator id travels over it, every rank solves its shard, rank 0 prints the one
    line.  The CPU build's cba_comm_* stands in for RCCL (a unix-domain socket between the processes); everything above the C ABI is the code
    that runs on the GPUs."""
    pytest.importorskip("torch")
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, CALISCOPE_BA_LIB=str(cpu_lib), PYTHONPATH=str(ROOT))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CBA_CONTROL_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(ROOT / "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "4", "--warmup", "1", "--no-cpu", "--also", ""]
    proc = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["rccl_ranks"] == 2 and d["steps"] == 4
    assert "2 processes (launcher)" in d["config"]["parallelism"] and d["config"]["n_obs_total"] == 450
    assert abs(d["value"] - 450 / (d["ms_per_step"] * 1e-3)) < 1e-2 * d["value"]
    assert d["solve"]["status"] > 0 and abs(d["final_rms_px"] - 0.5) < 0.5  # converged on the sharded problem (the single-rank figure: the test above)
    single = json.loads(subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--workload", "tiny", "--steps", "4", "--warmup", "1", "--no-cpu",
                                        "--also", ""], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300, check=True).stdout.strip().splitlines()[-1])
    assert abs(d["final_rms_px"] - single["final_rms_px"]) < 1e-6 and d["solve"]["nfev"] == single["solve"]["nfev"]
