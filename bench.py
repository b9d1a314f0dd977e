#!/usr/bin/env python
"""Headline benchmark: observations/sec per LM iteration (+ final RMS reprojection error, px).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg4] [--scaling strong|weak] [--no-cpu] [--also cfg2,cfg3,cfg5]

A *step* is one trust-region (LM) iteration of the hot path over the whole observation set: trial-point
evaluation plus — for accepted steps — linearisation (residuals, Jacobian blocks, J^T J / J^T r
accumulation) and the Schur-complement solve.  Steps are produced by running the solver from the
device-resident initial guess with the reference's default tolerances, restarting it when it converges, until
exactly K trial iterations have been executed (this is the reference's own normalisation,
``N_obs * (nfev - 1) / T_solve``).  Observations, initial guess and all solver state are
resident in HBM before the timed region starts; nothing but scalars crosses PCIe inside it.

Workload (``config.workload``): BASELINE.json's multi-GPU configuration cfg4 — 64 cameras / 200k points /
2M observations, extrinsics-only, linear loss.  With ``--gpus N`` the default is STRONG scaling, as BASELINE.json
words it ("2M obs sharded across 8 x MI355X"): the same 2M observations, their points partitioned over the N ranks
(caliscope_amd/sharding.py), the camera blocks and the reduced camera system all-reduced over RCCL each iteration;
``--scaling weak`` gives every rank its own 200k-point shard of the same 64 cameras instead.  The other configs (cfg2,
cfg3, and cfg5 = 128 cameras / 1M points / 10M observations with joint intrinsics, on one GPU) are run
untimed-by-the-driver after the headline and reported under ``also`` (cfg5 with its full roofline block).

``cpu_baseline`` is the reference's scipy call on a bounded sample of the workload, and the SAME sample is then solved
on the GPU: ``parity`` holds the differences (RMS px, cost, gauge-aligned poses / points) of the two solutions.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes(n_obs, n_points, n_cams, ncp, nct):
    """Per-launch algorithmic HBM bytes of each per-observation kernel.

    Storage model: observation record u,v f64 + cam,pt i32 = 24 B; point 24 B; V_p 48 B;
    g_p / scale / step 24 B each; packed camera block 8*(nc(nc+1)/2+nc) B."""
    ustride = nct * (nct + 1) // 2 + nct
    N, P = n_obs, n_points
    return {
        "cost": 24 * N + 24 * P,
        "build": 24 * N + 24 * P + 72 * P + 8 * n_cams * ustride,
        "jv": 24 * N + 24 * P + 24 * P,
        "schur": 24 * N + 24 * P + 48 * P + 24 * P + 24 * P + 8 * ncp * ncp + 8 * ncp,
        "backsub": 24 * N + 24 * P + 48 * P + 24 * P + 24 * P + 24 * P,
    }


def build_problem(name, seed=42, shard=0, **overrides):
    from caliscope_amd.bundle_parameterization import BundleParameterization
    from caliscope_amd.engine import BAProblem
    from caliscope_amd.synthetic import CONFIGS, make_config

    sc = make_config(name, seed=seed, shard=shard, **overrides)
    par = BundleParameterization.from_camera_array(sc.cameras_init, n_points=len(sc.points_init), refine_intrinsics=sc.refine_intrinsics)
    x0 = par.pack(sc.cameras_init, sc.points_init)
    f_scale = sc.f_scale_1px() if sc.loss != "linear" else 1.0
    prob = BAProblem(par, sc.camera_indices, sc.image_coords, sc.obj_indices, loss=sc.loss, f_scale=f_scale)
    return sc, par, x0, prob, CONFIGS[name]


def run_iterations(engine, k_steps, solve_kw):
    """Execute exactly k_steps trial iterations through cba_solve (the product's driver: the trust-region loop runs in
    the library, restarting from the x0 kept on the device); returns (n_solves, last_result)."""
    done, solves, last = 0, 0, None
    while done < k_steps:
        last = engine.solve(None, max_nfev=(k_steps - done) + 1, fetch_x=False, **solve_kw)
        done += last.nfev - 1
        solves += 1
        if last.nfev <= 1:  # already converged at x0: nothing to iterate on
            break
    return solves, last


def rms_px(engine, par, x, cam_idx):
    r, _ = engine.residuals(x)
    fx = np.array([b.fx_initial for b in par.blocks])[cam_idx]
    e = r.reshape(-1, 2) * fx[:, None]
    return float(np.sqrt(np.mean(np.sum(e * e, axis=1))))


class _Solo:
    rank, world = 0, 1

    def barrier(self):
        pass

    def allreduce_sum(self, a):
        return np.asarray(a, dtype=np.float64)

    def allreduce_max(self, v):
        return float(v)


def measure(name, steps, warmup, device_id=0, timers=True, seed=42, solve_kw=None, control=None, scaling="strong", **overrides):
    """Strong scaling: every rank generates the same scene and keeps the shard of points ``shard_problem`` gives it.
    Weak scaling: rank r draws its own points/observations of the same cameras.  Either way the engine all-reduces the
    camera blocks, the reduced camera system and the scalar sums over RCCL."""
    from caliscope_amd.hip_engine import HipEngine
    from caliscope_amd.sharding import shard_problem

    control = control or _Solo()
    solve_kw = dict(solve_kw or {})
    t_gen = time.time()
    n_total, n_params_global = None, None
    if scaling == "strong" and control.world > 1:
        sc, par, x0, prob, cfg = build_problem(name, seed=seed, shard=0, **overrides)
        n_total = prob.n_obs
        shard = shard_problem(prob, control.rank, control.world)
        x0 = shard.local_x(x0)
        n_params_global = int(prob.n_params)
        prob, par = shard.problem, shard.problem.parameterization
    else:
        sc, par, x0, prob, cfg = build_problem(name, seed=seed, shard=control.rank, **overrides)
    t_gen = time.time() - t_gen
    if par.has_finite_bounds:  # free intrinsics: the bounds of BundleParameterization.bounds(), as CaptureVolume.optimize passes them
        lb, ub = par.bounds()
        solve_kw.update(lb=np.ascontiguousarray(lb[: par.n_camera_params]), ub=np.ascontiguousarray(ub[: par.n_camera_params]))
    eng = HipEngine(prob, device_id=device_id)
    if control.world > 1:
        uid = control.broadcast_bytes(eng.comm_unique_id() if control.rank == 0 else None, 128)
        eng.comm_init(uid, control.rank, control.world)
    info = eng.info()
    eng.begin(x0)
    run_iterations(eng, max(warmup, 1), solve_kw)
    # timed region: exactly `steps` iterations, barrier + device sync on both sides, max over ranks
    control.barrier()
    t0 = time.perf_counter()
    solves, last = run_iterations(eng, steps, solve_kw)  # every engine call ends with a stream synchronize
    control.barrier()
    elapsed = control.allreduce_max(time.perf_counter() - t0)
    # instrumented repeat of the same `steps` iterations: HIP events around every kernel family on the engine's
    # stream.  Kept out of the region above because recording ~40 event pairs per iteration costs 6 % (cfg4) to
    # 50 % (cfg2) of the wall time, which would understate `value`.
    tm = {}
    if timers:
        eng.enable_timers(True)
        eng.reset_timers()
        run_iterations(eng, steps, solve_kw)
        tm = eng.timers()
        eng.enable_timers(False)
    # untimed: full solve for the accuracy figure (squared pixel errors summed over all shards)
    # (sharded: scipy's default evaluation cap 100 n refers to the whole problem, and every rank must use the same one)
    full = eng.solve(None, **solve_kw, **({"max_nfev": 100 * n_params_global} if n_params_global else {}))

    def rms_all(x):
        r, _ = eng.residuals(x)
        fx = np.array([b.fx_initial for b in par.blocks])[prob.camera_indices]
        e = r.reshape(-1, 2) * fx[:, None]
        tot = control.allreduce_sum(np.array([float(np.sum(e * e)), float(len(fx))]))
        return float(np.sqrt(tot[0] / tot[1]))

    rms, rms0 = rms_all(full.x), rms_all(x0)
    eng.close()
    return {
        "name": name, "n_obs": prob.n_obs, "n_obs_total": n_total if n_total is not None else prob.n_obs * control.world,
        "n_points": par.n_points, "n_cams": len(par.blocks), "ncp": par.n_camera_params, "full_njev": full.njev,
        "nct": 9 if any(b.n_params == 9 for b in par.blocks) else 6, "loss": prob.loss, "elapsed": elapsed, "steps": steps,
        "solves": solves, "timers": tm, "final_rms_px": rms, "initial_rms_px": rms0, "full_nfev": full.nfev,
        "full_status": full.status, "full_cost": full.cost, "info": info, "t_generate_s": t_gen,
        "scene": sc, "par": par, "x0": x0,
    }


# kernels of each timer family in the rocprofv3 summaries (the Schur pass is k_tprep + k_schur_reg3; k_schur_reg2 and
# k_schur_tile are the fallbacks)
PMC_KERNEL = {"schur": ("k_tprep", "k_schur_reg3", "k_schur_reg2", "k_schur_tile"), "build": ("k_build",), "jv": ("k_jv",),
              "backsub": ("k_backsub",), "cost": ("k_cost<false>",)}


def pmc_traffic(workload, family):
    """HBM bytes per launch of `family`'s kernel from the committed rocprofv3 PMC summary of this workload
    (profiles/pmc_<workload>.json: FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate --pmc passes); None if absent."""
    path = ROOT / "profiles" / f"pmc_{workload}.json"
    if not path.exists() or family not in PMC_KERNEL:
        return None
    rows = json.loads(path.read_text())
    found = [row["hbm_bytes"] for name, row in rows.items() if name.startswith(PMC_KERNEL[family])]
    return round(sum(found)) if found else None


def roofline_from(m):
    tm = m["timers"]
    if not tm:
        return None
    alg = algorithmic_bytes(m["n_obs"], m["n_points"], m["n_cams"], m["ncp"], m["nct"])
    fam = {k: v for k, v in tm.items() if k in alg and v[1] > 0}
    if not fam:
        return None
    dom = max(fam, key=lambda k: fam[k][0])
    ms, calls = fam[dom]
    avg_s = ms / calls * 1e-3
    achieved = alg[dom] / avg_s / 1e9
    table = {k: {"ms_total": round(v[0], 4), "launches": v[1], "avg_us": round(v[0] / v[1] * 1e3, 2),
                 **({"alg_bytes": alg[k], "GBps": round(alg[k] / (v[0] / v[1] * 1e-3) / 1e9, 1)} if k in alg else {})}
             for k, v in tm.items() if v[1] > 0}
    return {
        "bound": "hbm", "kernel": f"k_{dom}", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(m["name"], dom), "alg_bytes_per_launch": alg[dom],
        "note": "k_schur = the Schur pass (k_tprep writes one compact record per observation, k_schur_reg3 gathers them per "
                "camera-group tile and multiplies the pairs: DESIGN.md 4-5); durations from HIP events on the engine stream in an instrumented "
                "repeat of the timed steps; traffic from profiles/pmc_*.json (rocprofv3 --pmc, FETCH_SIZE x2 + WRITE_SIZE, "
                "summed over the kernels of the pass)",
        "avg_launch_us": round(avg_s * 1e6, 2), "kernels": table,
        # the co-bound SURVEY.md 8(d) names: FP64 vector rate of the pair products T_i T_j^T (2 * 3 nc^2 flop per pair) over the
        # duration of the Schur pass, against the 78.6 TFLOP/s FP64 vector peak
        "fp64_valu": (lambda fl, t: {"flops_per_launch": fl, "achieved_tflops": round(fl / t / 1e12, 2), "peak_tflops": 78.6,
                                     "frac": round(fl / t / 1e12 / 78.6, 4)})(
            int(m["info"]["schur_pairs"]) * 6 * m["nct"] ** 2, tm["schur"][0] / tm["schur"][1] * 1e-3) if "schur" in tm and tm["schur"][1] else None,
        # B_alg = 72 N + 312 P + 8 (nc C)^2 bytes per accepted LM iteration, over the measured step time
        "iteration": (lambda b, t: {"alg_bytes": b, "GBps": round(b / t / 1e9, 1), "frac": round(b / t / 1e9 / HBM_PEAK_GBS, 4)})(
            72 * m["n_obs"] + 312 * m["n_points"] + 8 * m["ncp"] ** 2, m["elapsed"] / max(m["steps"], 1)),
    }


def _similarity(src, dst):
    """Least-squares similarity (s, R, t), dst ~ s R src + t (Umeyama; the reference aligns solutions the same way,
    core/alignment.py:84-150) — BA leaves a 7-DoF gauge free, so two solvers' results are compared after alignment."""
    mu_s, mu_d = src.mean(0), dst.mean(0)
    a, b = src - mu_s, dst - mu_d
    U, sv, Vt = np.linalg.svd(b.T @ a / len(src))
    sign = np.ones(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        sign[2] = -1
    R = U @ np.diag(sign) @ Vt
    sc = float((sv * sign).sum() / (a * a).sum() * len(src))
    return sc, R, mu_d - sc * R @ mu_s


def solution_parity(par, x_a, x_b):
    """Gauge-aligned difference of two solutions: (max relative position difference of camera centres and points, max
    rotation angle between corresponding cameras [rad])."""
    from caliscope_amd.cameras import rvec_to_matrix

    def cams(x):
        cen, rot = [], []
        for off in par.camera_param_offsets:
            R = rvec_to_matrix(x[off:off + 3])
            rot.append(R)
            cen.append(-R.T @ x[off + 3:off + 6])
        return np.array(cen), np.array(rot)

    ca, Ra = cams(x_a)
    cb, Rb = cams(x_b)
    pa, pb = x_a[par.n_camera_params:].reshape(-1, 3), x_b[par.n_camera_params:].reshape(-1, 3)
    sc, R, t = _similarity(np.vstack([ca, pa]), np.vstack([cb, pb]))
    extent = np.abs(np.vstack([cb, pb])).max()
    pos = max(np.abs(sc * ca @ R.T + t - cb).max(), np.abs(sc * pa @ R.T + t - pb).max()) / extent
    ang = 0.0
    for A, B in zip(Ra, Rb):
        rel = (A @ R.T) @ B.T
        w = np.array([rel[2, 1] - rel[1, 2], rel[0, 2] - rel[2, 0], rel[1, 0] - rel[0, 1]])
        ang = max(ang, float(np.arctan2(0.5 * np.linalg.norm(w), 0.5 * (np.trace(rel) - 1.0))))
    return float(pos), float(ang)


def cpu_baseline(device_id=0):
    """The reference's scipy call on the oracle callables, one host core, on a bounded sample of the workload: cfg4's
    cameras and visibility recipe at half the points (64 cams / 100k points / 1M obs), full solve with the reference's
    default tolerances.  The same arrays and the same x0 then go through the product's solver on the GPU; `parity` is the
    difference of the two solutions (north star: RMS within 1e-4 px, poses / points within 1e-6 relative after gauge
    alignment — at default tolerances both solvers stop at ftol, so the alignment figure shows their stopping distance)."""
    from caliscope_amd.bundle_parameterization import BundleParameterization
    from caliscope_amd.least_squares import least_squares
    from caliscope_amd.synthetic import make_scene
    from oracle.residuals import joint_residuals
    from oracle.solver import optimize_scipy

    sc = make_scene("cfg4-sample", n_cams=64, n_points=100_000, n_obs=1_000_000)
    par = BundleParameterization.from_camera_array(sc.cameras_init, n_points=100_000, refine_intrinsics=False)
    x0 = par.pack(sc.cameras_init, sc.points_init)
    t0 = time.perf_counter()
    res = optimize_scipy(par, sc.camera_indices, sc.image_coords, sc.obj_indices, x0)
    dt = time.perf_counter() - t0
    iters = max(res.nfev - 1, 1)
    t1 = time.perf_counter()
    gpu = least_squares(None, x0, jac=None, bounds=par.bounds(), x_scale="jac", method="trf",
                        args=(par, sc.camera_indices, sc.image_coords, sc.obj_indices), devices=[device_id])
    dt_gpu = time.perf_counter() - t1
    fx = np.array([b.fx_initial for b in par.blocks])[sc.camera_indices]

    def rms(x):  # the oracle's residuals for both solutions: independent of the device code
        e = joint_residuals(x, par, sc.camera_indices, sc.image_coords, sc.obj_indices).reshape(-1, 2) * fx[:, None]
        return float(np.sqrt(np.mean(np.sum(e * e, axis=1))))

    rms_cpu, rms_gpu = rms(res.x), rms(gpu.x)
    pos, ang = solution_parity(par, gpu.x, res.x)
    base = {
        "value": round(sc.n_obs * iters / dt, 1), "unit": "obs/s", "cores": 1, "kind": "port",
        "sample": f"64 cams / 100k points / 1M obs (half of cfg4), scipy least_squares trf+lsmr, default tolerances: "
                  f"{res.nfev} evaluations in {dt:.1f} s; host has {os.cpu_count()} cores, the scipy path is single-threaded",
        "seconds": round(dt, 2), "nfev": int(res.nfev), "status": int(res.status), "cost": float(res.cost), "final_rms_px": round(rms_cpu, 6),
    }
    parity = {
        "sample": "the cpu_baseline sample: same arrays, same x0, default tolerances on both sides",
        "gpu": {"nfev": int(gpu.nfev), "status": int(gpu.status), "cost": float(gpu.cost), "final_rms_px": round(rms_gpu, 6),
                "seconds_end_to_end": round(dt_gpu, 3)},
        "d_rms_px": rms_gpu - rms_cpu, "rel_cost": (float(gpu.cost) - float(res.cost)) / float(res.cost),
        "aligned_pos": pos, "aligned_ang_rad": ang,
        "value_ratio_gpu_end_to_end_over_cpu": round(sc.n_obs * max(gpu.nfev - 1, 1) / dt_gpu / (sc.n_obs * iters / dt), 1),
    }
    return base, parity


def main():
    # The contract is ONE JSON line on stdout.  Libraries underneath (RCCL prints a version banner when a
    # communicator is created, HIP/amdgpu warnings) write to fd 1 directly, so everything except the final
    # line is sent to stderr at the file-descriptor level.
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="cfg4")
    ap.add_argument("--also", default="cfg2,cfg3,cfg5")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch multi-GPU runs with "
                         f"python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...")
    control = None
    if world > 1:
        from caliscope_amd.distributed import SocketControlPlane

        control = SocketControlPlane.from_env()  # host-side control plane only (TCP on the loopback); the data plane is RCCL inside the engine

    import __graft_entry__ as entry

    if rank == 0:
        entry.build()
    if control is not None:
        control.barrier()
    m = measure(args.workload, args.steps, args.warmup, device_id=local_rank, control=control, scaling=args.scaling)
    total_obs = m["n_obs_total"]
    value = total_obs * m["steps"] / m["elapsed"]
    out = {
        "metric": "observations/sec per LM iteration",
        "value": round(value, 1),
        "unit": "obs/s",
        "n_gpus": world,
        "steps": m["steps"],
        "warmup": args.warmup,
        "ms_per_step": round(m["elapsed"] / m["steps"] * 1e3, 4),
        "higher_is_better": True,
        "scaling": args.scaling if world > 1 else "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: {m['n_cams']} cams / {total_obs} obs in total ({m['n_points']} points / {m['n_obs']} obs on rank 0), "
                        f"{'extrinsics+intrinsics' if m['nct'] == 9 else 'extrinsics-only'} BA, {m['loss']} loss",
            "n_obs_total": total_obs, "params_per_camera": m["nct"],
            "parallelism": f"points sharded x{world} ({args.scaling if world > 1 else 'single GPU'}); RCCL all-reduce of camera blocks + reduced camera system per iteration",
            "tolerances": "ftol=xtol=gtol=1e-8 (reference defaults)", "solves_in_timed_region": m["solves"],
        },
        "final_rms_px": round(m["final_rms_px"], 6),
        "initial_rms_px": round(m["initial_rms_px"], 4),
        # accepted iterations re-linearise (njev - 1 of them after x0); the other trial points were rejected
        "solve": {"nfev": m["full_nfev"], "njev": m["full_njev"], "accepted_steps": m["full_njev"] - 1,
                  "rejected_trials": m["full_nfev"] - m["full_njev"], "status": m["full_status"], "cost": m["full_cost"]},
        "roofline": roofline_from(m),
        "engine": m["info"],
    }
    if not args.no_cpu and rank == 0 and world == 1:
        try:
            out["cpu_baseline"], out["parity"] = cpu_baseline(device_id=local_rank)
        except Exception as exc:  # the baseline must never break the bench line
            out["cpu_baseline"] = {"value": None, "unit": "obs/s", "cores": 1, "kind": "port", "sample": f"failed: {exc}"}
    also = {}
    if rank == 0 and world == 1 and args.also:
        for name in [s for s in args.also.split(",") if s and s != args.workload]:
            try:
                kw = {}
                a = measure(name, args.steps, args.warmup, device_id=local_rank, solve_kw=kw)
                rf = roofline_from(a)
                also[name] = {
                    "value": round(a["n_obs"] * a["steps"] / a["elapsed"], 1), "unit": "obs/s",
                    "ms_per_step": round(a["elapsed"] / a["steps"] * 1e3, 4), "final_rms_px": round(a["final_rms_px"], 6),
                    "workload": f"{a['n_cams']} cams / {a['n_points']} points / {a['n_obs']} obs, {a['loss']} loss",
                    "nfev": a["full_nfev"], "status": a["full_status"], "accepted_steps": a["full_njev"] - 1,
                    "rejected_trials": a["full_nfev"] - a["full_njev"], "initial_rms_px": round(a["initial_rms_px"], 4),
                    # cfg5 is the largest single-GPU configuration of BASELINE.json: its full roofline block, not a digest
                    "roofline": rf if (rf is None or name == "cfg5") else {k: rf[k] for k in ("kernel", "achieved", "frac", "avg_launch_us", "traffic")},
                }
            except Exception as exc:
                also[name] = {"error": str(exc)}
    if also:
        out["also"] = also
    if rank == 0:
        line = json.dumps(out, default=lambda o: int(o) if isinstance(o, np.integer) else float(o))
        sys.stdout.flush()
        os.write(real_stdout, (line + "\n").encode())
    if control is not None:
        control.barrier()
        control.close()


if __name__ == "__main__":
    main()
