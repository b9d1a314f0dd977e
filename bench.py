#!/usr/bin/env python
"""Headline benchmark: observations/sec per LM iteration (+ final RMS reprojection error, px).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg4] [--scaling strong|weak] [--no-cpu] [--also cfg2,cfg3,cfg5]
                    [--devices 0,1,..] [--xchg auto|rccl|direct]

``--gpus N`` works two ways.  Under a launcher that sets RANK / WORLD_SIZE (``python -m torch.distributed.run --nproc-per-node N ...``)
this process is one rank of N.  Started plainly, the N ranks run INSIDE this process — one host thread per device, the engines joined
by an RCCL communicator (threads share the unique id), exactly the route ``CaptureVolume.optimize()`` takes with
``CALISCOPE_HIP_DEVICES`` (caliscope_amd.distributed.solve_multi_device; the reference's solve is one in-process call,
core/capture_volume.py:387-411).  ``--xchg direct`` uses the library's peer-to-peer device group instead of RCCL (also the only
way to place several ranks on ONE device: ``--devices 0,0``).

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

``cpu_baseline`` is the reference's scipy call on the headline's OWN arrays (full cfg4: 4 evaluations, ~25 s on one host core);
``parity`` compares its solution with the headline's solve on the device (RMS px, cost, gauge-aligned poses / points) and times the
same arrays through the product seam end to end (``setup_ms``, ``value_end_to_end``).  ``also.cfg5.parity`` does the same on a
cfg5-recipe sample (128 cameras / 100k points / 1M observations, free intrinsics with the reference's bounds).
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
        # (SURVEY.md 8d's storage model.  Since round 6 six-parameter cameras stream the 96-byte T records k_tprep has just written instead of linearising
        # again: 100 bytes per observation actually move — `traffic` has the counters — the ALGORITHMIC figure stays the model's)
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


def run_iterations(engine, k_steps, solve_kw, count="trials"):
    """Execute k_steps iterations through cba_solve (the product's driver: the trust-region loop runs in the library, restarting from the
    x0 kept on the device); returns (n_solves, last_result).  count = "trials": exactly k_steps trial points, the metric's own unit (the
    headline); count = "accepted": until k_steps ACCEPTED steps have been made, however many rejected trials lie between them (the `also`
    workloads: a Huber solve from x0 begins with a run of rejected trials, and 20 trials held 2 accepted iterations in round 4)."""
    done, solves, last = 0, 0, None
    mix = {"accepted": 0, "rejected": 0, "rejected_timed": 0, "rejected_s": 0.0, "trials": 0}
    guard = 0
    while done < k_steps:
        budget = (k_steps - done) + 1 if count == "trials" else max(2 * (k_steps - done), 8) + 1
        last = engine.solve(None, max_nfev=budget, fetch_x=False, **solve_kw)
        done += (last.nfev - 1) if count == "trials" else (last.njev - 1)
        solves += 1
        mix["trials"] += last.nfev - 1
        mix["accepted"] += last.njev - 1
        mix["rejected"] += last.nfev - last.njev
        mix["rejected_timed"] += last.rejected_timed
        mix["rejected_s"] += last.rejected_seconds
        guard += 1
        if last.nfev <= 1 or guard > 50 * max(k_steps, 1):  # already converged at x0: nothing to iterate on
            break
    last.mix = mix
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


def measure(name, steps, warmup, device_id=0, timers=True, seed=42, solve_kw=None, control=None, scaling="strong", group=None, prebuilt=None,
            on_engine=None, count="trials", with_first_call=True, **overrides):
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
        # (in-process ranks share one generated scene: `prebuilt`; launcher ranks each generate the same one)
        sc, par, x0, prob, cfg = prebuilt if prebuilt is not None else build_problem(name, seed=seed, shard=0, **overrides)
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
    deterministic = os.environ.get("CBA_DETERMINISTIC", "0") not in ("", "0")  # (as the seam reads it: fixed-order sums, bit-identical runs)
    t_setup = time.perf_counter()
    eng = HipEngine(prob, device_id=device_id, deterministic=deterministic)  # sort, Schur plan, upload: paid once per problem structure, NOT part of `value`
    t_setup = time.perf_counter() - t_setup
    if on_engine is not None:
        on_engine(eng)  # (in-process ranks: a failing peer aborts this rank's communicator instead of leaving it in a collective)
    if control.world > 1:
        if group is not None:  # the library's peer-to-peer device group (one process, --xchg direct)
            eng.group_join(group, control.rank)
        else:                  # RCCL: rank 0's unique id travels over the host-side control plane
            uid = control.broadcast_bytes(eng.comm_unique_id() if control.rank == 0 else None, 128)
            eng.comm_init(uid, control.rank, control.world)
    two_stage = eng.info()["plan_state"] == 1  # a large handle: it starts on the quickly made Schur plan and swaps the balanced one in
    t_plan = time.perf_counter()
    eng.plan_wait()  # ... the timed region runs on the balanced one (cba_plan_wait raises if its background build failed)
    t_plan = time.perf_counter() - t_plan
    info = eng.info()
    if info["plan_state"] != 0:
        raise RuntimeError(f"the handle is not on its final Schur plan (plan_state {info['plan_state']}, plan_error {info['plan_error']})")
    eng.begin(x0)
    run_iterations(eng, max(warmup, 1), solve_kw, count)
    # timed region: exactly `steps` iterations, barrier + device sync on both sides, max over ranks
    control.barrier()
    t0 = time.perf_counter()
    solves, last = run_iterations(eng, steps, solve_kw, count)  # every engine call ends with a stream synchronize
    control.barrier()
    elapsed = control.allreduce_max(time.perf_counter() - t0)
    # instrumented repeat of the same `steps` iterations: HIP events around every kernel family on the engine's
    # stream.  Kept out of the region above because recording ~40 event pairs per iteration costs 6 % (cfg4) to
    # 50 % (cfg2) of the wall time, which would understate `value`.
    tm = {}
    if timers:
        eng.enable_timers(True)
        eng.reset_timers()
        run_iterations(eng, steps, solve_kw, count)
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
    # What a FIRST call of optimize() runs on (VERDICT r04 missing 4): a solve of a few iterations is over before the balanced plan exists, so
    # its iterations run on the quickly made one.  The same K steps on a handle forced to keep that plan (CBA_PLAN=cheap), single rank only.
    first_call = None
    if two_stage and control.world == 1 and with_first_call:
        old_env = os.environ.get("CBA_PLAN")
        os.environ["CBA_PLAN"] = "cheap"
        try:
            e1 = HipEngine(prob, device_id=device_id)
        finally:
            if old_env is None:
                os.environ.pop("CBA_PLAN", None)
            else:
                os.environ["CBA_PLAN"] = old_env
        try:
            assert e1.info()["plan_state"] == 2
            e1.begin(x0)
            run_iterations(e1, max(warmup, 1), solve_kw, count)
            t1 = time.perf_counter()
            _, last1 = run_iterations(e1, steps, solve_kw, count)
            dt1 = time.perf_counter() - t1
            first_call = {"plan": "cheap", "elapsed": dt1, "trials": last1.mix["trials"], "accepted": last1.mix["accepted"]}
        finally:
            e1.close()
    # the same handle built again in the now warm process (the set-up's large host arrays come from the library's block pool, the device buffers from
    # its arena pool: what a session that optimises, filters and optimises again pays): median of three, single rank only
    t_setup_warm = float("nan")
    if control.world == 1:
        again = []
        for _ in range(3):
            t_w = time.perf_counter()
            e2 = HipEngine(prob, device_id=device_id)
            again.append(time.perf_counter() - t_w)
            e2.close()
        t_setup_warm = float(np.median(again))
    return {
        "name": name, "n_obs": prob.n_obs, "n_obs_total": n_total if n_total is not None else prob.n_obs * control.world,
        "n_points": par.n_points, "n_cams": len(par.blocks), "ncp": par.n_camera_params, "full_njev": full.njev,
        "nct": 9 if any(b.n_params == 9 for b in par.blocks) else 6, "loss": prob.loss, "elapsed": elapsed, "steps": steps,
        "solves": solves, "timers": tm, "final_rms_px": rms, "initial_rms_px": rms0, "full_nfev": full.nfev,
        "full_status": full.status, "full_cost": full.cost, "info": info, "t_generate_s": t_gen,
        "scene": sc, "par": par, "x0": x0, "x_full": full.x, "setup_s": t_setup, "setup_warm_s": t_setup_warm, "plan_wait_s": t_plan, "rank": control.rank, "mix": last.mix,
        "two_stage": two_stage, "first_call": first_call, "count": count, "deterministic": deterministic,
    }


# kernels of each timer family in the rocprofv3 summaries (the Schur pass is k_tprep + k_schur_reg3)
PMC_KERNEL = {"schur": ("k_tprep", "k_schur_reg3"), "build": ("k_build",), "jv": ("k_jv",),
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


# The roof that actually binds (SURVEY.md 8d caveat): FP64 VALU issue.  A wave-level VALU instruction holds its SIMD's vector
# pipe for four clocks (64 lanes, 16 per clock); MI355X has 256 CUs x 4 SIMDs at 2.4 GHz.  profiles/sq_<workload>.json holds SQ_INSTS_VALU per launch of
# the dominant kernels (rocprofv3 --pmc, tools/gpu_profile_run.sh -> tools/sq_valu_floor.py); launches per ACCEPTED iteration below.
N_SIMD, SHADER_GHZ = 1024, 2.4


def valu_floor(workload, ncp, rows=None):
    """{kernel: issue-bound microseconds per accepted iteration} and their sum from the committed SQ counters; None if absent."""
    if rows is None:
        path = ROOT / "profiles" / f"sq_{workload}.json"
        if not path.exists():
            return None
        rows = json.loads(path.read_text())
    per_iter = {"k_tprep": 1, "k_schur_reg3": 1, "k_build_cs": 1, "k_jv": 1, "k_backsub": 1, "k_chol_step": (ncp + 31) // 32 + 1}
    out = {}
    for name, row in rows.items():
        for prefix, n in per_iter.items():
            if name.startswith(prefix) and "SQ_INSTS_VALU" in row:  # (variants of one kernel — the build at x and at the trial point — count once: the larger)
                out[prefix] = max(out.get(prefix, 0.0), row["SQ_INSTS_VALU"] * 4.0 / N_SIMD / (SHADER_GHZ * 1e3) * n)
    if not out:
        return None
    return {"per_kernel_us": {k: round(v, 1) for k, v in out.items()}, "iteration_us": round(sum(out.values()), 1),
            "how": "SQ_INSTS_VALU per launch x 4 clocks / 1024 SIMDs / 2.4 GHz x launches per accepted iteration (profiles/sq_*.json)"}


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
        "iteration": (lambda b, t, vf: {"alg_bytes": b, "GBps": round(b / t / 1e9, 1), "frac": round(b / t / 1e9 / HBM_PEAK_GBS, 4),
                                        # the two floors of one accepted iteration next to its measured time: HBM (algorithmic bytes at 8 TB/s) and FP64
                                        # VALU issue (the roof that binds: SURVEY.md 8d caveat); frac_of_valu_floor = floor / measured
                                        "measured_us": round(t * 1e6, 1), "hbm_floor_us": round(b / (HBM_PEAK_GBS * 1e9) * 1e6, 1),
                                        "valu_floor_us": vf["iteration_us"] if vf else None, "valu_floor": vf,
                                        "frac_of_valu_floor": round(vf["iteration_us"] / (t * 1e6), 3) if vf else None,
                                        "binding_roof": "fp64 valu issue" if vf and vf["iteration_us"] > b / (HBM_PEAK_GBS * 1e9) * 1e6 else "hbm"})(
            72 * m["n_obs"] + 312 * m["n_points"] + 8 * m["ncp"] ** 2, m["elapsed"] / max(m["steps"], 1), valu_floor(m["name"], m["ncp"])),
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


def solution_parity(par, x_a, x_b, detail=False):
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
    d_cam = np.abs(sc * ca @ R.T + t - cb).max(axis=1) / extent
    d_pts = np.abs(sc * pa @ R.T + t - pb).max(axis=1) / extent
    pos = max(d_cam.max(), d_pts.max())
    ang = 0.0
    for A, B in zip(Ra, Rb):
        rel = (A @ R.T) @ B.T
        w = np.array([rel[2, 1] - rel[1, 2], rel[0, 2] - rel[2, 0], rel[1, 0] - rel[0, 1]])
        ang = max(ang, float(np.arctan2(0.5 * np.linalg.norm(w), 0.5 * (np.trace(rel) - 1.0))))
    if detail:  # where the maximum sits: camera centres alone, and the distribution over the points
        return {"aligned_pos": float(pos), "aligned_ang_rad": float(ang), "cameras_pos": float(d_cam.max()),
                "points_pos_median": float(np.median(d_pts)), "points_pos_p999": float(np.quantile(d_pts, 0.999)), "points_pos_max": float(d_pts.max()),
                "points_above_1e-6": int(np.sum(d_pts > 1e-6)), "n_points": int(len(d_pts))}
    return float(pos), float(ang)


def weak_points(sc, par, x_product, x_ref, loss="linear", f_scale=1.0, limit=5):
    """The points two converged answers place more than 1e-6 apart (gauge-aligned), and what the DATA say about each: the oracle's cost at the product's
    answer with that one point moved to where the reference has it (brought into the product's gauge), relative to the cost.  A change below ~1e-13 means
    double precision cannot tell the two positions apart — e.g. a point whose observations all sit in Huber's linear region is nearly free along its
    rays — and neither solver's termination test can prefer one."""
    pa, pb = x_ref[par.n_camera_params:].reshape(-1, 3), x_product[par.n_camera_params:].reshape(-1, 3)

    def centres(x):
        from caliscope_amd.cameras import rvec_to_matrix
        return np.array([-rvec_to_matrix(x[o:o + 3]).T @ x[o + 3:o + 6] for o in par.camera_param_offsets])

    s, R, t = _similarity(np.vstack([centres(x_ref), pa]), np.vstack([centres(x_product), pb]))  # reference -> the product's gauge
    moved = s * pa @ R.T + t
    extent = np.abs(np.vstack([centres(x_product), pb])).max()
    dist = np.abs(moved - pb).max(axis=1) / extent
    idx = np.argsort(-dist)[:limit]
    idx = idx[dist[idx] > 1e-6]
    cost0 = oracle_cost(sc, par, x_product, loss, f_scale)
    out = []
    for q in idx:
        x_mod = np.array(x_product, dtype=float, copy=True)
        x_mod[par.n_camera_params + 3 * q: par.n_camera_params + 3 * q + 3] = moved[q]
        n_rows = int(np.sum(sc.obj_indices == q))
        out.append({"point": int(q), "aligned_distance": float(dist[q]), "observations": n_rows,
                    "rel_cost_change_if_moved_to_the_reference_position": (oracle_cost(sc, par, x_mod, loss, f_scale) - cost0) / cost0})
    return {"points_above_1e-6": int(np.sum(dist > 1e-6)), "listed": out,
            "all_listed_indistinguishable_at_1e-12_of_the_cost": bool(all(abs(o["rel_cost_change_if_moved_to_the_reference_position"]) <= 1e-12 for o in out))}


def oracle_cost(sc, par, x, loss="linear", f_scale=1.0):
    """0.5 * sum rho(f) at x from the ORACLE's residuals and scipy's own loss functions (nothing of the product involved)."""
    from oracle.residuals import joint_residuals

    f = joint_residuals(x, par, sc.camera_indices, sc.image_coords, sc.obj_indices)
    if loss == "linear":
        return 0.5 * float(f @ f)
    from scipy.optimize._lsq.least_squares import construct_loss_function

    return float(construct_loss_function(f.size, loss, f_scale)(f, cost_only=True))


def oracle_polish(sc, par, x_product, loss="linear", f_scale=1.0, max_nfev=60, tight_inner=True):
    """The independent direction of the converged-parity check: the REFERENCE's solver (oracle/solver.py = scipy on the
    oracle callables) started AT the product's converged point with ftol = xtol = gtol = 1e-15.  If that point is the minimum, scipy has
    nothing to gain: it must stop within a few evaluations, having moved (gauge-aligned) by far less than 1e-6 and lowered the cost by less
    than 1e-12 relative.  Run twice: with scipy's default inner LSMR tolerance (1e-6, the reference's call) and, `tight_inner`, with
    tr_options atol = btol = 1e-14 so that its steps are exact to rounding and it really tries to improve on the point."""
    from oracle.solver import optimize_scipy

    cost0 = oracle_cost(sc, par, x_product, loss, f_scale)
    out = {"what": "scipy (oracle callables) started at the product's converged x, ftol = xtol = gtol = 1e-15", "cost_at_product_x": cost0}
    ok = True
    for name, tr in (("lsmr_default", None), ("lsmr_tight", dict(atol=1e-14, btol=1e-14)))[: 2 if tight_inner else 1]:
        t0 = time.perf_counter()
        sp = optimize_scipy(par, sc.camera_indices, sc.image_coords, sc.obj_indices, x_product, loss=loss, f_scale=f_scale,
                            ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=max_nfev, tr_options=tr)
        moved = solution_parity(par, sp.x, x_product, detail=True)
        gain = (cost0 - float(sp.cost)) / cost0
        out[name] = {"nfev": int(sp.nfev), "njev": int(sp.njev), "status": int(sp.status), "seconds": round(time.perf_counter() - t0, 2),
                     "moved": moved, "rel_cost_gain": gain}
        ok = ok and moved["aligned_pos"] <= 1e-6 and moved["aligned_ang_rad"] <= 1e-6 and gain <= 1e-12
    out["scipy_moves_within_1e-6_and_gains_within_1e-12"] = bool(ok)
    return out


def stored_scipy_reference(case, x0):
    """tests/golden/scipy_refs/<case>.npz (tests/golden/make_scipy_refs.py: the reference's scipy call run on the CPU, minutes at these sizes)
    when it was made from this very x0, else None."""
    try:
        sys.path.insert(0, str(ROOT / "tests" / "golden"))
        import make_scipy_refs

        return make_scipy_refs.load(case, x0)
    except Exception:  # noqa: BLE001 - an optional fixture
        return None
    finally:
        sys.path.pop(0)


def _scipy_vs_product(sc, par, x0, x_gpu_solve, device_id, label, loss="linear", f_scale=1.0, stored=None):
    """The reference's scipy call (oracle callables, one host core) and the product on the SAME arrays and x0:
    returns (cpu_baseline block, parity block).  `x_gpu_solve`: a solution the device already produced from these arrays (the
    headline's full solve), or None — the solution of the seam call below is compared then.  Either way the arrays also go
    through the product seam (`caliscope_amd.least_squares.least_squares`, what CaptureVolume.optimize calls) once, timed end to
    end: handle set-up (sort, Schur plan, upload) + solve + download."""
    from caliscope_amd import engine_cache
    from caliscope_amd.least_squares import least_squares
    from oracle.residuals import joint_residuals
    from oracle.solver import optimize_scipy

    # `stored` = (default-tolerance case, tight case) of tests/golden/scipy_refs: scipy solutions of these very arrays computed beforehand
    ref_default = stored_scipy_reference(stored[0], x0) if stored else None
    ref_tight = stored_scipy_reference(stored[1], x0) if stored else None
    if ref_default is not None:
        from types import SimpleNamespace

        res = SimpleNamespace(x=ref_default["x"], nfev=ref_default["nfev"], status=ref_default["status"], cost=ref_default["cost"])
        dt = float(ref_default["seconds"])
    else:
        t0 = time.perf_counter()
        res = optimize_scipy(par, sc.camera_indices, sc.image_coords, sc.obj_indices, x0, loss=loss, f_scale=f_scale)
        dt = time.perf_counter() - t0
    iters = max(res.nfev - 1, 1)
    engine_cache.clear(trim=False)  # a cold call: the handle is built inside the timed call (the library keeps its pools: a warm process)
    t1 = time.perf_counter()
    gpu = least_squares(None, x0, jac=None, bounds=par.bounds(), x_scale="jac", method="trf", loss=loss, f_scale=f_scale,
                        args=(par, sc.camera_indices, sc.image_coords, sc.obj_indices), devices=[device_id])
    dt_gpu = time.perf_counter() - t1
    t2 = time.perf_counter()  # second call on the same observations: the kept handle is found by its fingerprint (engine_cache)
    warm = least_squares(None, x0, jac=None, bounds=par.bounds(), x_scale="jac", method="trf", loss=loss, f_scale=f_scale,
                         args=(par, sc.camera_indices, sc.image_coords, sc.obj_indices), devices=[device_id])
    dt_warm = time.perf_counter() - t2
    engine_cache.clear(trim=False)
    fx = np.array([b.fx_initial for b in par.blocks])[sc.camera_indices]

    def rms(x):  # the oracle's residuals for both solutions: independent of the device code
        e = joint_residuals(x, par, sc.camera_indices, sc.image_coords, sc.obj_indices).reshape(-1, 2) * fx[:, None]
        return float(np.sqrt(np.mean(np.sum(e * e, axis=1))))

    x_cmp = gpu.x if x_gpu_solve is None else x_gpu_solve
    rms_cpu, rms_gpu = rms(res.x), rms(x_cmp)
    pos, ang = solution_parity(par, x_cmp, res.x)
    # Whose distance is it?  Both solvers stop on ftol; scipy's LSMR steps are inexact, so it can stop well short of the minimum (the bounded
    # cfg5 recipe: 120 evaluations, cost 4e-6 above).  The product started again from scipy's stopping point with tight tolerances ends at the
    # minimum nearest to scipy's answer; the product started from x0 with the same tolerances must end at the SAME minimum.
    tight = dict(ftol=1e-13, xtol=1e-13, gtol=1e-13, max_nfev=2000)
    kw = dict(jac=None, bounds=par.bounds(), x_scale="jac", method="trf", loss=loss, f_scale=f_scale,
              args=(par, sc.camera_indices, sc.image_coords, sc.obj_indices), devices=[device_id])
    near = least_squares(None, res.x, **kw, **tight)
    mine = least_squares(None, x0, **kw, **tight)
    engine_cache.clear(trim=False)
    polish = {
        "what": "product at 1e-13 from scipy's stopping point (= the minimum nearest to scipy's answer) against the product at 1e-13 from x0",
        "same_minimum": solution_parity(par, mine.x, near.x, detail=True),
        "d_rms_px": rms(mine.x) - rms(near.x), "rel_cost": (float(mine.cost) - float(near.cost)) / float(near.cost),
        "scipy_to_its_minimum": solution_parity(par, res.x, near.x, detail=True),
        "rel_cost_scipy_above_minimum": (float(res.cost) - float(near.cost)) / float(near.cost),
        "nfev": {"from_x0": int(mine.nfev), "from_scipy_x": int(near.nfev)},
    }
    cost_gpu = oracle_cost(sc, par, x_cmp, loss, f_scale)
    # the independent direction: scipy restarted at the product's tight solution (tight inner solves where a matrix-vector product is cheap)
    opolish = oracle_polish(sc, par, mine.x, loss, f_scale, tight_inner=sc.n_obs <= 1_200_000)
    tight_ref = None
    if ref_tight is not None:  # SURVEY.md 7 protocol (iii): scipy at ftol = xtol = gtol = 1e-15 from the same x0, against the product at 1e-13
        d = solution_parity(par, mine.x, ref_tight["x"], detail=True)
        c_mine = oracle_cost(sc, par, mine.x, loss, f_scale)
        tight_ref = {"what": "product at 1e-13 from x0 against scipy at 1e-15 (inner LSMR 1e-14) from x0, tests/golden/scipy_refs/" + stored[1],
                     "scipy": {k: ref_tight[k] for k in ("nfev", "njev", "status", "cost", "seconds", "settings")},
                     "detail": d, "d_rms_px": rms(mine.x) - rms(ref_tight["x"]), "rel_cost": (c_mine - float(ref_tight["cost"])) / float(ref_tight["cost"]),
                     "within_north_star": bool(d["aligned_pos"] <= 1e-6 and d["aligned_ang_rad"] <= 1e-6
                                               and abs(rms(mine.x) - rms(ref_tight["x"])) <= 1e-4)}
    base = {
        "value": round(sc.n_obs * iters / dt, 1), "unit": "obs/s", "cores": 1, "kind": "port",
        "sample": f"{label}, scipy least_squares trf+lsmr, default tolerances: "
                  f"{res.nfev} evaluations in {dt:.1f} s; host has {os.cpu_count()} cores, the scipy path is single-threaded"
                  + (f" (solution and time stored beforehand: tests/golden/scipy_refs/{stored[0]}.npz)" if ref_default is not None else ""),
        "seconds": round(dt, 2), "nfev": int(res.nfev), "status": int(res.status), "cost": float(res.cost), "final_rms_px": round(rms_cpu, 6),
        # stored: solution AND seconds come from tests/golden/scipy_refs (one core of the build container), not from this box in this run
        "stored": ref_default is not None, "timed_on": "the build container, beforehand (tests/golden/make_scipy_refs.py)" if ref_default is not None else "this host, in this run",
    }
    gpu_iters = max(gpu.nfev - 1, 1)
    parity = {
        "sample": f"{label}: same arrays, same x0, default tolerances on both sides"
                  + ("; the device solution compared is the headline's own full solve" if x_gpu_solve is not None else ""),
        "gpu": {"nfev": int(gpu.nfev), "status": int(gpu.status), "cost": cost_gpu, "final_rms_px": round(rms_gpu, 6),
                "seconds_end_to_end": round(dt_gpu, 4), "seconds_end_to_end_cached_handle": round(dt_warm, 4),
                "setup_ms": round(float(getattr(gpu, "setup_seconds", float("nan"))) * 1e3, 2),
                "solve_ms": round(float(getattr(gpu, "solve_seconds", float("nan"))) * 1e3, 2),
                "value_end_to_end": round(sc.n_obs * gpu_iters / dt_gpu, 1), "host_cores": os.cpu_count()},
        "d_rms_px": rms_gpu - rms_cpu, "rel_cost": (cost_gpu - float(res.cost)) / float(res.cost),
        "aligned_pos": pos, "aligned_ang_rad": ang, "detail": solution_parity(par, x_cmp, res.x, detail=True),
        "within_north_star": bool(abs(rms_gpu - rms_cpu) <= 1e-4 and pos <= 1e-6 and ang <= 1e-6),
        # what a maintainer diffing the two solutions at the reference's DEFAULT tolerances sees, and whose distance it is
        "default_tolerance_distance": {
            "aligned_pos": pos, "aligned_ang_rad": ang,
            "scipy_rel_cost_above_the_minimum": polish["rel_cost_scipy_above_minimum"],
            "product_rel_cost_above_the_minimum": (cost_gpu - float(near.cost)) / float(near.cost),
            "scipy_distance_to_the_minimum": polish["scipy_to_its_minimum"]["aligned_pos"],
            "reading": "both stop on ftol = 1e-8; scipy's LSMR steps are solved to 1e-6 only, so its default call can stop short of the minimum by "
                       "`scipy_rel_cost_above_the_minimum` of the cost — that, not the product, is the distance above when it exceeds 1e-6 "
                       "(`tight_reference` / `oracle_polish` compare both solvers AT the minimum)"},
        "polish": polish,
        "oracle_polish": opolish,
        "tight_reference": tight_ref,
        "same_minimum_within_north_star": bool(abs(polish["d_rms_px"]) <= 1e-4 and polish["same_minimum"]["aligned_pos"] <= 1e-6
                                               and polish["same_minimum"]["aligned_ang_rad"] <= 1e-6),
        "value_ratio_gpu_end_to_end_over_cpu": round(sc.n_obs * gpu_iters / dt_gpu / (sc.n_obs * iters / dt), 1),
        "value_ratio_uses_stored_cpu_seconds": ref_default is not None,
    }
    return base, parity


def cpu_baseline(m, device_id=0):
    """scipy on the headline's own arrays (one host core; `m` = the headline's measure() record), compared with the headline's own
    solve.  north star: RMS within 1e-4 px, poses / points within 1e-6 relative after gauge alignment."""
    sc = m["scene"]
    label = f"the headline's arrays: {m['name']}, {m['n_cams']} cams / {m['n_points']} points / {m['n_obs']} obs (full size)"
    fs = sc.f_scale_1px() if sc.loss != "linear" else 1.0
    return _scipy_vs_product(sc, m["par"], m["x0"], m["x_full"], device_id, label, loss=sc.loss, f_scale=fs)


def cfg5_sample_parity(device_id=0, n_points=10_000):
    """cfg5's recipe at a size scipy finishes: 128 cameras, `n_points` points seen by 10 cameras each, free intrinsics with the perturbed start
    (f x 1.03, k1 + 0.02, k2 + 0.05) and the bounds of core/bundle_parameterization.py:151-164 — scipy runs trf_bounds (96 evaluations on this
    recipe: LSMR steps crawl along the intrinsics), the product its Coleman-Li variant (cba_solve).  Default 10k points / 100k observations
    (about a minute of one host core, so that the default bench run stays within minutes); SURVEY.md 8d's 1M-observation sample
    (--cfg5-sample-points 100000: ~7 minutes of scipy) is run once per round by tools/parity_at_size.py -> profiles/parity_r03.json."""
    from caliscope_amd.bundle_parameterization import BundleParameterization
    from caliscope_amd.synthetic import make_scene

    sc = make_scene("cfg5-sample", n_cams=128, n_points=n_points, n_obs=10 * n_points, refine=True)
    par = BundleParameterization.from_camera_array(sc.cameras_init, n_points=n_points, refine_intrinsics=True)
    x0 = par.pack(sc.cameras_init, sc.points_init)
    stored = {10_000: ("cfg5s100k_default", "cfg5s100k_tight"), 100_000: ("cfg5s1M_default", "cfg5s1M_tight")}.get(n_points)
    base, parity = _scipy_vs_product(sc, par, x0, None, device_id,
                                     f"cfg5 recipe sample: 128 cams / {n_points} points / {10 * n_points} obs, refine_intrinsics=True, bounds",
                                     stored=stored)
    parity["scipy"] = {k: base[k] for k in ("seconds", "nfev", "status", "cost", "final_rms_px")}
    return parity


def cfg3_tight_parity(device_id=0):
    """cfg3 (32 cams / 50k points / 400k obs, 5 % outliers, Huber at 1 px) run to the minimum on both sides, in the driver's own bench line (VERDICT r04
    item 2 iii): the product at ftol = xtol = gtol = 1e-15 through the seam against the stored scipy solve of the same x0 at 1e-15 with tight inner
    LSMR (tests/golden/scipy_refs/cfg3_tight.npz, 65 evaluations, made by tests/golden/make_scipy_refs.py on the oracle callables), and
    ``oracle_polish``: scipy (default inner tolerance) started AT the product's answer.  One of the 50 000 points has all its observations in
    Huber's linear region and is nearly free along its rays: ``weak_points`` reports what moving it to scipy's position does to the ORACLE's cost."""
    from caliscope_amd import engine_cache
    from caliscope_amd.least_squares import least_squares
    from oracle.residuals import joint_residuals

    sc, par, x0, prob, _ = build_problem("cfg3")
    ref = stored_scipy_reference("cfg3_tight", x0)
    if ref is None:
        return {"skipped": "tests/golden/scipy_refs/cfg3_tight.npz is absent or was made from another x0"}
    fs = prob.f_scale
    t0 = time.perf_counter()
    got = least_squares(None, x0, jac=None, bounds=par.bounds(), x_scale="jac", loss=prob.loss, f_scale=fs, method="trf",
                        args=(par, sc.camera_indices, sc.image_coords, sc.obj_indices), devices=[device_id],
                        ftol=1e-15, xtol=1e-15, gtol=1e-15, max_nfev=20000)
    dt = time.perf_counter() - t0
    engine_cache.clear(trim=False)
    fx = np.array([b.fx_initial for b in par.blocks])[sc.camera_indices]

    def rms(x):
        e = joint_residuals(x, par, sc.camera_indices, sc.image_coords, sc.obj_indices).reshape(-1, 2) * fx[:, None]
        return float(np.sqrt(np.mean(np.sum(e * e, axis=1))))

    d = solution_parity(par, got.x, ref["x"], detail=True)
    c_mine = oracle_cost(sc, par, got.x, prob.loss, fs)
    d_rms = rms(got.x) - rms(ref["x"])
    out = {
        "what": "product at 1e-15 from x0 against scipy at 1e-15 (inner LSMR 1e-14) from the same x0, tests/golden/scipy_refs/cfg3_tight.npz",
        "scipy": {k: ref[k] for k in ("nfev", "njev", "status", "cost", "seconds", "settings")}, "scipy_stored": True,
        "gpu": {"nfev": int(got.nfev), "njev": int(got.njev), "status": int(got.status), "cost_by_oracle": c_mine, "seconds_end_to_end": round(dt, 3)},
        "detail": d, "d_rms_px": d_rms, "rel_cost": (c_mine - float(ref["cost"])) / float(ref["cost"]),
        "cameras_within_1e-6": bool(d["cameras_pos"] <= 1e-6 and d["aligned_ang_rad"] <= 1e-6),
        "within_north_star": bool(d["aligned_pos"] <= 1e-6 and d["aligned_ang_rad"] <= 1e-6 and abs(d_rms) <= 1e-4),
    }
    if 0 < d["points_above_1e-6"] <= 50:
        out["weak_points"] = weak_points(sc, par, got.x, ref["x"], prob.loss, fs)
    out["oracle_polish"] = oracle_polish(sc, par, got.x, prob.loss, fs, tight_inner=False)
    return out


def step_mix(m):
    """The timed region's steps by kind.  A step of the metric is one trial point; an ACCEPTED step costs a whole iteration (linearisation,
    Schur pass, dense solve, back-substitution, trial build), a rejected trial evaluated by a call of its own (cba_trial) one cost pass.  The
    driver times those calls (cba_result.t_rejected_s); a rejected FIRST trial of a fused iteration was built inside cba_step and costs what
    an accepted one costs, so it counts with the accepted ones here."""
    mix = m["mix"]
    full_iters = mix["accepted"] + mix["rejected"] - mix["rejected_timed"]
    t_full = max(m["elapsed"] - mix["rejected_s"], 0.0)
    out = {
        # the Schur plan the timed steps ran on: "dealt" = the balanced plan (behind cba_plan_wait for handles that start on the quick one)
        "plan": "dealt", "counted": m.get("count", "trials"), "trial_points": mix.get("trials", mix["accepted"] + mix["rejected"]),
        "accepted_steps": mix["accepted"], "rejected_trials": mix["rejected"], "rejected_trials_timed": mix["rejected_timed"],
        "ms_per_accepted_step": round(t_full / full_iters * 1e3, 4) if full_iters else None,
        "ms_per_rejected_trial": round(mix["rejected_s"] / mix["rejected_timed"] * 1e3, 4) if mix["rejected_timed"] else None,
    }
    fc = m.get("first_call")
    if fc:  # the same steps on the quickly made plan: what the iterations of a first optimize() call cost before the balanced plan is swapped in
        out["first_call"] = {"plan": fc["plan"], "ms_per_step": round(fc["elapsed"] / max(fc["trials"], 1) * 1e3, 4),
                             "value": round(m["n_obs"] * fc["trials"] / fc["elapsed"], 1), "trial_points": fc["trials"], "accepted_steps": fc["accepted"]}
    elif not m.get("two_stage", False):
        out["first_call"] = "this handle is built on the balanced plan at once: a first call runs the same iterations"
    return out


def _also_block(a, name):
    rf = roofline_from(a)
    sm = step_mix(a)
    per_acc = sm["ms_per_accepted_step"]
    return {
        # accepted iterations only: with K small the first Huber evaluations from x0 are mostly rejected trials (a 25 us cost pass each), and a
        # "step" averaged over that mix says nothing about the iteration (cfg3: 0.077 ms at K = 20 against 0.177 ms at K = 40 in round 3)
        "value": round(a["n_obs"] / (per_acc * 1e-3), 1) if per_acc else None, "unit": "obs/s",
        "ms_per_step": per_acc, "timed_region": {**sm, "steps": a["steps"], "ms_per_step_mixed": round(a["elapsed"] / max(sm["trial_points"], 1) * 1e3, 4)},
        "final_rms_px": round(a["final_rms_px"], 6),
        "workload": f"{a['n_cams']} cams / {a['n_points']} points / {a['n_obs']} obs, {a['loss']} loss",
        "nfev": a["full_nfev"], "status": a["full_status"], "accepted_steps": a["full_njev"] - 1,
        "rejected_trials": a["full_nfev"] - a["full_njev"], "initial_rms_px": round(a["initial_rms_px"], 4),
        "setup_ms": round(a["setup_s"] * 1e3, 1), "setup_ms_warm": None if a["setup_warm_s"] != a["setup_warm_s"] else round(a["setup_warm_s"] * 1e3, 1),
        "plan_wait_ms": round(a["plan_wait_s"] * 1e3, 1),
        # cfg5 is the largest single-GPU configuration of BASELINE.json: its full roofline block, not a digest
        "roofline": rf if (rf is None or name == "cfg5") else {k: rf[k] for k in ("kernel", "achieved", "frac", "avg_launch_us", "traffic")},
    }


def run_ranks_in_process(args, devices, xchg):
    """``--gpus N`` without a launcher: the N ranks as host threads of this process, one engine per entry of `devices`, joined by
    RCCL (threads share the unique id through the ThreadControlPlane) or by the library's device group (`xchg` = "direct").
    Returns the per-rank measure() records, rank order."""
    import threading

    from caliscope_amd.distributed import ThreadControlPlane, _ThreadGroupState
    from caliscope_amd.hip_engine import DeviceGroup

    world = len(devices)
    prebuilt = build_problem(args.workload) if args.scaling == "strong" else None  # one scene, shared by the ranks (read-only)
    group = DeviceGroup(world) if xchg == "direct" else None
    state = _ThreadGroupState(world)
    records, errors = [None] * world, [None] * world
    engines, engines_lock = [None] * world, threading.Lock()

    def member(rank):
        ctl = ThreadControlPlane(state, rank)

        def keep(engine):
            with engines_lock:
                engines[rank] = engine

        try:
            records[rank] = measure(args.workload, args.steps, args.warmup, device_id=devices[rank], control=ctl, scaling=args.scaling,
                                    group=group, prebuilt=prebuilt, on_engine=keep)
        except BaseException as exc:  # noqa: BLE001 - reported by the main thread
            errors[rank] = exc
            ctl.abort()
            if group is not None:
                group.abort()
            else:  # RCCL: peers sitting in an all-reduce this rank will never join (as caliscope_amd.distributed.solve_multi_device does)
                with engines_lock:
                    peers = [e for r, e in enumerate(engines) if r != rank and e is not None]
                for e in peers:
                    try:
                        e.comm_abort()  # (excluded against the peer's own close() by the engine's lock)
                    except Exception:  # noqa: BLE001 - best effort on the failure path
                        pass
        finally:
            with engines_lock:
                engines[rank] = None

    threads = [threading.Thread(target=member, args=(r,), name=f"bench-rank{r}", daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if group is not None:
        group.close()
    failed = [(r, e) for r, e in enumerate(errors) if e is not None]
    if failed:
        real = [(r, e) for r, e in failed if "BrokenBarrier" not in type(e).__name__] or failed
        raise RuntimeError(f"rank {real[0][0]} of {world} failed: {real[0][1]!r}") from real[0][1]
    return records


def _source_digest():
    from caliscope_amd.build import source_digest

    return source_digest()


def main(argv=None):
    # The contract is ONE JSON line on stdout.  Libraries underneath (RCCL prints a version banner when a
    # communicator is created, HIP/amdgpu warnings) write to fd 1 directly, so everything except the final
    # line is sent to stderr at the file-descriptor level.
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        line = _run(argv)
    finally:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
    if line is not None:
        os.write(1, (line + "\n").encode())
    os.close(real_stdout)
    return line


def _run(argv):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="cfg4")
    ap.add_argument("--also", default="cfg2,cfg3,cfg5")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-first-call", action="store_true",
                    help="skip the repeat of the timed steps on the quickly made Schur plan (profiling runs: keeps one population of launches per kernel)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--devices", default="", help="in-process ranks: device ordinal of every rank (default 0..N-1)")
    ap.add_argument("--cfg5-sample-points", type=int, default=10_000, help="points of the cfg5-recipe parity sample (10 observations each)")
    ap.add_argument("--xchg", choices=("auto", "rccl", "direct"), default="auto",
                    help="in-process ranks: RCCL communicator (default for distinct devices) or the library's peer-to-peer device group")
    args = ap.parse_args(argv)

    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else args.gpus
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
    if launched and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    import __graft_entry__ as entry

    control = None
    if launched and world > 1:
        from caliscope_amd.distributed import SocketControlPlane

        control = SocketControlPlane.from_env()  # host-side control plane only (TCP on the loopback); the data plane is RCCL inside the engine
    if rank == 0:
        entry.build()
    if control is not None:
        control.barrier()

    per_rank = None
    route = "single GPU"
    if not launched and world > 1:
        # the N ranks inside this process: what `python bench.py --gpus N` means on a node without a launcher
        from caliscope_amd.hip_engine import device_count

        devices = [int(t) for t in args.devices.split(",") if t.strip() != ""] or list(range(world))
        if len(devices) != world:
            raise SystemExit(f"--devices names {len(devices)} devices for --gpus {world}")
        xchg = args.xchg if args.xchg != "auto" else os.environ.get("CBA_XCHG", "rccl" if len(set(devices)) == world else "direct")
        n_dev = device_count()
        if max(devices) >= n_dev or min(devices) < 0:
            raise SystemExit(f"--gpus {world}: device ordinals {devices} requested but this node shows {n_dev} HIP device(s); "
                             f"run with --gpus <= {max(n_dev, 1)}, or name ranks per device with --devices (e.g. --devices 0,0 --xchg direct "
                             f"places two ranks of the protocol on one GPU)")
        if xchg == "rccl" and len(set(devices)) != world:
            raise SystemExit("RCCL needs one distinct device per rank; use --xchg direct to place several ranks on one device")
        per_rank = run_ranks_in_process(args, devices, xchg)
        m = per_rank[0]
        route = f"{world} ranks in one process (one host thread per device {devices}), exchange: {'RCCL communicator' if xchg == 'rccl' else 'peer-to-peer device group'}"
        exchange_backend = xchg
    else:
        m = measure(args.workload, args.steps, args.warmup, device_id=local_rank, control=control, scaling=args.scaling,
                    with_first_call=not args.no_first_call)
        exchange_backend = "rccl" if world > 1 else None
        if world > 1:
            route = f"{world} processes (launcher), one per GPU, exchange: RCCL communicator"
    total_obs = m["n_obs_total"]
    value = total_obs * m["steps"] / m["elapsed"]
    obs_per_rank = [r["n_obs"] for r in per_rank] if per_rank else [m["n_obs"]]
    comm = m["timers"].get("exchange") if m["timers"] else None
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
            "parallelism": f"points sharded x{world} ({args.scaling if world > 1 else 'single GPU'}); all-reduce of camera blocks + reduced camera system per iteration; {route}"
                           + (f"; obs per rank {obs_per_rank}" if world > 1 else ""),
            "tolerances": "ftol=xtol=gtol=1e-8 (reference defaults)", "solves_in_timed_region": m["solves"],
            "sums": "fixed order (CBA_DETERMINISTIC=1: bit-identical from run to run)" if m.get("deterministic") else "FP64 atomics (default)",
        },
        "rccl_ranks": world if exchange_backend == "rccl" else 0,
        # device time of the all-reduces per step (HIP events around every exchange on rank 0's stream, instrumented repeat); None on one GPU
        "comm_ms_per_step": round(comm[0] / max(m["steps"], 1), 4) if comm and comm[1] else None,
        "setup_ms": round(m["setup_s"] * 1e3, 1),  # cba_create for this workload (sort, first Schur plan, upload): once per problem structure, not in `value`
        "setup_ms_warm": None if m["setup_warm_s"] != m["setup_warm_s"] else round(m["setup_warm_s"] * 1e3, 1),  # ... built again in the warm process (median of 3)
        "plan_wait_ms": round(m["plan_wait_s"] * 1e3, 1),  # ... and how much longer the balanced plan took (a solve would have started meanwhile)
        "final_rms_px": round(m["final_rms_px"], 6),
        "initial_rms_px": round(m["initial_rms_px"], 4),
        # accepted iterations re-linearise (njev - 1 of them after x0); the other trial points were rejected
        "timed_region": step_mix(m),
        "solve": {"nfev": m["full_nfev"], "njev": m["full_njev"], "accepted_steps": m["full_njev"] - 1,
                  "rejected_trials": m["full_nfev"] - m["full_njev"], "status": m["full_status"], "cost": m["full_cost"]},
        "roofline": roofline_from(m),
        "engine": m["info"],
        "library_source_sha256": _source_digest(),  # the code this line was measured on (profiles/parity_r*.json carries the same field)
    }
    if not args.no_cpu and rank == 0 and world == 1:
        try:
            out["cpu_baseline"], out["parity"] = cpu_baseline(m, device_id=local_rank)
        except Exception as exc:  # the baseline must never break the bench line
            out["cpu_baseline"] = {"value": None, "unit": "obs/s", "cores": 1, "kind": "port", "sample": f"failed: {exc!r}"}
    also = {}
    if rank == 0 and world == 1 and args.also:
        for name in [s for s in args.also.split(",") if s and s != args.workload]:
            try:
                a = measure(name, args.steps, args.warmup, device_id=local_rank, solve_kw={}, count="accepted",  # K ACCEPTED steps each
                            with_first_call=not args.no_first_call)
                also[name] = _also_block(a, name)
                if name == "cfg5" and not args.no_cpu:
                    also[name]["parity"] = cfg5_sample_parity(device_id=local_rank, n_points=args.cfg5_sample_points)
                if name == "cfg3" and not args.no_cpu:
                    also[name]["parity"] = cfg3_tight_parity(device_id=local_rank)
            except Exception as exc:
                also.setdefault(name, {})["error"] = repr(exc)
    if also:
        out["also"] = also
    line = None
    if rank == 0:
        line = json.dumps(out, default=lambda o: int(o) if isinstance(o, np.integer) else float(o))
    if control is not None:
        control.barrier()
        control.close()
    return line


if __name__ == "__main__":
    main()
