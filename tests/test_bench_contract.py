"""bench.py's pure parts on the CPU: the algorithmic-bytes table (SURVEY.md 8d storage model), the roofline block built
from kernel-family timers, and the committed end-of-round bench line against the JSON contract."""
import importlib.util
import json

import pytest
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", ROOT / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_algorithmic_bytes_follow_the_storage_model():
    b = _bench().algorithmic_bytes(2_000_000, 200_000, 64, 384, 6)
    N, P = 2_000_000, 200_000
    assert b["cost"] == 24 * N + 24 * P
    assert b["build"] == 24 * N + 96 * P + 8 * 64 * 27  # packed camera block: upper triangle + gradient = 27 doubles
    assert b["schur"] == 24 * N + 120 * P + 8 * 384 * 384 + 8 * 384
    assert b["backsub"] == 24 * N + 144 * P and b["jv"] == 24 * N + 48 * P


def test_roofline_block_from_timers():
    bench = _bench()
    m = {"timers": {"schur": (15.0, 40), "build": (3.6, 40), "cost": (0.8, 40), "vector_ops": (0.5, 100)}, "n_obs": 2_000_000, "n_points": 200_000,
         "n_cams": 64, "ncp": 384, "nct": 6, "name": "cfg4", "elapsed": 0.04, "steps": 40, "info": {"schur_pairs": 11_000_000}}
    r = bench.roofline_from(m)
    assert r["bound"] == "hbm" and r["kernel"] == "k_schur" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    alg = bench.algorithmic_bytes(2_000_000, 200_000, 64, 384, 6)["schur"]
    assert abs(r["achieved"] - alg / (15.0 / 40 * 1e-3) / 1e9) < 0.1 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-4
    assert r["iteration"]["alg_bytes"] == 72 * 2_000_000 + 312 * 200_000 + 8 * 384**2
    assert r["fp64_valu"]["flops_per_launch"] == 11_000_000 * 6 * 36 and 0 < r["fp64_valu"]["frac"] < 1
    assert bench.roofline_from({**m, "timers": {}}) is None


def test_run_iterations_counts_trials_or_accepted_steps():
    """bench.run_iterations: the headline executes exactly K trial points (the metric's unit); the `also` workloads go on until K ACCEPTED steps have
    been made, however many rejected trials lie between them (round 4's cfg3 line held 2 accepted steps at K = 20)."""
    from types import SimpleNamespace

    bench = _bench()

    class Fake:
        """A solve of at most max_nfev evaluations whose every third trial is rejected."""

        def __init__(self):
            self.calls = []

        def solve(self, x0, max_nfev, fetch_x, **kw):
            trials = max_nfev - 1
            rejected = trials // 3
            self.calls.append(max_nfev)
            return SimpleNamespace(nfev=trials + 1, njev=trials - rejected + 1, rejected_timed=rejected, rejected_seconds=1e-5 * rejected)

    e = Fake()
    solves, last = bench.run_iterations(e, 20, {}, "trials")
    assert last.mix["trials"] == 20 and last.mix["accepted"] + last.mix["rejected"] == 20 and solves == 1
    e = Fake()
    solves, last = bench.run_iterations(e, 20, {}, "accepted")
    assert last.mix["accepted"] >= 20 and last.mix["trials"] == last.mix["accepted"] + last.mix["rejected"] and last.mix["rejected"] > 0


def test_valu_floor_from_sq_counters():
    """roofline.iteration.valu_floor_us: wave-level VALU instructions x 4 clocks / 1024 SIMDs / 2.4 GHz, summed over the kernels of one accepted iteration
    (round 4's counters on cfg4: 111 us against the 26 us of the algorithmic bytes at 8 TB/s — FP64 issue, not HBM, is the lower roof)."""
    bench = _bench()
    rows = {"k_backsub<6, false>": {"SQ_INSTS_VALU": 8.155e6}, "k_build_cs<6, false, false>": {"SQ_INSTS_VALU": 1.004e7}, "k_chol_step": {"SQ_INSTS_VALU": 24430.0},
            "k_jv<6, 1, false>": {"SQ_INSTS_VALU": 6.575e6}, "k_schur_reg3<6, 1, 2>": {"SQ_INSTS_VALU": 3.029e7}, "k_tprep<6, 0, false>": {"SQ_INSTS_VALU": 1.314e7}}
    vf = bench.valu_floor("cfg4", 384, rows)
    assert abs(vf["per_kernel_us"]["k_schur_reg3"] - 3.029e7 * 4 / 1024 / 2400) < 0.06
    assert abs(vf["per_kernel_us"]["k_chol_step"] - 13 * 24430.0 * 4 / 1024 / 2400) < 0.06
    assert 105.0 < vf["iteration_us"] < 118.0
    assert bench.valu_floor("no-such-workload", 384) is None


def _latest_bench_line():
    files = sorted((ROOT / "profiles").glob("r[0-9][0-9]_end_bench.json"))
    assert files, "no committed bench line under profiles/"
    return files[-1].name, json.loads(files[-1].read_text().strip().splitlines()[-1])


def test_committed_bench_line_has_the_contract_fields():
    """The newest end-of-round bench line under profiles/ (written by the round's last GPU run, tools/gpu_profile_run.sh) against the JSON contract
    of the task: the driver's keys, the roofline and cpu_baseline objects, and — from round 3 on — the parity legs on BASELINE-sized inputs."""
    name, d = _latest_bench_line()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "observations/sec per LM iteration" and d["dtype"] == "f64" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert d["scaling"] in ("strong", "weak") and d["n_gpus"] == 1 and d["data"] == "synthetic"
    assert "workload" in d["config"] and d["config"]["workload"].startswith("cfg4") and "model" not in d["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"], key
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0 and abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / 8000.0) < 1e-3
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in d["cpu_baseline"], key
    assert d["cpu_baseline"]["kind"] in ("port", "reference") and d["cpu_baseline"]["cores"] == 1
    assert abs(d["value"] - d["config"]["n_obs_total"] * 1e3 / d["ms_per_step"]) < 1e-3 * d["value"]
    if name < "r03":
        return
    # round 3: the scipy leg runs on the headline's own arrays; cfg5 carries its roofline block and a parity sample of its recipe
    for key in ("rccl_ranks", "comm_ms_per_step", "setup_ms", "parity"):
        assert key in d, key
    par = d["parity"]
    assert "full size" in d["cpu_baseline"]["sample"] and "headline" in par["sample"]
    for key in ("d_rms_px", "rel_cost", "aligned_pos", "aligned_ang_rad", "within_north_star", "detail", "gpu"):
        assert key in par, key
    assert abs(par["d_rms_px"]) <= 1e-4 and par["aligned_pos"] <= 1e-6 and par["aligned_ang_rad"] <= 1e-6 and par["within_north_star"] is True
    for key in ("seconds_end_to_end", "setup_ms", "value_end_to_end"):
        assert key in par["gpu"], key
    c5 = d["also"]["cfg5"]
    for key in ("bound", "achieved", "peak", "frac", "traffic", "kernels"):
        assert key in c5["roofline"], key
    # cfg5 recipe (free intrinsics + bounds): scipy's LSMR steps stop on ftol a few 1e-6 of the cost ABOVE the minimum (123 evaluations), so at default
    # tolerances the two answers differ by ~1e-5 in the weakest camera direction while the RMS agrees to 1e-6 px.  What the line must show is that
    # both end in the SAME minimum: the product continued at 1e-13 from scipy's stopping point lands on the product's own answer from x0.
    c5p = c5["parity"]
    assert abs(c5p["d_rms_px"]) <= 1e-4 and c5p["rel_cost"] <= 1e-9  # the product's cost is not above scipy's
    assert c5p["same_minimum_within_north_star"] is True
    same = c5p["polish"]["same_minimum"]
    assert same["aligned_pos"] <= 1e-6 and same["aligned_ang_rad"] <= 1e-6 and same["points_above_1e-6"] == 0
    assert 0.0 <= c5p["polish"]["rel_cost_scipy_above_minimum"] <= 1e-4
    assert "refine_intrinsics=True" in c5p["sample"]
    assert par["same_minimum_within_north_star"] is True and par["polish"]["same_minimum"]["aligned_pos"] <= 1e-6


def _oracle_polish_ok(block):
    """scipy restarted AT the product's converged x with 1e-15 tolerances: it must stop having moved less than 1e-6 and gained less than 1e-12."""
    assert block["scipy_moves_within_1e-6_and_gains_within_1e-12"] is True
    for leg in ("lsmr_default", "lsmr_tight"):
        if block.get(leg):
            assert block[leg]["moved"]["aligned_pos"] <= 1e-6 and block[leg]["moved"]["points_above_1e-6"] == 0 and block[leg]["rel_cost_gain"] <= 1e-12


def test_round4_bench_line_closes_parity_from_the_oracle_side():
    """Round 4: the bench line separates accepted steps from rejected trials, names the library sources it ran on, and shows parity in BOTH directions
    (product against scipy from x0; scipy restarted at the product's answer does not move)."""
    name, d = _latest_bench_line()
    if name < "r04":
        pytest.skip("no round-4 bench line committed yet")
    parity_file = "parity_r04.json" if name < "r05" else _latest_parity_file()[0]
    for key in ("timed_region", "library_source_sha256", "plan_wait_ms"):
        assert key in d, key
    tr = d["timed_region"]
    assert tr["accepted_steps"] + tr["rejected_trials"] >= d["steps"] and abs(tr["ms_per_accepted_step"] - d["ms_per_step"]) < 1e-9
    _oracle_polish_ok(d["parity"]["oracle_polish"])
    c5p = d["also"]["cfg5"]["parity"]
    _oracle_polish_ok(c5p["oracle_polish"])
    # cfg5 recipe against scipy run to ITS minimum (1e-15, inner LSMR 1e-14; tests/golden/scipy_refs): a plain comparison inside north_star
    tight = c5p["tight_reference"]["detail"]
    assert tight["aligned_pos"] <= 1e-6 and tight["aligned_ang_rad"] <= 1e-6 and tight["points_above_1e-6"] == 0
    c3 = d["also"]["cfg3"]["timed_region"]  # Huber: rejected trials are timed on their own, not averaged into the step
    assert c3["rejected_trials_timed"] > 0 and c3["ms_per_rejected_trial"] < c3["ms_per_accepted_step"]
    # the parity file of the same run was written on the same library sources
    par = json.loads((ROOT / "profiles" / parity_file).read_text())
    assert par["library_source_sha256"] == d["library_source_sha256"]


def _latest_parity_file():
    files = sorted((ROOT / "profiles").glob("parity_r[0-9][0-9].json"))
    assert files
    return files[-1].name, json.loads(files[-1].read_text())


# KNOWN PARITY GAP (recorded in DESIGN.md 6 and VERDICT r04): north_star's literal 1e-6 on every converged world point is NOT met on cfg3 (Huber, 5 %
# outliers): one of its 50 000 points, all of whose observations sit in Huber's linear region, ends 1.5e-6 from scipy's position.  The suite does not
# widen the bar for it: a case listed here must (a) say so itself (`within_north_star` false), (b) list EVERY point above 1e-6 in `weak_points` and
# (c) show for each of them that moving it, alone, to the reference's position changes the ORACLE's cost by less than 1e-12 relative — i.e. the data
# cannot tell the two positions apart.  Cameras, the 99.9 % quantile of the points, RMS and cost must still meet the bar.  Nothing else is excused.
KNOWN_POINT_GAPS = {"cfg3_tight"}


def _converged_case_ok(name, c):
    assert abs(c["d_rms_px"]) <= 1e-4 and c["detail"]["aligned_ang_rad"] <= 1e-6 and c["detail"]["cameras_pos"] <= 1e-6 and abs(c["rel_cost"]) <= 1e-9, name
    assert c["detail"]["points_pos_p999"] <= 1e-6, name
    n_above = c["detail"]["points_above_1e-6"]
    if n_above == 0:
        assert c["within_north_star"] is True, name
        return
    assert name in KNOWN_POINT_GAPS, f"{name}: {n_above} points above 1e-6 and the case is not a recorded gap"
    assert c["within_north_star"] is False, name  # the file says it, too
    wp = c["weak_points"]
    assert wp["points_above_1e-6"] == n_above == len(wp["listed"]), name  # every one of them is listed ...
    for pt in wp["listed"]:                                                # ... and indistinguishable to the data
        assert abs(pt["rel_cost_change_if_moved_to_the_reference_position"]) <= 1e-12, (name, pt)


def test_committed_parity_at_size():
    """profiles/parity_r*.json, the newest (tools/parity_at_size.py, written by the round's last GPU run): GPU against scipy on BASELINE-sized inputs,
    both directions of the converged-level protocol.  The 1e-6 bar is asserted as north_star states it; the one recorded exception is handled by
    _converged_case_ok and by nothing looser."""
    name, d = _latest_parity_file()
    for case in ("cfg2", "cfg3_tight"):  # both sides run to their minimum: a plain comparison, and scipy restarted at the product's answer stays put
        c = d[case]
        _converged_case_ok(case, c)
        _oracle_polish_ok(c["oracle_polish"])
    # cfg3 with the product's robust-stage settings (ftol 1e-4, max_nfev 60): the same trajectory, evaluation for evaluation
    c3p = d["cfg3_product"]
    assert c3p["scipy"]["nfev"] == c3p["gpu"]["nfev"] and c3p["scipy"]["njev"] == c3p["gpu"]["njev"] and abs(c3p["rel_cost"]) <= 1e-6
    # cfg3 at the reference's default tolerances: scipy's LSMR steps stop on ftol ~1e-6 of the cost above the minimum the product reaches (cfg3_tight
    # is the comparison at the minimum); the RMS agrees within the 1e-4 px bar and the product's cost is not above scipy's
    c3 = d["cfg3_default"]
    assert c3["scipy"]["status"] > 0 and c3["gpu"]["status"] > 0 and abs(c3["d_rms_px"]) <= 1e-4 and c3["rel_cost"] <= 1e-9
    for case in ("cfg5_sample_100k", "cfg5_sample_1M"):
        c5 = d[case]
        assert abs(c5["d_rms_px"]) <= 1e-4 and c5["rel_cost"] <= 1e-9 and c5["same_minimum_within_north_star"] is True, case
        tight = c5["tight_reference"]["detail"]
        assert tight["aligned_pos"] <= 1e-6 and tight["aligned_ang_rad"] <= 1e-6 and tight["points_above_1e-6"] == 0, case
        _oracle_polish_ok(c5["oracle_polish"])


def test_round5_bench_line_says_what_a_caller_gets():
    """Round 5: the line names the Schur plan its timed steps ran on and carries the same steps on the quickly made plan a FIRST
    optimize() call runs on; the `also` workloads are timed over K ACCEPTED steps; cfg3 run to the minimum is compared with the stored scipy solve in the
    driver's own line; stored scipy timings are marked as stored; the default-tolerance distance comes with its reading."""
    name, d = _latest_bench_line()
    if name < "r05":
        pytest.skip("no round-5 bench line committed yet")
    tr = d["timed_region"]
    assert tr["plan"] == "dealt" and tr["counted"] == "trials" and tr["trial_points"] == d["steps"]
    fc = tr["first_call"]
    assert fc["plan"] == "cheap" and fc["ms_per_step"] >= 0.9 * d["ms_per_step"] and fc["value"] > 1e7  # (the quick plan's pair kernel is slower, never faster)
    for wl, a in d["also"].items():
        t = a["timed_region"]
        assert t["counted"] == "accepted" and t["accepted_steps"] >= d["steps"], wl
    c3 = d["also"]["cfg3"]["parity"]
    assert c3["scipy_stored"] is True and c3["cameras_within_1e-6"] is True and abs(c3["d_rms_px"]) <= 1e-4 and abs(c3["rel_cost"]) <= 1e-9
    _converged_case_ok("cfg3_tight", c3)
    assert c3["oracle_polish"]["scipy_moves_within_1e-6_and_gains_within_1e-12"] is True
    assert d["cpu_baseline"]["stored"] is False  # the headline's scipy leg is timed live on the box
    c5 = d["also"]["cfg5"]["parity"]
    assert c5["value_ratio_uses_stored_cpu_seconds"] is True
    dist = d["parity"]["default_tolerance_distance"]
    assert "reading" in dist and dist["scipy_rel_cost_above_the_minimum"] >= -1e-12 and dist["product_rel_cost_above_the_minimum"] <= 1e-9
    par_name, par = _latest_parity_file()
    assert par_name >= "parity_r05.json" and par["library_source_sha256"] == d["library_source_sha256"]
    full = par.get("cfg5_full_linear_algebra")  # (round 6) FULL cfg5 through size-independent properties against the oracle's rows and sparse Jacobian
    if full is not None:
        assert "10000000 obs" in full["workload"] and full["step_ok"] is True
        assert full["residual_rows_max_abs_diff"] <= 1e-11 and full["cost_rel_diff"] <= 1e-12 and full["gradient_rel_inf"] <= 1e-10
        assert full["jacobi_scale_rel_max"] <= 1e-10 and full["normal_equation_residual_of_the_device_step_rel_inf"] <= 1e-8


def test_weak_points_reports_what_the_data_say_about_a_displaced_point():
    """bench.weak_points (parity_at_size.py, round 4): for every point two answers place more than 1e-6 apart it moves THAT point, alone, to the
    reference's position in the product's gauge and reports the oracle's cost change.  A well-observed point moved by 3e-5 changes the cost visibly
    (not "indistinguishable"); a gauge change of the whole reference (rotation, scale, shift) is aligned away and flags nothing."""
    import numpy as np
    from helpers import small_problem

    bench = _bench()
    sc, par, x0 = small_problem(n_cams=5, n_points=120, k=5)
    n = par.n_camera_params
    ref = x0.copy()
    ref[n + 3 * 7: n + 3 * 7 + 3] += [3e-5, 0.0, 0.0]
    out = bench.weak_points(sc, par, x0, ref, "linear", 1.0)
    assert out["points_above_1e-6"] == 1 and out["listed"][0]["point"] == 7 and out["listed"][0]["observations"] >= 1
    assert abs(out["listed"][0]["rel_cost_change_if_moved_to_the_reference_position"]) > 1e-9 and out["all_listed_indistinguishable_at_1e-12_of_the_cost"] is False  # (x0 is no minimum: either sign)
    # the same answer in another gauge: points and camera centres rotated, scaled and shifted together
    from caliscope_amd.cameras import matrix_to_rvec, rvec_to_matrix

    Q = rvec_to_matrix(np.array([0.1, -0.2, 0.05]))
    sc_g, t_g = 1.3, np.array([0.4, -0.1, 0.2])
    moved = x0.copy()
    moved[n:] = (sc_g * x0[n:].reshape(-1, 3) @ Q.T + t_g).ravel()
    for off in par.camera_param_offsets:
        R = rvec_to_matrix(x0[off:off + 3])
        centre = -R.T @ x0[off + 3:off + 6]
        R_new = R @ Q.T                                 # X_cam = R (X - c)  ->  R Q^T (X' - c'), X' = s Q X + t (up to the scale, which the alignment takes out)
        moved[off:off + 3] = matrix_to_rvec(R_new)
        moved[off + 3:off + 6] = -R_new @ (sc_g * Q @ centre + t_g)
    same = bench.weak_points(sc, par, x0, moved, "linear", 1.0)
    assert same["points_above_1e-6"] == 0 and same["listed"] == [] and same["all_listed_indistinguishable_at_1e-12_of_the_cost"] is True
