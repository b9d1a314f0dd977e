"""bench.py's pure parts on the CPU: the algorithmic-bytes table (SURVEY.md 8d storage model), the roofline block built
from kernel-family timers, and the committed end-of-round bench line against the JSON contract."""
import importlib.util
import json
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


def test_committed_parity_at_size():
    """profiles/parity_r03.json (tools/parity_at_size.py, written by the round's last GPU run): GPU against scipy on BASELINE-sized inputs."""
    d = json.loads((ROOT / "profiles" / "parity_r03.json").read_text())
    c2 = d["cfg2"]  # both at 1e-13: a plain comparison
    assert abs(c2["d_rms_px"]) <= 1e-4 and c2["aligned_pos"] <= 1e-6 and c2["aligned_ang_rad"] <= 1e-6
    # cfg3 with the product's robust-stage settings (ftol 1e-4, max_nfev 60): the same trajectory, evaluation for evaluation
    c3p = d["cfg3_product"]
    assert c3p["scipy"]["nfev"] == c3p["gpu"]["nfev"] and c3p["scipy"]["njev"] == c3p["gpu"]["njev"] and abs(c3p["rel_cost"]) <= 1e-6
    # cfg3 to scipy's own convergence (534 evaluations): Huber on 17 px of noise, the valley is flat — scipy stops 1.15e-6 of the cost above the
    # minimum the product reaches; the RMS agrees within the 1e-4 px bar, and continued from scipy's answer the product ends where it ended from x0
    c3 = d["cfg3_converged"]
    assert c3["scipy"]["status"] > 0 and c3["gpu"]["status"] > 0 and abs(c3["d_rms_px"]) <= 1e-4 and c3["rel_cost"] <= 1e-9
    pol = d["cfg3_tight"]["polish"]
    assert pol["minimum_from_x0_vs_minimum_from_scipy"]["aligned_pos"] <= 1e-6 and pol["minimum_from_x0_vs_minimum_from_scipy"]["points_above_1e-6"] == 0
    assert 0.0 <= pol["rel_cost_scipy_above_minimum"] <= 1e-4 and abs(pol["rel_cost_gpu_above_minimum"]) <= 1e-9
    c5 = d["cfg5_sample_1M"]
    assert abs(c5["d_rms_px"]) <= 1e-4 and c5["rel_cost"] <= 1e-9 and c5["same_minimum_within_north_star"] is True
    assert c5["polish"]["same_minimum"]["aligned_pos"] <= 1e-6 and c5["polish"]["same_minimum"]["points_above_1e-6"] == 0
