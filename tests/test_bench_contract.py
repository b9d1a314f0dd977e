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


def test_committed_bench_line_has_the_contract_fields():
    line = (ROOT / "profiles" / "r01_end_bench.json").read_text().strip().splitlines()[-1]
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "observations/sec per LM iteration" and d["dtype"] == "f64" and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert "workload" in d["config"] and d["config"]["workload"].startswith("cfg4")
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"], key
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in d["cpu_baseline"], key
    assert abs(d["value"] - d["config"]["n_obs_total"] * 1e3 / d["ms_per_step"]) < 1e-3 * d["value"]
