"""CBA_PLAN=swap (opt-in): a handle starts with the cheap Schur plan and swaps the dealt one in when its thread is done.  Per mode: handle set-up (best
of five), then solves on one handle — the first right away, the later ones after the plan thread had time — with the pair kernel's time per launch."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench
from caliscope_amd.hip_engine import HipEngine

for name, over in (("cfg4", dict(n_points=100_000, n_obs=1_000_000)), ("cfg4", {}), ("cfg5", dict(n_points=100_000, n_obs=1_000_000))):
    sc, par, x0, prob, cfg = bench.build_problem(name, **over)
    kw = {}
    if par.has_finite_bounds:
        lb, ub = par.bounds()
        kw = dict(lb=np.ascontiguousarray(lb[: par.n_camera_params]), ub=np.ascontiguousarray(ub[: par.n_camera_params]))
    HipEngine(prob).close()
    ref = None
    for mode in ("full", "swap", "cheap"):
        os.environ["CBA_PLAN"] = mode
        best = 1e9
        for rep in range(5):
            t = time.perf_counter(); e = HipEngine(prob); best = min(best, time.perf_counter() - t); e.close()
        t = time.perf_counter(); e = HipEngine(prob); t_create = time.perf_counter() - t
        rows = []
        for k in range(3):
            e.enable_timers(True); e.reset_timers()
            t = time.perf_counter(); r = e.solve(x0, **kw); dt = time.perf_counter() - t
            tm = e.timers()
            pairs = tm["schur_pairs"][0] / max(tm["schur_pairs"][1], 1) * 1e3
            rows.append(f"solve {k}: {dt * 1e3:.1f} ms, nfev {r.nfev}, cost {r.cost:.12e}, pair kernel {pairs:.0f} us")
            if ref is None: ref = r.cost
            assert abs(r.cost - ref) <= 1e-9 * ref, (mode, r.cost, ref)
            time.sleep(0.5 if name != "cfg5" else 1.0)
        e.close()
        print(f"{name} {over} CBA_PLAN={mode}: create best of five {best * 1e3:.1f} ms (this handle {t_create * 1e3:.1f}); " + "; ".join(rows), flush=True)
    os.environ.pop("CBA_PLAN")
