"""Handle set-up against the number of host threads that deal the Schur plan (CBA_PLAN_THREADS): best of five creates per setting."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from caliscope_amd.hip_engine import HipEngine

for name, over in (("cfg4", dict(n_points=100_000, n_obs=1_000_000)), ("cfg4", {}), ("cfg5", {})):
    sc, par, x0, prob, cfg = bench.build_problem(name, **over)
    HipEngine(prob).close()
    for th in (0, 32, 64, 96, 128, 192, 256):
        if th: os.environ["CBA_PLAN_THREADS"] = str(th)
        else: os.environ.pop("CBA_PLAN_THREADS", None)
        best = 1e9
        for rep in range(3 if name == "cfg5" else 5):
            t = time.perf_counter(); e = HipEngine(prob); best = min(best, time.perf_counter() - t); e.close()
        print(f"{name} {over} plan threads {th or 'default'}: create {best * 1e3:.1f} ms", flush=True)
