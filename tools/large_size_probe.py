#!/usr/bin/env python
"""One MI355X, a problem four times BASELINE's largest configuration: the cfg5 recipe (128 nine-parameter cameras, joint intrinsics with
bounds) at 4M points / 40M observations.  Says what the set-up costs at that size, what the handle holds in HBM, what an accepted iteration
takes and that the iterations do what they do at 10M (cost falls, same evaluations to convergence as the recipe's smaller sizes).

    python tools/large_size_probe.py [n_obs=40000000] [accepted_steps=6]
"""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import bench
from caliscope_amd.hip_engine import HipEngine

n_obs = int(float(sys.argv[1])) if len(sys.argv) > 1 else 40_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
t = time.perf_counter()
sc, par, x0, prob, cfg = bench.build_problem("cfg5", n_points=n_obs // 10, n_obs=n_obs)
print(f"scene: {len(par.blocks)} cameras, {par.n_points} points, {prob.n_obs} observations, {par.n_params} parameters; generated in {time.perf_counter() - t:.1f} s", flush=True)
lb, ub = par.bounds()
ncp = par.n_camera_params
kw = dict(lb=np.ascontiguousarray(lb[:ncp]), ub=np.ascontiguousarray(ub[:ncp]))
t = time.perf_counter()
eng = HipEngine(prob, device_id=0)
t_create = time.perf_counter() - t
t = time.perf_counter()
eng.plan_wait()
t_plan = time.perf_counter() - t
info = eng.info()
print(f"cba_create {t_create * 1e3:.0f} ms (quick plan), balanced plan {t_plan * 1e3:.0f} ms later; handle holds {info['device_bytes'] / 2**30:.2f} GiB of HBM "
      f"({info['device_bytes'] / prob.n_obs:.0f} bytes per observation), {info['n_chunks']} chunks, {info['schur_pairs']} pairs, plan_state {info['plan_state']}", flush=True)
eng.begin(x0)
rms0 = bench.rms_px(eng, par, x0, sc.camera_indices)
bench.run_iterations(eng, 2, kw, "accepted")
t = time.perf_counter()
_, last = bench.run_iterations(eng, steps, kw, "accepted")
dt = time.perf_counter() - t
acc = max(last.mix["accepted"], 1)
print(f"{acc} accepted iterations ({last.mix['rejected']} rejected trials) in {dt * 1e3:.1f} ms: {dt / acc * 1e3:.2f} ms per iteration = "
      f"{prob.n_obs / (dt / acc):.3e} observations/s per iteration (cfg5 at 10M: 4.5 ms = 2.2e9)", flush=True)
t = time.perf_counter()
res = eng.solve(x0, **kw)
dt = time.perf_counter() - t
rms1 = bench.rms_px(eng, par, res.x, sc.camera_indices)
print(f"full solve from x0: status {res.status}, {res.nfev} evaluations, {dt * 1e3:.0f} ms, RMS {rms0:.4f} -> {rms1:.6f} px, cost {res.cost:.6e}", flush=True)
eng.close()
