#!/usr/bin/env python
"""What one rank of an N-rank sharded solve costs per iteration, measured ALONE on one MI355X, and the strong-scaling projection built from it.

    python tools/shard_projection.py [cfg4|cfg5] [steps]

For N = 1, 2, 4, 8 the script takes rank 0's shard of the workload (caliscope_amd/sharding.py: contiguous point ranges balanced by observations,
cameras replicated) and runs the SHARDED code path on it — a one-rank RCCL communicator (CBA_FORCE_COMM=1): the same kernels, the same packing and
the same number of collectives as a rank of a real N-rank solve, with nobody else on the device and nothing on the wire.  HIP-event timers give
the per-family device time per iteration; the wall time per iteration includes the launch overhead of RCCL's one-rank collectives.

Projection to N GPUs (no multi-GPU node was available to any round; `bench.py --gpus N --devices 0,0,.. --xchg direct` runs the protocol at world 8
on one device for CORRECTNESS — same final RMS and evaluation count — but its timings are 8 ranks sharing one GPU):
    t(N) = wall per iteration of shard 0 of N on its own + wire time of the exchanged bytes,
    wire = (packed reduced system + camera blocks + scalars) x 2 (N - 1) / N / (link bandwidth x links used), direct exchange over xGMI
           (reduce-scatter + all-gather over the 7 links, 153 GB/s each, taken at 60 % efficiency) + 4 collectives x 10 us latency.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["CBA_FORCE_COMM"] = "1"
import numpy as np

import bench
from caliscope_amd.hip_engine import HipEngine
from caliscope_amd.sharding import shard_problem

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else (20 if name != "cfg5" else 8)
sc, par, x0, prob, cfg = bench.build_problem(name)
solve_kw = {}
if par.has_finite_bounds:
    lb, ub = par.bounds()
    solve_kw.update(lb=np.ascontiguousarray(lb[: par.n_camera_params]), ub=np.ascontiguousarray(ub[: par.n_camera_params]))
ncp = par.n_camera_params
nct = 9 if any(b.n_params == 9 for b in par.blocks) else 6
exchanged = 8 * (ncp * (ncp + 1) // 2 + ncp + len(par.blocks) * (nct * (nct + 1) // 2 + nct) + 64)  # bytes per iteration: packed S | b, camera blocks, scalars
out = {"workload": name, "steps": steps, "exchanged_bytes_per_iteration": exchanged, "ranks": {}}
t1 = None
for world in (1, 2, 4, 8):
    shard = shard_problem(prob, 0, world)
    eng = HipEngine(shard.problem, device_id=0)
    eng.comm_init(eng.comm_unique_id(), 0, 1)  # one-rank communicator: the sharded route of the library, nothing on the wire
    eng.plan_wait()
    eng.begin(shard.local_x(x0))
    kw = dict(solve_kw)
    bench.run_iterations(eng, 4, kw, "accepted")
    t0 = time.perf_counter()
    _, last = bench.run_iterations(eng, steps, kw, "accepted")
    dt = time.perf_counter() - t0
    eng.enable_timers(True)
    eng.reset_timers()
    _, last_t = bench.run_iterations(eng, steps, kw, "accepted")
    tm = eng.timers()
    eng.enable_timers(False)
    eng.close()
    acc = max(last.mix["accepted"], 1)
    acc_t = max(last_t.mix["accepted"], 1)
    ms = dt / acc * 1e3
    wire_ms = 0.0 if world == 1 else (exchanged * 2 * (world - 1) / world / (0.6 * 153e9 * min(world - 1, 7)) + 4 * 10e-6) * 1e3
    fam = {k: round(v[0] / acc_t * 1e3, 1) for k, v in tm.items() if v[1]}  # us per accepted iteration
    t_n = ms + wire_ms
    if world == 1:
        t1 = t_n
    out["ranks"][world] = {"obs_on_rank0": int(shard.problem.n_obs), "ms_per_iteration_alone": round(ms, 4), "wire_ms_estimate": round(wire_ms, 4),
                           "projected_ms": round(t_n, 4), "projected_speedup_vs_1": round(t1 / t_n, 2), "device_us_per_iteration": fam}
    print(world, out["ranks"][world], flush=True)
print(json.dumps(out))
