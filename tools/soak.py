#!/usr/bin/env python
"""Seam calls in a row (create / solve / destroy every time, the kept handle dropped each call): time per call, host RSS, device memory in use and
the process's threads BY NAME — whose they are — before the first handle, in steady state and after ``engine_cache.clear()`` (which calls
``cba_trim``: the library's pools go back).

    python tools/soak.py [calls=60]

Three problems in turn: full cfg4 (2M obs), half of it (1M), the cfg5 recipe at 1M observations (free intrinsics + bounds).  Same nfev and cost in
every call (asserted).  ``device in use`` is total - free of hipMemGetInfo for the whole device: the HIP runtime's own context (code objects, scratch,
queues) is in it, which is what the line measured BEFORE the first handle shows."""
import collections
import ctypes
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import psutil

import bench
from caliscope_amd import engine_cache
from caliscope_amd.least_squares import least_squares

proc = psutil.Process()
hip = ctypes.CDLL("libamdhip64.so")


def dev_in_use():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return (t.value - f.value) / 2**20


def threads_by_name():
    names = collections.Counter()
    for tid in os.listdir("/proc/self/task"):
        try:
            names[open(f"/proc/self/task/{tid}/comm").read().strip()] += 1
        except OSError:
            pass
    return dict(names.most_common())


def state(label):
    print(f"{label}: RSS {proc.memory_info().rss / 2**20:.0f} MB, device in use {dev_in_use():.0f} MB, threads {proc.num_threads()} {threads_by_name()}", flush=True)


n_calls = int(sys.argv[1]) if len(sys.argv) > 1 else 60
hip.hipSetDevice(0)
hip.hipFree(0)  # creates the context
state("HIP context only, before any handle")
probs = [bench.build_problem("cfg4"), bench.build_problem("cfg4", n_points=100_000, n_obs=1_000_000), bench.build_problem("cfg5", n_points=100_000, n_obs=1_000_000)]
state("problems generated (numpy / scipy threads appear here)")
ref, rows = {}, {0: [], 1: [], 2: []}
for i in range(n_calls):
    k = i % 3
    sc, par, x0, prob, cfg = probs[k]
    engine_cache.clear(trim=False)
    t = time.perf_counter()
    r = least_squares(None, x0, jac=None, bounds=par.bounds(), x_scale="jac", method="trf", args=(par, sc.camera_indices, sc.image_coords, sc.obj_indices))
    dt = (time.perf_counter() - t) * 1e3
    if k not in ref:
        ref[k] = (r.nfev, r.cost)
    if i >= 3:
        rows[k].append((dt, r.setup_seconds * 1e3, r.solve_seconds * 1e3))
    assert abs(r.nfev - ref[k][0]) <= 1 and abs(r.cost - ref[k][1]) <= 1e-9 * ref[k][1], (i, r.nfev, r.cost, ref[k])
    if i % 9 == 8 or i < 6:
        print(f"call {i:3d} (problem {k}): {dt:6.1f} ms (set-up {r.setup_seconds * 1e3:5.1f}, solve {r.solve_seconds * 1e3:5.1f})  RSS {proc.memory_info().rss / 2**20:7.0f} MB  "
              f"device in use {dev_in_use():7.0f} MB  threads {proc.num_threads()}", flush=True)
state("steady state, a handle kept")
for k in rows:
    a = np.array(rows[k])
    print(f"problem {k}: {len(a)} warm calls: end to end min / median / max {a[:, 0].min():.1f} / {np.median(a[:, 0]):.1f} / {a[:, 0].max():.1f} ms, set-up {a[:, 1].min():.1f} / "
          f"{np.median(a[:, 1]):.1f} / {a[:, 1].max():.1f}, solve {a[:, 2].min():.1f} / {np.median(a[:, 2]):.1f} / {a[:, 2].max():.1f}; solves above twice the median: "
          f"{int((a[:, 2] > 2 * np.median(a[:, 2])).sum())}")
engine_cache.clear(trim=False)
time.sleep(1.0)
state("after engine_cache.clear(trim=False): the handle is gone, the library's pools are kept")
released = engine_cache.clear()
time.sleep(1.0)
state(f"after engine_cache.clear() = cba_trim ({released / 2**20:.0f} MB released by the library)")
