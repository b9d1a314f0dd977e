#!/usr/bin/env python
"""Whose device memory is it?  ``hipMemGetInfo`` (total - free, whole device) after each stage of a handle's life, so that what stays in use after
``engine_cache.clear()`` / ``cba_trim`` can be told apart: the HIP runtime's own (context, code objects of the loaded library, queues, scratch) or
the library's (arena chunks, kept pools).

    python tools/device_memory_probe.py

Stages: HIP context; the library loaded (its code object, 114 kernels); a stream created and destroyed; a tiny handle (cfg2) built, solved,
destroyed, trimmed; then the same with cfg4 (2M observations)."""
import ctypes
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
hip = ctypes.CDLL("libamdhip64.so")
last = [None]


def used(label):
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipDeviceSynchronize()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    mb = (t.value - f.value) / 2**20
    print(f"{label:78s} {mb:8.0f} MB in use" + ("" if last[0] is None else f"  ({mb - last[0]:+.0f})"), flush=True)
    last[0] = mb


hip.hipSetDevice(0)
hip.hipFree(0)
used("HIP context (hipFree(0))")
from caliscope_amd import _lib, engine_cache  # noqa: E402

lib = _lib.load()
lib.cba_device_count()
used("library loaded (dlopen registers its code object; nothing launched yet)")
st = ctypes.c_void_p()
hip.hipStreamCreate(ctypes.byref(st))
used("one more HIP stream created")
hip.hipStreamDestroy(st)
used("... and destroyed")

import bench  # noqa: E402
from caliscope_amd.hip_engine import HipEngine  # noqa: E402

for name in os.environ.get("PROBE_CFGS", "cfg2,cfg4,cfg4").split(","):
    sc, par, x0, prob, cfg = bench.build_problem(name)
    used(f"{name}: problem generated on the host")
    eng = HipEngine(prob, device_id=0)
    used(f"{name}: handle created (cba_create: arena, records, quick plan)")
    eng.plan_wait()
    used(f"{name}: balanced plan installed (cba_plan_wait)")
    res = eng.solve(x0)
    used(f"{name}: solved ({res.nfev} evaluations; every kernel of the route has run once)")
    info = eng.info()
    print(f"    cba_info: device bytes of the handle {info.get('device_bytes', 0) / 2**20:.0f} MB", flush=True)
    eng.close()
    used(f"{name}: handle destroyed (first arena chunk, stream and mailbox go to the library's pool)")
    released = engine_cache.clear()
    used(f"{name}: cba_trim ({released / 2**20:.0f} MB of host + device pools released by the library)")
