"""Keep the last device handle(s) alive between ``least_squares`` calls.

``calibrate_extrinsics`` calls ``CaptureVolume.optimize()`` several times in a row (reference core/calibrate_extrinsics.py:206-250);
two consecutive stages often solve the SAME observations with another loss or another starting point.  Building a handle is the
expensive part of a small solve (sorting the observations, the Schur plan, ~60 device buffers: 90 ms at 2M observations against a
10 ms solve), so the seam looks a finished handle up by a fingerprint of everything ``cba_create`` consumed — camera tables,
observation arrays, constraint rows, device, deterministic flag — and, on a hit, only changes the loss (``cba_set_loss``).

``CALISCOPE_HIP_ENGINE_CACHE`` = number of handles kept (default 1, 0 disables).  A handle in use is taken out of the cache, so two
threads never share one; :func:`clear` (also run at interpreter exit) destroys what is kept.  A kept handle keeps its device memory
(records and plan: ~0.6 GB at 2M observations, ~4 GB at 10M): handles above ``CALISCOPE_HIP_ENGINE_CACHE_MAX_GB`` (default 16) are
destroyed at check-in instead of kept.  The fingerprint uses ``xxhash`` when it is installed and ``hashlib.blake2b`` otherwise.
"""

from __future__ import annotations

import atexit
import os
import threading
from collections import OrderedDict

import numpy as np

from caliscope_amd.bundle_parameterization import device_tables

_lock = threading.Lock()
_kept: "OrderedDict[bytes, object]" = OrderedDict()
stats = {"hits": 0, "misses": 0}


def _capacity() -> int:
    try:
        return max(0, int(os.environ.get("CALISCOPE_HIP_ENGINE_CACHE", "1")))
    except ValueError:
        return 1


def _max_bytes() -> int:
    try:
        return int(float(os.environ.get("CALISCOPE_HIP_ENGINE_CACHE_MAX_GB", "16")) * (1 << 30))
    except ValueError:
        return 16 << 30


def _hasher():
    try:
        import xxhash  # optional: ~10x faster than blake2b on the observation arrays

        return xxhash.xxh3_128()
    except ImportError:
        import hashlib

        return hashlib.blake2b(digest_size=16)


def fingerprint(problem, device_id: int, deterministic: bool) -> bytes:
    """Digest of what the handle was built from (not of loss / f_scale: those can be changed on a live handle)."""
    h = _hasher()
    par = problem.parameterization
    tabs = device_tables(par)
    h.update(np.array([device_id, int(deterministic), len(par.blocks), par.n_points, problem.n_obs, problem.n_constraints], dtype=np.int64).tobytes())
    h.update(repr(sorted((k, v) for k, v in os.environ.items() if k.startswith("CBA_"))).encode())  # the library's switches are read at cba_create
    for key, dt in (("cam_n_params", np.int32), ("cam_model", np.int32), ("cam_const", np.float64)):
        h.update(np.ascontiguousarray(tabs[key], dtype=dt).tobytes())
    for a in (problem.camera_indices, problem.obj_indices, problem.image_coords):
        h.update(memoryview(np.ascontiguousarray(a)).cast("B"))
    if problem.n_constraints:
        for a in problem.constraint_args():
            h.update(memoryview(np.ascontiguousarray(a)).cast("B"))
    return h.digest()


def checkout(problem, device_id: int = -1):
    """A handle for `problem` (its loss set), and the key to hand back to :func:`checkin`."""
    from caliscope_amd.hip_engine import HipEngine

    deterministic = os.environ.get("CBA_DETERMINISTIC", "0") not in ("", "0")
    if _capacity() == 0:
        return HipEngine(problem, device_id=device_id), None
    key = fingerprint(problem, device_id, deterministic)
    with _lock:
        engine = _kept.pop(key, None)
    if engine is not None:
        try:
            engine.set_loss(problem.loss, problem.f_scale)
        except Exception:
            engine.close()
            raise
        stats["hits"] += 1
        return engine, key
    stats["misses"] += 1
    return HipEngine(problem, device_id=device_id), key


def checkin(key, engine) -> None:
    """Keep `engine` for the next call (or destroy it: cache off, or more handles than the capacity)."""
    cap = _capacity()
    if key is None or cap == 0:
        engine.close()
        return
    try:
        too_big = int(engine.info()["device_bytes"]) > _max_bytes()
    except Exception:
        too_big = True
    if too_big:
        engine.close()
        return
    evicted = []
    with _lock:
        old = _kept.pop(key, None)
        if old is not None:
            evicted.append(old)
        _kept[key] = engine
        while len(_kept) > cap:
            evicted.append(_kept.popitem(last=False)[1])
    for e in evicted:
        e.close()


def clear(trim: bool = True) -> int:
    """Destroy the kept handle(s) and, `trim`, give back what the library pools between handles (``cba_trim``: arena chunks, streams, pinned
    staging, the huge-page host blocks of the set-up).  Returns the bytes the library released."""
    with _lock:
        engines = list(_kept.values())
        _kept.clear()
    for e in engines:
        e.close()
    if not trim:
        return 0
    try:
        from caliscope_amd import _lib

        if _lib._lib is None:  # never loaded in this process (interpreter exit without a solve): nothing pooled, and loading it would start the runtime
            return 0
        return int(_lib._lib.cba_trim())
    except Exception:  # noqa: BLE001 - the library may be absent (interpreter exit on a box without it)
        return 0


atexit.register(clear)
