"""Host logic of caliscope_amd.engine_cache that needs no device: the fingerprint's hash fallback and the size bound of kept handles."""

import sys

import numpy as np

from caliscope_amd import engine_cache
from tests.helpers import small_problem
from caliscope_amd.engine import BAProblem


class _FakeEngine:
    def __init__(self, device_bytes):
        self.device_bytes, self.closed = device_bytes, False

    def info(self):
        return {"device_bytes": self.device_bytes}

    def close(self):
        self.closed = True


def _problem():
    sc, par, x0 = small_problem(n_cams=3, n_points=20, k=3)
    return BAProblem(par, sc.camera_indices, sc.image_coords, sc.obj_indices)


def test_fingerprint_works_without_xxhash(monkeypatch):
    prob = _problem()
    with_xx = engine_cache.fingerprint(prob, 0, False)
    monkeypatch.setitem(sys.modules, "xxhash", None)  # `import xxhash` now raises ImportError
    a = engine_cache.fingerprint(prob, 0, False)
    b = engine_cache.fingerprint(prob, 0, False)
    assert a == b and len(a) == 16 and len(with_xx) == 16
    prob.image_coords = prob.image_coords + 1e-9
    assert engine_cache.fingerprint(prob, 0, False) != a
    assert engine_cache.fingerprint(_problem(), 1, False) != a  # another device is another handle


def test_handles_above_the_size_bound_are_not_kept(monkeypatch):
    engine_cache.clear()
    monkeypatch.setenv("CALISCOPE_HIP_ENGINE_CACHE", "1")
    monkeypatch.setenv("CALISCOPE_HIP_ENGINE_CACHE_MAX_GB", "1")
    big, small = _FakeEngine(3 << 30), _FakeEngine(1 << 20)
    engine_cache.checkin(b"k-big", big)
    assert big.closed and not engine_cache._kept
    engine_cache.checkin(b"k-small", small)
    assert not small.closed and list(engine_cache._kept) == [b"k-small"]
    newer = _FakeEngine(1 << 20)
    engine_cache.checkin(b"k-newer", newer)  # capacity 1: the older handle goes
    assert small.closed and list(engine_cache._kept) == [b"k-newer"]
    engine_cache.clear()
    assert newer.closed
