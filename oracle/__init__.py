"""CPU oracle for the bundle-adjustment hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain numpy restatement of the arithmetic on the reference's
hot path (``/root/reference/src/caliscope/core/reprojection.py``,
``core/bundle_parameterization.py``, ``core/capture_volume.py:322-444``) plus
the three OpenCV functions that path calls (``cv2.Rodrigues``,
``cv2.projectPoints``, ``cv2.fisheye.projectPoints``).  OpenCV
(``opencv-python`` 5.0.0.93, ``uv.lock:1649``) is a third-party dependency that
is NOT under ``/root/reference`` and is not installed here, so its published
camera model is restated from the formulas.  SciPy 1.15.3
(the pinned solver, ``uv.lock:2404``) IS installed and is used as-is.

Pinning (see tests/test_oracle_*.py and DESIGN.md):
  * ``cv2.projectPoints`` semantics are pinned to 1e-10 px by reproducing the
    reference's golden vector ``tests/fixtures/synthetic/default_ring_baseline/
    image_points_noisy.csv`` (2800 rows) through :mod:`oracle.scene`.
  * the analytic Jacobian is pinned by the reference's own known-answer test
    recipe (central finite differences, 1e-6 per column,
    ``tests/synthetic/test_analytic_jacobian.py:37-50``).
  * real-data convention check: the calibrated ``post_optimization`` session
    reprojects at sub-pixel RMSE.
  * ``cv2.fisheye.projectPoints`` and ``cv2.Rodrigues`` matrix->vector near pi
    have no fixture in the reference ("parity unpinned" for those two; they are
    validated by finite differences and round-trip identities only).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  Nothing under ``caliscope_amd/`` does.
"""
