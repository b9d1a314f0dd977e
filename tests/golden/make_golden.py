"""Provenance of tests/golden/: copies of the reference's OWN data fixtures (not source code).

Run in the build container (where /root/reference is mounted):  python tests/golden/make_golden.py

  default_ring_baseline/image_points_noisy.csv
      <- /root/reference/tests/fixtures/synthetic/default_ring_baseline/image_points_noisy.csv
      the golden vector produced by the real cv2.projectPoints + numpy default_rng(42)
      (pinned by reference tests/synthetic/primitives/test_scene.py:641-664, atol 1e-10).
  post_optimization/{camera_array.toml,xy_CHARUCO.csv,xyz_CHARUCO.csv}
      <- /root/reference/tests/sessions/post_optimization/...
      a real calibrated 4-camera session (BASELINE.json configs[0]); used by the reference in
      tests/test_reprojection_report.py and tests/test_capture_volume.py.
  reference_host/*.npz  <- tests/golden/make_reference_host_fixtures.py: random inputs and what the REFERENCE'S OWN host code returned for them in the
      build container (eleven families: tables, parameterization, constraint compilers and rows, filters, report bookkeeping, on-disk formats,
      the optimize() seam, the stage driver, triangulation); its docstring says what was stubbed (cv2 / rtoml imports) and why that does not
      touch the code under test.  Regenerating reproduces the committed files byte for byte.
  scipy_refs/*.npz  <- tests/golden/make_scipy_refs.py: solutions of the reference's scipy call (oracle callables) at BASELINE sizes, computed on the
      CPU of the build container (minutes to an hour and a half each); consumers check the stored x0 digest against their own x0.
"""
import shutil
from pathlib import Path

REF = Path("/root/reference/tests")
HERE = Path(__file__).parent

COPIES = {
    REF / "fixtures/synthetic/default_ring_baseline/image_points_noisy.csv": HERE / "default_ring_baseline/image_points_noisy.csv",
    REF / "sessions/post_optimization/camera_array.toml": HERE / "post_optimization/camera_array.toml",
    REF / "sessions/post_optimization/calibration/extrinsic/CHARUCO/xy_CHARUCO.csv": HERE / "post_optimization/xy_CHARUCO.csv",
    REF / "sessions/post_optimization/calibration/extrinsic/CHARUCO/xyz_CHARUCO.csv": HERE / "post_optimization/xyz_CHARUCO.csv",
}

if __name__ == "__main__":
    for src, dst in COPIES.items():
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copyfile(src, dst)
        print(f"{src} -> {dst} ({dst.stat().st_size} bytes)")
