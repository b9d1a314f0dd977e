"""The C-ABI library loads without a GPU and exports every symbol include/caliscope_ba.h declares;
host-side planning (no device needed) is checked against numpy."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

from caliscope_amd import _lib, build
from caliscope_amd.exceptions import BackendError

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)
    return _lib.load()


def test_header_symbols_are_exported_and_bound(lib):
    header = (ROOT / "include" / "caliscope_ba.h").read_text()
    declared = set(re.findall(r"\b(cba_[a-z_0-9]+)\s*\(", header))
    declared -= {"cba_problem"}
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.cba_version() == 100
    assert lib.cba_timer_count() == 13 and lib.cba_timer_name(2) == b"build" and lib.cba_timer_name(12) == b"exchange"


def test_create_fails_loudly_without_device(lib):
    if lib.cba_device_count() > 0:
        pytest.skip("a HIP device is present")
    from caliscope_amd.engine import BAProblem
    from caliscope_amd.hip_engine import HipEngine
    from tests.helpers import small_problem

    sc, par, _ = small_problem(n_cams=3, n_points=10, k=3)
    with pytest.raises(BackendError, match="no HIP device|NO_DEVICE|no CPU fallback"):
        HipEngine(BAProblem(par, sc.camera_indices, sc.image_coords, sc.obj_indices))


def _plan(lib, n_points, obs_pt, cap, obs_cam=None, n_cams=0):
    obs_pt = np.ascontiguousarray(obs_pt, dtype=np.int32)
    n = len(obs_pt)
    cam_p = None
    if obs_cam is not None:
        obs_cam = np.ascontiguousarray(obs_cam, dtype=np.int32)
        cam_p = obs_cam.ctypes.data_as(_lib.c_int32_p)
    order = np.zeros(max(n, 1), dtype=np.int64)
    pstart = np.zeros(n_points + 1, dtype=np.int64)
    cstart = np.zeros(n + 2, dtype=np.int64)
    nch = lib.cba_host_plan(n_points, n, obs_pt.ctypes.data_as(_lib.c_int32_p), cam_p, n_cams, cap,
                            order.ctypes.data_as(_lib.c_int64_p), pstart.ctypes.data_as(_lib.c_int64_p),
                            cstart.ctypes.data_as(_lib.c_int64_p))
    return nch, order[:n], pstart, cstart[: max(nch, 0) + 1]


def test_host_plan_sorts_by_point_and_chunks_whole_points(lib):
    rng = np.random.default_rng(0)
    n_points = 500
    counts = rng.integers(0, 12, n_points)  # ragged, including points without observations
    obs_pt = np.repeat(np.arange(n_points), counts)
    rng.shuffle(obs_pt)
    nch, order, pstart, cstart = _plan(lib, n_points, obs_pt, 256)
    assert nch > 0
    assert np.array_equal(order, np.argsort(obs_pt, kind="stable"))
    assert np.array_equal(pstart, np.concatenate([[0], np.cumsum(counts)]))
    assert cstart[0] == 0 and cstart[-1] == len(obs_pt)
    sizes = np.diff(cstart)
    assert sizes.min() > 0 and sizes.max() <= 256
    assert np.all(np.isin(cstart, pstart)), "chunk boundaries must coincide with point boundaries"
    # with cameras: sorted by (point, camera), ties in input order
    obs_cam = rng.integers(0, 7, len(obs_pt))
    _, order2, pstart2, _ = _plan(lib, n_points, obs_pt, 256, obs_cam, 7)
    assert np.array_equal(order2, np.lexsort((np.arange(len(obs_pt)), obs_cam, obs_pt)))
    assert np.array_equal(pstart2, pstart)
    # greedy packing: a chunk plus the next point would overflow
    sorted_pts = obs_pt[order]
    for c in range(nch - 1):
        nxt = sorted_pts[cstart[c + 1]]
        assert sizes[c] + counts[nxt] > 256


def test_host_plan_threaded_passes_on_sorted_input(lib):
    """Above 2 x 131072 observations the range / sortedness pass and, for input already in (point, camera) order, the point table and the
    identity order run on several threads (slices of the observation range): same answers as the sort."""
    rng = np.random.default_rng(1)
    n_points = 90_000
    counts = rng.integers(0, 14, n_points)  # ragged, unobserved points at both ends of slices
    counts[:3] = 0
    counts[-2:] = 0
    obs_pt = np.repeat(np.arange(n_points), counts)
    assert len(obs_pt) > 4 * 131072
    obs_cam = np.concatenate([np.sort(rng.choice(40, c, replace=False)) for c in counts]).astype(np.int32)
    nch, order, pstart, cstart = _plan(lib, n_points, obs_pt, 256, obs_cam, 40)
    assert nch > 0 and np.array_equal(order, np.arange(len(obs_pt)))
    assert np.array_equal(pstart, np.concatenate([[0], np.cumsum(counts)]))
    assert cstart[0] == 0 and cstart[-1] == len(obs_pt) and np.diff(cstart).max() <= 256 and np.all(np.isin(cstart, pstart))
    # one pair of neighbours out of camera order, deep inside a later slice: the sort path, same tables
    k = int(np.flatnonzero(np.diff(obs_pt) == 0)[-5])
    cam2 = obs_cam.copy()
    cam2[k], cam2[k + 1] = cam2[k + 1], cam2[k]
    nch2, order2, pstart2, cstart2 = _plan(lib, n_points, obs_pt, 256, cam2, 40)
    assert nch2 == nch and np.array_equal(pstart2, pstart) and np.array_equal(cstart2, cstart)
    expect = np.arange(len(obs_pt)); expect[k], expect[k + 1] = k + 1, k
    assert np.array_equal(order2, expect)
    # a bad index in the last slice is reported with its position
    bad = obs_cam.copy()
    bad[-7] = 40
    nch3, *_ = _plan(lib, n_points, obs_pt, 256, bad, 40)
    assert nch3 == -1 and f"observation {len(obs_pt) - 7}: camera index 40".encode() in lib.cba_last_error()


def test_host_plan_rejects_bad_input(lib):
    nch, *_ = _plan(lib, 4, np.array([0, 1, 7]), 256)
    assert nch == -1 and b"out of range" in lib.cba_last_error()
    # a point larger than a chunk gets chunks of its own
    obs_pt = np.concatenate([np.zeros(5), np.ones(700), np.full(3, 2), np.full(256, 3), np.full(257, 4)]).astype(np.int32)
    nch, order, pstart, cstart = _plan(lib, 5, obs_pt, 256)
    assert list(cstart) == [0, 5, 261, 517, 705, 708, 964, 1220, 1221] and nch == 8
    nch, order, pstart, cstart = _plan(lib, 3, np.array([], dtype=np.int32), 256)
    assert nch == 0


def test_no_kernel_asks_the_runtime_for_a_stack(lib, tmp_path):
    """Private (scratch) memory of the kernels, read from the code object inside the built library.  A kernel with a DYNAMIC stack (a recursive
    device function) makes the HIP runtime reserve hipLimitStackSize per lane for every wave slot of the device — 512 MB of HBM on an MI355X, taken
    inside the kernel's first launch and kept for the life of the process (tools/device_memory_probe.py; round 5 found the one-workgroup step kernel
    doing that through trf::real_roots calling itself).  Fixed-size scratch is tolerated only where it is known: the fixed-order (deterministic)
    build of nine-parameter cameras, which parks six rounds of running sums per thread."""
    import shutil
    import subprocess

    llvm = Path("/opt/rocm/lib/llvm/bin")
    bundler, readelf, objcopy = llvm / "clang-offload-bundler", llvm / "llvm-readelf", shutil.which("objcopy") or str(llvm / "llvm-objcopy")
    if not (bundler.exists() and readelf.exists() and Path(objcopy).exists()):
        pytest.skip("LLVM binary utilities of the ROCm image not found")
    fat, co = tmp_path / "fatbin.bin", tmp_path / "gfx950.co"
    subprocess.run([objcopy, "-O", "binary", "--only-section=.hip_fatbin", str(build.OUT), str(fat)], check=True)
    subprocess.run([str(bundler), "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    notes = subprocess.run([str(readelf), "--notes", str(co)], check=True, capture_output=True, text=True).stdout
    kernels = re.findall(r"\.name:\s+(\S+)\s+\.private_segment_fixed_size:\s+(\d+)", notes)  # (.name directly precedes the size in the metadata map)
    assert len(kernels) > 100, len(kernels)
    assert notes.count(".uses_dynamic_stack: false") == len(kernels) and ".uses_dynamic_stack: true" not in notes
    with_scratch = {name: int(size) for name, size in kernels if int(size)}
    assert all(name.startswith("_ZN3cba7k_buildILi9ELi") for name in with_scratch), with_scratch
