"""The dealt plan of the register Schur kernel (csrc/schur_plan.h) on the CPU: compiled by g++ with a replay harness
(tests/native/plan_harness.cpp) that walks the transposed pair codes the way k_schur_reg3 does.  Every camera-pair block
must receive exactly the pairs of ``sum_p W_p V'^-1 W_p^T`` (the reference forms the same sums
implicitly in ``J^T J``, core/reprojection.py:128-234), and the plan must keep the lanes busy."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
I32P, F64P, I64P = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_long)


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = tmp_path_factory.mktemp("plan") / "libplan_harness.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-pthread", str(ROOT / "tests" / "native" / "plan_harness.cpp"), "-o", str(out)],
                   check=True)
    lib = C.CDLL(str(out))
    lib.plan_replay.restype = C.c_int
    lib.plan_replay.argtypes = [C.c_int] * 17 + [I32P, I32P, F64P, F64P, I64P]
    lib.bind_replay.restype = C.c_int
    lib.bind_replay.argtypes = [C.c_int] * 13 + [C.c_double, C.c_int, C.c_int, I32P, I32P, I32P, I32P, I32P, I32P, I32P, F64P, I32P]
    return lib


def _visibility(rng, n_cams, n_points, k_lo, k_hi, duplicates=0.0, unobserved=0.0):
    cams, starts = [], [0]
    for _ in range(n_points):
        if rng.random() < unobserved:
            starts.append(starts[-1])
            continue
        k = int(rng.integers(k_lo, k_hi + 1))
        c = np.sort(rng.choice(n_cams, size=min(k, n_cams), replace=False))
        if rng.random() < duplicates:  # the same camera sees the point twice (static markers, repeated frames)
            c = np.sort(np.concatenate([c, rng.choice(c, size=1)]))
        cams.append(c)
        starts.append(starts[-1] + len(c))
    return np.concatenate(cams).astype(np.int32), np.asarray(starts, dtype=np.int32)


def _replay(lib, n_cams, hcam, hps, T, nc, *, gmax=16, chunk_cap=None, region_chunks=64, heavy_obs=0, layout="reg3", pair_cap=None, cheap=False):
    P = len(hps) - 1
    G = -(-n_cams // gmax)
    g = -(-n_cams // G)
    threads = 256
    rep = threads // (g * g) if (nc == 6 and g * g <= threads // 2) else 1
    rec = T.shape[1]
    # LDS layout of k_schur_reg3 (Reg3Cfg): 320 slots, 4 waves x 80, 7 pieces apart in runs of 576 (nc = 6); 384 slots, 12 x 32, 11
    # apart in runs of 384 (nc = 9)
    assert layout == "reg3"
    epw, lst, wp, cap = (80, 7, 576, 320) if nc == 6 else (32, 11, 384, 384)
    if chunk_cap is None:
        chunk_cap = cap
    nT = G * (G + 1) // 2
    acc = np.zeros((nT, threads, nc * nc))
    stats = np.zeros(10, dtype=np.int64)
    if pair_cap is None:  # Reg3Cfg::PAIR_CAP
        pair_cap = 0
    rc = lib.plan_replay(n_cams, P, G, g, rep, nc, rec, chunk_cap, epw, lst, wp, region_chunks, heavy_obs, 4, threads // 64, pair_cap, int(cheap), hcam.ctypes.data_as(I32P),
                         hps.ctypes.data_as(I32P), T.ctypes.data_as(F64P), acc.ctypes.data_as(F64P), stats.ctypes.data_as(I64P))
    assert rc == 0, rc
    return acc, stats, (G, g, rep)


def _expected(n_cams, hcam, hps, T, nc, G, g):
    """Direct sums: real blocks (li, lj) of every tile and, for diagonal tiles, the per-camera total of the helper threads."""
    Tm = T[:, : 3 * nc].reshape(-1, nc, 3)
    nT = G * (G + 1) // 2
    real = np.zeros((nT, g * g, nc, nc))
    helper = np.zeros((G, g, nc, nc))
    tile = {}
    t = 0
    for a in range(G):
        for b in range(a, G):
            tile[(a, b)] = t
            t += 1
    for q in range(len(hps) - 1):
        rows = range(hps[q], hps[q + 1])
        for i in rows:
            for j in rows:
                ci, cj = hcam[i], hcam[j]
                blk = Tm[i] @ Tm[j].T
                if ci == cj:
                    helper[ci // g, ci % g] += blk
                elif ci < cj:
                    real[tile[(ci // g, cj // g)], (ci % g) * g + (cj % g)] += blk
    return real, helper


def _check(lib, rng, n_cams, n_points, k_lo, k_hi, nc, **kw):
    vis = {k: kw.pop(k) for k in ("duplicates", "unobserved") if k in kw}
    hcam, hps = _visibility(rng, n_cams, n_points, k_lo, k_hi, **vis)
    rec = 18 if nc == 6 else 30
    T = np.zeros((hps[-1], rec))
    T[:, : 3 * nc] = rng.normal(size=(hps[-1], 3 * nc))
    acc, stats, (G, g, rep) = _replay(lib, n_cams, hcam, hps, T, nc, **kw)
    real, helper = _expected(n_cams, hcam, hps, T, nc, G, g)
    nblk = g * g
    # fold the rep thread slots of every block id
    blocks = np.zeros((acc.shape[0], nblk, nc * nc))
    for s in range(rep):
        blocks += acc[:, s * nblk:(s + 1) * nblk]
    assert np.all(acc[:, rep * nblk:] == 0.0)
    t = 0
    for a in range(G):
        na = min(g, n_cams - a * g)
        for b in range(a, G):
            got = blocks[t].reshape(nblk, nc, nc)
            for li in range(g):
                for lj in range(g):
                    if a == b and lj <= li:
                        continue
                    np.testing.assert_allclose(got[li * g + lj], real[t, li * g + lj], rtol=1e-12, atol=1e-12)
            if a == b:  # helper k (k-th id of the lower triangle) serves camera k mod na (k_reg_fold)
                tot = np.zeros((g, nc, nc))
                k = 0
                for li in range(g):
                    for lj in range(li + 1):
                        tot[k % max(na, 1)] += got[li * g + lj]
                        k += 1
                np.testing.assert_allclose(tot, helper[a], rtol=1e-12, atol=1e-12)
            t += 1
    return stats


def test_pairs_reach_their_blocks_random_visibility(harness):
    rng = np.random.default_rng(5)
    _check(harness, rng, 64, 900, 2, 10, 6)           # four groups of 16, ten tiles
    _check(harness, rng, 8, 300, 2, 8, 6)             # one small group: rep = 4 threads per block
    _check(harness, rng, 20, 400, 1, 6, 6)            # ragged: two groups of 10, single-view points, rep = 2
    _check(harness, rng, 40, 500, 3, 9, 9)            # nine-parameter cameras


def test_duplicate_rows_and_unobserved_points(harness):
    rng = np.random.default_rng(6)
    _check(harness, rng, 32, 600, 2, 7, 6, duplicates=0.2, unobserved=0.1)
    _check(harness, rng, 5, 200, 1, 5, 6, duplicates=0.5, unobserved=0.3)


def test_small_chunks_and_regions(harness):
    rng = np.random.default_rng(7)
    _check(harness, rng, 48, 700, 2, 12, 6, chunk_cap=64, region_chunks=4)   # many regions, caps rise in the leftover passes
    _check(harness, rng, 16, 50, 16, 16, 6, chunk_cap=40)                    # every point fills two fifths of a chunk


def test_cheap_plan_sums_the_same_pairs(harness):
    """Reg2Params::cheap — the plan a handle may start with while the dealt one is being made (CBA_PLAN=swap): one open chunk filled in point
    order, records in arrival order.  Same sums, every pair once; fewer busy lanes and more LDS conflicts are its price."""
    rng = np.random.default_rng(21)
    _check(harness, rng, 64, 900, 2, 10, 6, cheap=True)
    _check(harness, rng, 8, 300, 2, 8, 6, cheap=True)
    _check(harness, rng, 20, 400, 1, 6, 6, cheap=True, duplicates=0.2, unobserved=0.1)
    _check(harness, rng, 40, 500, 3, 9, 9, cheap=True)
    _check(harness, rng, 48, 700, 2, 12, 6, chunk_cap=64, region_chunks=4, cheap=True)
    stats = _check(harness, rng, 64, 10000, 10, 10, 6, cheap=True)
    assert stats[1] == 10000 * 55 and 0.3 < stats[1] / stats[2] < 0.68, stats[1] / stats[2]


def test_lean_cheap_path_builds_the_general_path_s_plan(harness):
    """The cheap plan has a path of its own in build_reg2_plan (run_job_cheap, round 4: one pass, codes in final form, fixed-size lists — 0.6 of the CPU
    time; it is what a two-stage handle waits for in cba_create).  It must produce, bit for bit, the arrays the general path produces with
    prm.cheap: same digest over obs, chunk_start, code_start, nit, codes, tile_chunk_begin."""
    rng = np.random.default_rng(33)
    for n_cams, n_points, k_lo, k_hi, nc, kw in ((64, 2000, 2, 10, 6, {}), (8, 300, 2, 8, 6, {}), (20, 400, 1, 6, 6, dict(duplicates=0.2, unobserved=0.1)),
                                                 (40, 500, 3, 9, 9, {}), (48, 700, 2, 12, 6, dict(chunk_cap=64, region_chunks=4)), (5, 200, 1, 5, 6, dict(duplicates=0.5, unobserved=0.3)),
                                                 (128, 1500, 10, 10, 9, {})):
        vis = dict(duplicates=kw.pop("duplicates", 0.0), unobserved=kw.pop("unobserved", 0.0))
        hcam, hps = _visibility(rng, n_cams, n_points, k_lo, k_hi, **vis)
        T = np.zeros((hps[-1], 18 if nc == 6 else 30))
        _, lean, _ = _replay(harness, n_cams, hcam, hps, T, nc, cheap=1, **kw)
        _, general, _ = _replay(harness, n_cams, hcam, hps, T, nc, cheap=2, **kw)
        assert lean[9] == general[9] and list(lean[:5]) == list(general[:5]), (n_cams, n_points, nc)


def test_heavy_points_are_left_to_their_own_kernel(harness):
    rng = np.random.default_rng(8)
    hcam, hps = _visibility(rng, 16, 200, 2, 6)
    # one static marker: 60 rows
    heavy = np.sort(rng.integers(0, 16, 60)).astype(np.int32)
    hcam = np.concatenate([hcam, heavy]); hps = np.concatenate([hps, [hps[-1] + 60]]).astype(np.int32)
    T = np.zeros((hps[-1], 18)); T[:] = rng.normal(size=T.shape)
    acc, stats, (G, g, rep) = _replay(harness, 16, hcam, hps, T, 6, heavy_obs=40)
    T2 = T.copy(); T2[hps[-2]:] = 0.0   # the heavy point must not contribute
    real, helper = _expected(16, hcam, hps, T2, 6, G, g)
    got = acc[0, :256].reshape(256, 6, 6)
    for li in range(16):
        for lj in range(li + 1, 16):
            np.testing.assert_allclose(got[li * 16 + lj], real[0, li * 16 + lj], rtol=1e-12, atol=1e-12)


def test_lane_utilisation_at_the_bench_shape(harness):
    """64 cameras, every point seen by 10 (cfg4's shape, BASELINE.json configs[3]) at a twentieth of its size: the dealt
    plan keeps >= 70 % of the lane-iterations busy (round 1's greedy window: 48 %)."""
    rng = np.random.default_rng(9)
    stats = _check(harness, rng, 64, 10000, 10, 10, 6)
    n_pairs, lane_iters = stats[1], stats[2]
    assert n_pairs == 10000 * 55
    assert n_pairs / lane_iters > 0.68, n_pairs / lane_iters
    # LDS bank conflicts of the record reads (ds_read_b128: four groups of 16 lanes, one cycle per group when the 16 records sit
    # in 16 different bank groups): the coloured slots stay below 1.6 cycles per group, the arrival order needs ~2.5
    groups, cycles, arrival = stats[5], stats[6], stats[7]
    assert cycles / groups < 1.7 and arrival / groups > 2.2, (cycles / groups, arrival / groups)


@pytest.mark.parametrize("shape", [dict(n_cams=64, n_points=20000, k=10, max_blocks=512), dict(n_cams=64, n_points=20000, k=10, max_blocks=512, cost_a=-1.0),
                                   dict(n_cams=40, n_points=3000, k=8, max_blocks=512), dict(n_cams=128, n_points=6000, k=10, max_blocks=256, nc=9),
                                   dict(n_cams=8, n_points=400, k=6, max_blocks=512), dict(n_cams=64, n_points=20000, k=10, max_blocks=512, fine=True)])
def test_workgroup_binding_covers_every_chunk_once(harness, shape, monkeypatch):
    """csrc/wg_binding.h: the persistent workgroups of the pair kernel walk (first, first + stride, ... < end); every chunk of every tile must be
    walked by exactly one workgroup of that tile, and the workgroups are handed out in proportion to the tiles' estimated cost."""
    cfg = dict(shape)
    nc, layout, cost_a, max_blocks = cfg.pop("nc", 6), cfg.pop("layout", "reg3"), cfg.pop("cost_a", 2.0), cfg.pop("max_blocks")
    if cfg.pop("fine", False):  # CBA_BIND=fine: counts per tile to one workgroup, cost-proportional XCD slices (an option of the library)
        monkeypatch.setenv("PLAN_BIND_FINE", "1")
    rng = np.random.default_rng(12)
    hcam, hps = _visibility(rng, cfg["n_cams"], cfg["n_points"], cfg["k"], cfg["k"])
    gmax = 16
    G = -(-cfg["n_cams"] // gmax)
    g = -(-cfg["n_cams"] // G)
    threads = 256
    rep = threads // (g * g) if (nc == 6 and g * g <= threads // 2) else 1
    epw, lst, wp, cap = (80, 7, 576, 320) if nc == 6 else (32, 11, 384, 384)
    n_waves, phys = 4, 4
    nT = G * (G + 1) // 2
    cap_wg = max_blocks + nT
    wt, wf, we, ws = (np.zeros(cap_wg, dtype=np.int32) for _ in range(4))
    tcb, cost, xcd = np.zeros(nT + 1, dtype=np.int32), np.zeros(nT), np.zeros(1, dtype=np.int32)
    grid = harness.bind_replay(cfg["n_cams"], len(hps) - 1, G, g, rep, cap, epw, lst, wp, 32, n_waves, 0, phys, cost_a, max_blocks, 1,
                               hcam.ctypes.data_as(I32P), hps.ctypes.data_as(I32P), *(a.ctypes.data_as(I32P) for a in (wt, wf, we, ws, tcb)),
                               cost.ctypes.data_as(F64P), xcd.ctypes.data_as(I32P))
    assert 0 < grid <= cap_wg
    n_chunks = int(tcb[nT])
    seen = np.zeros(n_chunks, dtype=np.int32)
    per_tile = np.zeros(nT, dtype=np.int64)
    for b in range(grid):
        t = wt[b]
        per_tile[t] += 1
        walk = np.arange(wf[b], we[b], ws[b])
        assert ws[b] >= 1 and tcb[t] <= wf[b] and we[b] <= tcb[t + 1]
        seen[walk] += 1
    assert np.all(seen == 1), (int((seen == 0).sum()), int((seen > 1).sum()))
    assert np.all(per_tile >= 1) and np.all(np.diff(wt[:grid]) >= 0)  # tiles own contiguous runs of workgroups (k_reg_reduce relies on it)
    if xcd[0]:  # workgroup b sits on XCD b mod 8; the workgroups of a tile on one XCD share one slice of its chunk range, and the slices follow the XCD order
        assert grid % 8 == 0
        for t in range(nT):
            ids = np.flatnonzero(wt[:grid] == t)
            ends = []
            for x in range(8):
                on_x = ids[ids % 8 == x]
                if len(on_x):
                    assert len(set(we[on_x])) == 1 and np.all(ws[on_x] == len(on_x)) and sorted(wf[on_x] - wf[on_x].min()) == list(range(len(on_x)))
                    ends.append((x, int(wf[on_x].min()), int(we[on_x][0])))
            assert all(a[2] == b[1] for a, b in zip(ends, ends[1:])) and ends[0][1] == tcb[t] and ends[-1][2] == tcb[t + 1]
            assert per_tile[t] < 8 or np.ptp([len(ids[ids % 8 == x]) for x in range(8)]) <= 1
    if grid >= 4 * nT:  # enough workgroups to balance: cost per workgroup within a quarter of the mean
        weight = cost if cost_a >= 0 else np.diff(tcb).astype(float)
        load = weight / per_tile
        assert load.max() <= 1.3 * load.mean(), (load, per_tile)


def test_workgroup_renumbering_is_a_permutation_that_keeps_the_xcd(harness):
    """csrc/wg_binding.h logical_workgroup: what k_schur_reg3 uses as its workgroup id."""
    for grid in (16, 256, 512, 288, 40, 7):
        ids = np.array([harness.logical_workgroup_of(b, grid) for b in range(grid)])
        assert sorted(ids) == list(range(grid))
        if grid % 16 == 0:
            assert np.all(ids % 8 == np.arange(grid) % 8)
            first_half = ids[: grid // 2] // 8
            assert np.all(first_half % 2 == 0) and np.all((ids[grid // 2:] // 8) % 2 == 1)  # groups of eight alternate between the halves
        else:
            assert np.all(ids == np.arange(grid))
