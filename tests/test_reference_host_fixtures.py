"""The host logic either side of the solve against OUTPUTS OF THE REFERENCE ITSELF: tests/golden/reference_host/case_*.npz hold random tables and
what the reference's own ``CaptureVolume`` made of them in the build container (tests/golden/make_reference_host_fixtures.py runs its code
unmodified, cv2 / rtoml stubbed because nothing on this path calls them) — the observation -> world-point map (core/capture_volume.py:119-139),
the constraint rows (:446-516), the rigidity report (:532-605), the sync-index range.  Duplicate world keys, static objects, holes, observations
without a point and constraints that cannot fire are all in the cases."""
import warnings
from pathlib import Path

import numpy as np
import pandas as pd
import pytest

from caliscope_amd.cameras import CameraArray, CameraData
from caliscope_amd.capture_volume import CaptureVolume
from caliscope_amd.constraints import CentroidDistanceConstraint, ConstraintSet, DistanceConstraint
from caliscope_amd.point_data import ImagePoints, WorldPoints

CASES = sorted((Path(__file__).parent / "golden" / "reference_host").glob("case_*.npz"))
WORLD_COLS = ["sync_index", "object_id", "keypoint_id", "x_coord", "y_coord", "z_coord", "frame_time"]
IMG_COLS = ["sync_index", "cam_id", "object_id", "keypoint_id", "img_loc_x", "img_loc_y"]


def _volume(ref):
    wdf = pd.DataFrame(ref["world"], columns=WORLD_COLS).astype({"sync_index": "int64", "object_id": "int64", "keypoint_id": "int64"})
    idf = pd.DataFrame(ref["image"], columns=IMG_COLS).astype({c: "int64" for c in IMG_COLS[:4]})
    K = np.array([[400.0, 0.0, 200.0], [0.0, 400.0, 200.0], [0.0, 0.0, 1.0]])
    cams = CameraArray({c: CameraData(cam_id=c, size=(400, 400), matrix=K.copy(), distortions=np.zeros(5), rotation=np.eye(3),
                                      translation=np.array([0.1 * c, 0.0, 0.0])) for c in (0, 1)})
    dist = tuple(DistanceConstraint(int(a), int(b), int(c), int(d), float(e), float(f)) for a, b, c, d, e, f in ref["distances"])
    cent = tuple(CentroidDistanceConstraint(int(a), int(b), float(c), float(d)) for a, b, c, d in ref["centroids"])
    cs = ConstraintSet(dist, frozenset(int(o) for o in ref["static_ids"]), centroid_distances=cent)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # (duplicate keys and thin geometry are warned about, as in the reference)
        return CaptureVolume(cams, ImagePoints(idf), WorldPoints(wdf), cs)


def _sorted_rows(*columns):
    n = len(columns[0])
    if n == 0:
        return np.zeros((0, 0))
    rows = np.column_stack([np.asarray(c, dtype=np.float64).reshape(n, -1) for c in columns])
    return rows[np.lexsort(rows.T[::-1])]


def test_the_fixtures_are_there():
    assert len(CASES) == 10


@pytest.mark.parametrize("path", CASES, ids=lambda p: p.stem)
def test_host_logic_equals_the_reference_s_own_output(path):
    ref = np.load(path)
    vol = _volume(ref)
    # the tables keep the caller's row order (the map and the constraint rows are row numbers)
    assert np.array_equal(vol.world_points.df[WORLD_COLS[:3]].to_numpy(), ref["world"][:, :3].astype(np.int64))
    assert np.array_equal(vol.img_to_obj_map, ref["img_to_obj_map"])
    arrays = vol._build_constraint_arrays()
    assert (arrays is not None) == bool(ref["has_rows"])
    if arrays is not None:
        ga, gb, dist, sig = arrays
        assert ga.dtype == gb.dtype == np.int32 and ga.shape == gb.shape == ref["groups_a"].shape
        # the reference walks a SET of shared sync indices: the order of a constraint's rows is not defined there; the rows are
        assert np.array_equal(_sorted_rows(ga, gb, dist, sig), _sorted_rows(ref["groups_a"], ref["groups_b"], ref["row_distance"], ref["row_sigma"]))
    rep = vol.rigidity_report()
    mine = np.array([[v.object_id_a, v.keypoint_id_a, v.object_id_b, v.keypoint_id_b, v.sync_index, 1 if v.kind == "centroid" else 0] for v in rep.violations],
                    dtype=np.int64).reshape(-1, 6)
    got = _sorted_rows(mine, [v.expected for v in rep.violations], [v.actual for v in rep.violations])
    want = _sorted_rows(ref["violations"], ref["violation_expected"], ref["violation_actual"])
    assert got.shape == want.shape
    if len(got):
        assert np.array_equal(got[:, :7], want[:, :7]) and np.allclose(got[:, 7], want[:, 7], rtol=0, atol=1e-12)
    assert rep.rmse_mm == pytest.approx(float(ref["rmse_mm"]), abs=1e-9) and rep.max_violation_mm == pytest.approx(float(ref["max_violation_mm"]), abs=1e-9)
    assert np.array_equal(vol.unique_sync_indices, ref["unique_sync_indices"])
    assert (vol.world_points.min_index, vol.world_points.max_index) == tuple(int(v) for v in ref["world_min_max"])


# ---- the parameterization (core/bundle_parameterization.py) on random camera arrays -----------------------------------------------------
BUNDLES = sorted((Path(__file__).parent / "golden" / "reference_host").glob("bundle_*.npz"))


def _camera_array(ref):
    from caliscope_amd.cameras import rvec_to_matrix

    cams = {}
    for i, cid in enumerate(ref["cam_ids"]):
        fisheye = bool(ref["fisheye"][i])
        dist = ref["dist"][i][: 4 if fisheye else 5]
        posed = bool(ref["posed"][i])
        cams[int(cid)] = CameraData(cam_id=int(cid), size=(int(ref["sizes"][i][0]), int(ref["sizes"][i][1])), matrix=ref["K"][i].copy(), distortions=dist.copy(),
                                    fisheye=fisheye, ignore=bool(ref["ignore"][i]), rotation=rvec_to_matrix(ref["rvec"][i]) if posed else None,
                                    translation=ref["t"][i].copy() if posed else None)
    return CameraArray(cams)


def test_the_parameterization_fixtures_are_there():
    assert len(BUNDLES) == 8


@pytest.mark.parametrize("path", BUNDLES, ids=lambda p: p.stem)
def test_parameterization_equals_the_reference_s_own_output(path):
    from caliscope_amd.bundle_parameterization import BundleParameterization

    ref = np.load(path)
    arr = _camera_array(ref)
    par = BundleParameterization.from_camera_array(arr, int(ref["n_points"]), refine_intrinsics=bool(ref["refine"]))
    blocks = np.array([[b.cam_id, int(b.free_intrinsics), b.fx_initial, b.fy_initial, b.cx, b.cy, int(b.fisheye), b.k1_initial, b.k2_initial, len(b.dist_fixed),
                        *(list(b.dist_fixed) + [0.0] * (4 - len(b.dist_fixed)))] for b in par.blocks], dtype=np.float64)
    assert np.array_equal(blocks, ref["blocks"])  # which cameras, in which order, what is free, every constant — exactly
    assert np.array_equal(par.camera_param_offsets, ref["offsets"]) and par.n_camera_params == int(ref["n_camera_params"])
    x0 = par.pack(arr, ref["points"])
    assert x0.shape == ref["x0"].shape
    rot = np.zeros(x0.size, dtype=bool)  # entries that went through a rotation-vector conversion (scipy's on the reference's side: docstring of the generator)
    for off in par.camera_param_offsets:
        rot[off:off + 3] = True
    assert np.array_equal(x0[~rot], ref["x0"][~rot]) and np.allclose(x0[rot], ref["x0"][rot], rtol=0, atol=1e-12)
    lb, ub = par.bounds()
    assert np.array_equal(lb, ref["lb"]) and np.array_equal(ub, ref["ub"])
    x1 = ref["x1"]
    got = sorted((w.cam_id, {"f": 0, "k1": 1, "k2": 2}[w.parameter], {"lower": 0, "upper": 1}[w.bound], w.value) for w in par.bound_warnings(x1))
    want = sorted(tuple(r) for r in ref["warnings"].tolist())
    assert len(got) == len(want) and all(g[:3] == tuple(int(v) for v in w[:3]) and g[3] == w[3] for g, w in zip(got, want))
    arr2 = _camera_array(ref)
    pts_back = par.unpack_into(arr2, x1.copy())
    assert np.array_equal(pts_back, ref["points_back"])
    for k, b in enumerate(par.blocks):
        cam = arr2.cameras[b.cam_id]
        assert np.allclose(cam.rotation, ref["unpacked_R"][k], rtol=0, atol=1e-12) and np.array_equal(np.ravel(cam.translation), ref["unpacked_t"][k])
        assert np.array_equal(cam.matrix, ref["unpacked_K"][k])
        d = np.ravel(cam.distortions)
        assert np.array_equal(d, ref["unpacked_dist"][k][: d.size]) and np.all(np.isnan(ref["unpacked_dist"][k][d.size:]))
    est = np.array([[e.cam_id, e.f_recovered, e.k1_recovered, e.k2_recovered, e.f_initial, e.k1_initial, e.k2_initial] for e in par.intrinsic_estimates(arr2)],
                   dtype=np.float64).reshape(-1, 7)
    assert np.array_equal(est, ref["estimates"])
    for k in range(len(par.blocks)):
        rvec, tvec, K, dist = par.trial_projection_inputs(x1, k)
        assert np.array_equal(rvec, ref["trial_rvec"][k]) and np.array_equal(tvec, ref["trial_tvec"][k]) and np.array_equal(K, ref["trial_K"][k])
        assert np.array_equal(np.ravel(dist), ref["trial_dist"][k][: np.size(dist)]) and np.all(np.isnan(ref["trial_dist"][k][np.size(dist):]))
    # the tables the C ABI receives: the generator flattened the REFERENCE'S OWN parameterization object with this package's device_tables (the
    # one-line patch passes the reference's class through the duck-typed seam); this package's class must flatten to the same
    from caliscope_amd.bundle_parameterization import device_tables, n_params_of

    tabs = device_tables(par)
    assert n_params_of(par) == int(ref["n_params_of"]) and {f"device_{k}" for k in tabs} == {k for k in ref.files if k.startswith("device_")}
    for k, v in tabs.items():
        assert np.array_equal(np.asarray(v), ref[f"device_{k}"]), k
    sp = par.sparsity(ref["cam_idx"], ref["obj_idx"], 4, ref["groups_a"], ref["groups_b"]).tocoo()
    assert tuple(sp.shape) == tuple(int(v) for v in ref["sparsity_shape"])
    mine = set(zip(sp.row[sp.data != 0].tolist(), sp.col[sp.data != 0].tolist()))
    assert mine == set(zip(ref["sparsity_rows"].tolist(), ref["sparsity_cols"].tolist()))


# ---- the point tables (core/point_data.py) on random tracks with holes --------------------------------------------------------------------
TABLES = sorted((Path(__file__).parent / "golden" / "reference_host").glob("tables_*.npz"))


def _same_table(df, want, columns):
    """Same columns in the same order, same rows in the same order; NaN equals NaN."""
    assert list(df.columns) == list(columns), (list(df.columns), list(columns))
    got = df.to_numpy(dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.allclose(got, want, rtol=0, atol=1e-12, equal_nan=True), float(np.nanmax(np.abs(got - want)))


def test_the_table_fixtures_are_there():
    assert len(TABLES) == 6


@pytest.mark.parametrize("path", TABLES, ids=lambda p: p.stem)
def test_point_tables_equal_the_reference_s_own_output(path, tmp_path):
    ref = np.load(path)
    icols, wcols = [str(c) for c in ref["image_columns"]], [str(c) for c in ref["world_columns"]]
    idf = pd.DataFrame(ref["image"], columns=IMG_COLS).astype({c: "int64" for c in IMG_COLS[:4]})
    wdf = pd.DataFrame(ref["world"], columns=WORLD_COLS).astype({"sync_index": "int64", "object_id": "int64", "keypoint_id": "int64"})
    ip, wp = ImagePoints(idf), WorldPoints(wdf)
    _same_table(ip.df, ref["image_validated"], icols)  # optional columns added, column order
    _same_table(wp.df, ref["world_validated"], wcols)
    for gap in (1, 3, 5):
        for mine, key, cols in ((ip.fill_gaps(gap).df, f"image_filled_{gap}", icols), (wp.fill_gaps(gap).df, f"world_filled_{gap}", wcols)):
            theirs = [str(c) for c in ref[key + "_columns"]]
            assert set(theirs) - set(cols) <= {"gap_size"}  # the reference's filled frame drags its helper column along; the schema columns are compared
            _same_table(mine, ref[key][:, [theirs.index(c) for c in cols]], cols)
    _same_table(wp.smooth(fps=30.0, cutoff_freq=6.0, order=2).df, ref["world_smoothed"], wcols)
    _same_table(wp.smooth(fps=60.0, cutoff_freq=4.0, order=3).df, ref["world_smoothed_o3"], wcols)
    _same_table(ip.filter_to_objects([int(o) for o in ref["kept_objects"]]).df, ref["image_filtered"], icols)
    # on-disk format: what the reference wrote is read back to the same table, and what this package writes is what the reference wrote
    (tmp_path / "ref_xy.csv").write_text(str(ref["image_csv"]))
    (tmp_path / "ref_xyz.csv").write_text(str(ref["world_csv"]))
    _same_table(ImagePoints.from_csv(tmp_path / "ref_xy.csv").df, ref["image_csv_back"], icols)
    _same_table(WorldPoints.from_csv(tmp_path / "ref_xyz.csv").df, ref["world_csv_back"], wcols)
    ip.to_csv(tmp_path / "xy.csv")
    wp.to_csv(tmp_path / "xyz.csv")
    assert (tmp_path / "xy.csv").read_text() == str(ref["image_csv"])
    assert (tmp_path / "xyz.csv").read_text() == str(ref["world_csv"])


# ---- on-disk formats: directories written by this package, read by the reference's CaptureVolume.load() ---------------------------------------
INTEROP = sorted((Path(__file__).parent / "golden" / "reference_host").glob("interop_*.npz"))


def test_the_interop_fixtures_are_there():
    assert len(INTEROP) == 6


@pytest.mark.parametrize("path", INTEROP, ids=lambda p: p.stem)
def test_saved_volumes_read_as_the_reference_read_them(path, tmp_path):
    """The fixture's files were written by ``CaptureVolume.save()`` of this package and loaded by the reference's ``CaptureVolume.load()``; here the
    same files go through this package's loader, which must return what the reference's returned, field by field — and saving the loaded volume
    again must reproduce the files byte for byte."""
    ref = np.load(path)
    src = tmp_path / "saved"
    src.mkdir()
    for name, text in zip(ref["file_names"], ref["file_texts"]):
        (src / str(name)).write_text(str(text))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vol = CaptureVolume.load(src)
    ids = sorted(vol.camera_array.cameras)
    assert ids == ref["cam_ids"].tolist()

    def opt(v):
        return np.nan if v is None else float(v)

    for k, c in enumerate(ids):
        cam = vol.camera_array.cameras[c]
        assert tuple(cam.size) == tuple(ref["sizes"][k]) and cam.rotation_count == ref["rotation_count"][k]
        for mine, theirs in ((opt(cam.error), ref["error"][k]), (opt(cam.exposure), ref["exposure"][k]), (opt(cam.grid_count), ref["grid_count"][k])):
            assert (np.isnan(mine) and np.isnan(theirs)) or mine == theirs
        assert bool(cam.ignore) == bool(ref["ignore"][k]) and bool(cam.fisheye) == bool(ref["fisheye"][k])
        assert np.array_equal(cam.matrix, ref["K"][k])
        d = np.ravel(cam.distortions)
        assert np.array_equal(d, ref["dist"][k][: d.size]) and np.all(np.isnan(ref["dist"][k][d.size:]))
        assert (cam.rotation is not None) == bool(ref["posed"][k])
        if cam.rotation is not None:
            assert np.allclose(cam.rotation, ref["R"][k], rtol=0, atol=1e-12) and np.array_equal(np.ravel(cam.translation), ref["t"][k])
    _same_table(vol.image_points.df, ref["image"], [str(c) for c in ref["image_columns"]])
    _same_table(vol.world_points.df, ref["world"], [str(c) for c in ref["world_columns"]])
    assert np.array_equal(vol.img_to_obj_map, ref["img_to_obj_map"])
    assert (vol.constraints is not None) == bool(ref["has_constraints"])
    if vol.constraints is not None:
        cs = vol.constraints
        mine = np.array([[d.object_id_a, d.keypoint_id_a, d.object_id_b, d.keypoint_id_b, d.distance, d.sigma] for d in cs.distances], dtype=np.float64).reshape(-1, 6)
        assert np.array_equal(mine, ref["distances"])
        mine = np.array([[c.object_id_a, c.object_id_b, c.distance, c.sigma] for c in cs.centroid_distances], dtype=np.float64).reshape(-1, 4)
        assert np.array_equal(mine, ref["centroids"])
        assert sorted(cs.static_object_ids) == ref["static_ids"].tolist()
        mine = np.array([[r.object_id_from, r.keypoint_id_from, r.object_id_to, r.keypoint_id_to, r.obj_loc_x, r.obj_loc_y, r.obj_loc_z] for r in cs.point_remaps],
                        dtype=np.float64).reshape(-1, 7)
        assert np.array_equal(mine, ref["remaps"])
        assert (cs.back_face_thickness_m is None and np.isnan(ref["thickness"])) or cs.back_face_thickness_m == float(ref["thickness"])
    again = tmp_path / "again"
    vol.save(again)
    assert sorted(f.name for f in again.iterdir()) == sorted(str(n) for n in ref["file_names"])
    for name, text in zip(ref["file_names"], ref["file_texts"]):
        if str(name) == "camera_array.toml":  # (rotations go matrix -> vector -> matrix -> vector: equal to rounding, compared as numbers below)
            continue
        assert (again / str(name)).read_text() == str(text), str(name)
    import tomli

    a, b = tomli.loads((again / "camera_array.toml").read_text()), tomli.loads(str(ref["file_texts"][list(ref["file_names"]).index("camera_array.toml")]))
    assert a.keys() == b.keys() and a["cameras"].keys() == b["cameras"].keys()
    for cid, entry in b["cameras"].items():
        assert a["cameras"][cid].keys() == entry.keys()
        for key, value in entry.items():
            if isinstance(value, list):
                assert np.allclose(np.array(a["cameras"][cid][key], dtype=np.float64), np.array(value, dtype=np.float64), rtol=0, atol=1e-12), (cid, key)
            else:
                assert a["cameras"][cid][key] == value, (cid, key)


# ---- the constraint compilers (core/constraints.py:84-190 from_marker_set, :397-418 from_chessboard) -------------------------------------------
COMPILERS = sorted((Path(__file__).parent / "golden" / "reference_host").glob("compilers_*.npz"))


def _set_arrays(cs):
    def rows(a):
        return a[np.lexsort(a.T[::-1])] if len(a) else a

    return dict(
        distances=rows(np.array([[d.object_id_a, d.keypoint_id_a, d.object_id_b, d.keypoint_id_b, d.distance, d.sigma] for d in cs.distances], dtype=np.float64).reshape(-1, 6)),
        centroids=rows(np.array([[c.object_id_a, c.object_id_b, c.distance, c.sigma] for c in cs.centroid_distances], dtype=np.float64).reshape(-1, 4)),
        static_ids=np.array(sorted(cs.static_object_ids), dtype=np.int64),
        remaps=rows(np.array([[r.object_id_from, r.keypoint_id_from, r.object_id_to, r.keypoint_id_to, r.obj_loc_x, r.obj_loc_y, r.obj_loc_z] for r in cs.point_remaps],
                             dtype=np.float64).reshape(-1, 7)),
        thickness=np.array(np.nan if cs.back_face_thickness_m is None else cs.back_face_thickness_m))


def test_the_compiler_fixtures_are_there():
    assert len(COMPILERS) == 8


@pytest.mark.parametrize("path", COMPILERS, ids=lambda p: p.stem)
def test_constraint_compilers_equal_the_reference_s_own_output(path):
    """Stand-ins carrying exactly what the compilers read from the reference's objects (stored by the generator, which also ran this package's
    compilers on the reference's OWN ``ArucoMarkerSet`` / ``Chessboard`` objects and recorded that the results were identical)."""
    from types import SimpleNamespace

    ref = np.load(path)
    assert ref["same_on_reference_objects"].all()
    markers = {int(m): SimpleNamespace(marker_id=int(m), size_m=float(sz), static=bool(st), corners=c)
               for m, sz, st, c in zip(ref["marker_ids"], ref["marker_size"], ref["marker_static"], ref["marker_corners"])}
    links = [SimpleNamespace(marker_a=int(a), marker_b=int(b), distance_m=float(d), corner_a=None if ca < 0 else int(ca), corner_b=None if cb < 0 else int(cb),
                             sigma_m=None if np.isnan(sg) else float(sg), is_center=bool(ic)) for a, b, d, ca, cb, sg, ic in ref["links"]]
    mirrors = [SimpleNamespace(marker_a=int(r[0]), marker_b=int(r[1]), anchor_corner_a=int(r[2]), anchor_corner_b=int(r[3]), thickness_m=float(r[4]),
                               sigma_m=None if np.isnan(r[5]) else float(r[5]), is_zero_thickness=bool(r[6]),
                               corner_mapping=tuple((int(r[7 + 2 * k]), int(r[8 + 2 * k])) for k in range(4))) for r in ref["mirrors"]]
    cs = ConstraintSet.from_marker_set(SimpleNamespace(markers=markers, links=tuple(links), mirror_pairs=tuple(mirrors)),
                                       sigma_m=float(ref["sigma"][0]), center_sigma_m=float(ref["sigma"][1]))
    for key, value in _set_arrays(cs).items():
        assert np.array_equal(value, ref[f"set_{key}"], equal_nan=True), key
    rows, cols, size_cm, sigma = ref["board"]
    pts = ref["board_points"].astype(np.float32)  # (Chessboard.get_object_points returns float32: chessboard.py:31-50)
    board = SimpleNamespace(rows=int(rows), columns=int(cols), square_size_cm=float(size_cm), get_object_points=lambda: pts)
    for key, value in _set_arrays(ConstraintSet.from_chessboard(board, sigma_m=float(sigma))).items():
        assert np.array_equal(value, ref[f"board_{key}"], equal_nan=True), key
    # from_charuco: thin boards and two-sided ones with a substrate (back face as object 1, ties at the thickness, braces) — the reference's compiler
    # was given the same stand-in (it reads board.getChessboardCorners(), board.getSquareLength() and thickness_m, nothing else)
    sq, thick, csig, tsig = (float(v) for v in ref["charuco"])
    grid = ref["charuco_corners"].astype(np.float32)
    ch = SimpleNamespace(board=SimpleNamespace(getChessboardCorners=lambda: grid, getSquareLength=lambda: sq), thickness_m=thick)
    for key, value in _set_arrays(ConstraintSet.from_charuco(ch, sigma_m=csig, thickness_sigma_m=tsig)).items():
        assert np.array_equal(value, ref[f"charuco_{key}"], equal_nan=True), key


# ---- the outlier filters between the solver passes (core/capture_volume.py:607-753), on an injected report ---------------------------------------
FILTERS = sorted((Path(__file__).parent / "golden" / "reference_host").glob("filter_*.npz"))


def test_the_filter_fixtures_are_there():
    assert len(FILTERS) == 6


@pytest.mark.parametrize("path", FILTERS, ids=lambda p: p.stem)
def test_outlier_filters_equal_the_reference_s_own_output(path):
    """The reference's filters read nothing of the reprojection report but ``raw_errors``; the fixture's volumes carried a report of the reference's
    own class with random errors in the cached property's slot (the generator says why).  Same tables, same errors here: the surviving
    observations must be the same rows in the same order, the surviving world points the same SET of rows (the reference re-attaches static points
    at the end of the table, this package keeps the table's order), every observation must still point at the same world key."""
    from caliscope_amd.capture_volume import ReprojectionReport

    ref = np.load(path)
    wdf = pd.DataFrame(ref["world"], columns=WORLD_COLS).astype({"sync_index": "int64", "object_id": "int64", "keypoint_id": "int64"})
    idf = pd.DataFrame(ref["image"], columns=IMG_COLS).astype({c: "int64" for c in IMG_COLS[:4]})
    K = np.array([[400.0, 0.0, 200.0], [0.0, 400.0, 200.0], [0.0, 0.0, 1.0]])
    cams = CameraArray({c: CameraData(cam_id=c, size=(400, 400), matrix=K.copy(), distortions=np.zeros(5), rotation=np.eye(3),
                                      translation=np.array([0.1 * c, 0.0, 0.0])) for c in (0, 1)})
    static = frozenset(int(o) for o in ref["static_ids"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vol = CaptureVolume(cams, ImagePoints(idf), WorldPoints(wdf), ConstraintSet((), static) if static else None)
    raw = pd.DataFrame(ref["raw_errors"], columns=["sync_index", "cam_id", "object_id", "keypoint_id", "error_x", "error_y", "euclidean_error"])
    raw = raw.astype({c: "int64" for c in ("sync_index", "cam_id", "object_id", "keypoint_id")})
    matched = vol.img_to_obj_map >= 0
    assert len(raw) == int(matched.sum()) and np.array_equal(raw["cam_id"].to_numpy(), idf["cam_id"].to_numpy()[matched])
    report = ReprojectionReport(overall_rmse=0.0, by_camera={}, by_point={}, n_unmatched_observations=int((~matched).sum()), unmatched_rate=0.0,
                                unmatched_by_camera={}, raw_errors=raw, n_observations_matched=len(raw), n_observations_total=len(idf), n_cameras=2, n_points=len(wdf))
    vol.__dict__["reprojection_report"] = report  # (the cached property's slot, as in the generator)

    def world_key_of_rows(volume):
        w = volume.world_points.df[WORLD_COLS[:3]].to_numpy()
        m = volume.img_to_obj_map
        return np.where(m[:, None] >= 0, w[np.maximum(m, 0)], -7)

    for n in range(int(ref["n_runs"])):
        kind, value, scope, floor = ref[f"run{n}_args"]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            if kind == 0.0:
                out = vol.filter_by_percentile_error(float(value), scope="per_camera" if scope == 0.0 else "overall", min_per_camera=int(floor))
            else:
                out = vol.filter_by_absolute_error(float(value), min_per_camera=int(floor))
        assert out.optimization_status is None
        mine = out.image_points.df[IMG_COLS].to_numpy(dtype=np.float64)
        if int(ref[f"run{n}_floor_cameras"]) >= 2:
            # Two or more cameras below the safety floor in one call: the reference, under the pandas of its lock file (2.3.3), tops up only the first
            # as documented (its boolean mask becomes an object column on the first assignment, `~mask` is then true everywhere — the generator has the
            # trace) and leaves the later ones BELOW the floor.  This package does what the reference's docstring says for every camera; the
            # reference's survivors are a subset of this package's, and no camera here ends below min(floor, its rows).
            theirs = {tuple(r) for r in ref[f"run{n}_image"][:, :4].astype(np.int64).tolist()}
            ours = {tuple(r) for r in mine[:, :4].astype(np.int64).tolist()}
            assert theirs <= ours, n  # (equal when the later cameras had nothing inside their threshold: all of their rows were "dropped" rows anyway)
            cam_rows, cam_kept = np.bincount(raw["cam_id"].to_numpy(), minlength=2), np.bincount(mine[:, 1].astype(np.int64), minlength=2)
            assert np.all(cam_kept >= np.minimum(int(floor), cam_rows)), (n, cam_kept, cam_rows, floor)
            continue
        assert np.array_equal(mine, ref[f"run{n}_image"]), n
        assert np.array_equal(_sorted_rows(out.world_points.df[WORLD_COLS].to_numpy(dtype=np.float64)), _sorted_rows(ref[f"run{n}_world"]), equal_nan=True), n
        theirs_w, theirs_m = ref[f"run{n}_world"][:, :3].astype(np.int64), ref[f"run{n}_map"]
        assert np.array_equal(world_key_of_rows(out), np.where(theirs_m[:, None] >= 0, theirs_w[np.maximum(theirs_m, 0)], -7)), n


# ---- the bookkeeping of the reprojection report (core/capture_volume.py:151-236) around injected pixel errors ----------------------------------
REPORTS = sorted((Path(__file__).parent / "golden" / "reference_host").glob("report_*.npz"))


def test_the_report_fixtures_are_there():
    assert len(REPORTS) == 6


@pytest.mark.parametrize("path", REPORTS, ids=lambda p: p.stem)
def test_report_bookkeeping_equals_the_reference_s_own_output(path, monkeypatch):
    """Camera array with an unposed camera (5) and an ignored one (9) that both have observations, and a posed camera (12) without any.  The pixel
    errors are the ones the generator handed to the reference in the place of its ``reprojection_errors`` (one row per counted observation, in the
    order the reference asked for them); they replace this package's device evaluation the same way."""
    ref = np.load(path)
    wdf = pd.DataFrame(ref["world"], columns=WORLD_COLS).astype({"sync_index": "int64", "object_id": "int64", "keypoint_id": "int64"})
    idf = pd.DataFrame(ref["image"], columns=IMG_COLS).astype({c: "int64" for c in IMG_COLS[:4]})
    K = np.array([[400.0, 0.0, 200.0], [0.0, 400.0, 200.0], [0.0, 0.0, 1.0]])

    def cam(c, posed=True, ignore=False):
        return CameraData(cam_id=c, size=(400, 400), matrix=K.copy(), distortions=np.zeros(5), ignore=ignore,
                          rotation=np.eye(3) if posed else None, translation=np.array([0.1 * c, 0.0, 0.0]) if posed else None)

    cams = CameraArray({0: cam(0), 1: cam(1), 5: cam(5, posed=False), 9: cam(9, ignore=True), 12: cam(12)})
    static = frozenset(int(o) for o in ref["static_ids"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vol = CaptureVolume(cams, ImagePoints(idf), WorldPoints(wdf), ConstraintSet((), static) if static else None)
    asked = {}

    def stored_errors(self, camera_indices, image_coords, obj_indices, _engine_factory=None):
        asked["camera_indices"] = np.asarray(camera_indices).copy()
        return ref["errors"].copy()

    monkeypatch.setattr(CaptureVolume, "_pixel_errors", stored_errors)
    rep = vol.compute_reprojection_report()
    assert np.array_equal(asked["camera_indices"], ref["camera_indices"])  # the same observations, in the same order, with the same camera numbering
    assert rep.overall_rmse == pytest.approx(float(ref["overall_rmse"]), rel=1e-14)
    mine = np.array(sorted(rep.by_camera.items()), dtype=np.float64).reshape(-1, 2)
    assert np.array_equal(mine[:, 0], ref["by_camera"][:, 0]) and np.allclose(mine[:, 1], ref["by_camera"][:, 1], rtol=1e-14, atol=0)
    mine = np.array(sorted((o, k, v) for (o, k), v in rep.by_point.items()), dtype=np.float64).reshape(-1, 3)
    assert np.array_equal(mine[:, :2], ref["by_point"][:, :2]) and np.allclose(mine[:, 2], ref["by_point"][:, 2], rtol=1e-14, atol=0)
    assert rep.n_unmatched_observations == int(ref["n_unmatched"]) and rep.unmatched_rate == pytest.approx(float(ref["unmatched_rate"]), rel=1e-15)
    assert np.array_equal(np.array(sorted(rep.unmatched_by_camera.items()), dtype=np.int64).reshape(-1, 2), ref["unmatched_by_camera"])
    assert list(rep.raw_errors.columns) == [str(c) for c in ref["raw_columns"]]
    got = rep.raw_errors.to_numpy(dtype=np.float64)
    assert np.array_equal(got[:, :6], ref["raw_errors"][:, :6]) and np.allclose(got[:, 6], ref["raw_errors"][:, 6], rtol=1e-15, atol=0)
    assert [rep.n_observations_matched, rep.n_observations_total, rep.n_cameras, rep.n_points] == ref["counts"].tolist()


# ---- the seam: what optimize() hands to least_squares and what it makes of the result (core/capture_volume.py:322-444) ---------------------------
SEAMS = sorted((Path(__file__).parent / "golden" / "reference_host").glob("seam_*.npz"))


def test_the_seam_fixtures_are_there():
    assert len(SEAMS) == 10


@pytest.mark.parametrize("path", SEAMS, ids=lambda p: p.stem)
def test_optimize_hands_over_and_takes_back_what_the_reference_does(path, monkeypatch):
    """The generator ran the reference's ``optimize()`` with ``least_squares`` replaced by a recorder returning a scripted result.  Here this
    package's ``optimize()`` runs on the same volume with ITS ``least_squares`` replaced the same way (returning the stored result vector): the
    call must carry what the reference's carried — ``x0``, the observation arrays, the constraint rows with their weights, the bounds, every keyword —
    and the returned volume must be the one the reference built: cameras, points, ``OptimizationStatus`` with its reason string and bound
    warnings, or the same ``CalibrationError`` text for a strict call that did not converge."""
    import caliscope_amd.capture_volume as cv_mod
    from caliscope_amd.cameras import rvec_to_matrix
    from caliscope_amd.exceptions import CalibrationError

    ref = np.load(path)
    wdf = pd.DataFrame(ref["world"], columns=WORLD_COLS).astype({"sync_index": "int64", "object_id": "int64", "keypoint_id": "int64"})
    idf = pd.DataFrame(ref["image"], columns=IMG_COLS).astype({c: "int64" for c in IMG_COLS[:4]})
    cams = {}
    for i, cid in enumerate(ref["cam_ids"]):
        fisheye, posed = bool(ref["fisheye"][i]), not np.isnan(ref["rvec"][i]).any()
        cams[int(cid)] = CameraData(cam_id=int(cid), size=(int(ref["sizes"][i][0]), int(ref["sizes"][i][1])), matrix=ref["K"][i].copy(),
                                    distortions=ref["dist"][i][: 4 if fisheye else 5].copy(), fisheye=fisheye, ignore=bool(ref["ignore"][i]),
                                    rotation=rvec_to_matrix(ref["rvec"][i]) if posed else None, translation=ref["t"][i].copy() if posed else None)
    cs = None
    if bool(ref["has_constraints"]):
        cs = ConstraintSet(tuple(DistanceConstraint(int(a), int(b), int(c), int(d), float(e), float(f)) for a, b, c, d, e, f in ref["distances"]),
                           frozenset(int(o) for o in ref["static_ids"]),
                           centroid_distances=tuple(CentroidDistanceConstraint(int(a), int(b), float(c), float(d)) for a, b, c, d in ref["centroids"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        vol = CaptureVolume(CameraArray(cams), ImagePoints(idf), WorldPoints(wdf), cs)
    ftol, max_nfev, verbose, strict, use_constraints, pixel_sigma, refine, f_scale = ref["call"]
    seen = {}

    from types import SimpleNamespace

    def recorder(fun, x0, args=(), jac=None, **kwargs):
        seen.update(x0=np.array(x0, dtype=np.float64), args=args, kwargs=kwargs)
        return SimpleNamespace(x=ref["x_result"].copy(), status=int(ref["result_status"]), nfev=17, cost=1.25, optimality=1e-9, success=int(ref["result_status"]) > 0)

    monkeypatch.setattr(cv_mod, "least_squares", recorder)
    kw = dict(ftol=float(ftol), max_nfev=None if max_nfev < 0 else int(max_nfev), verbose=int(verbose), strict=bool(strict), use_constraints=bool(use_constraints),
              pixel_sigma=float(pixel_sigma), refine_intrinsics=bool(refine), loss=str(ref["loss"]), f_scale=float(f_scale))
    error, out = "", None
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = vol.optimize(**kw)
    except CalibrationError as exc:
        error = str(exc)
    # -- what was handed over
    par, cam_idx, uv, obj_idx, ga, gb, cd, cw = seen["args"][:8]
    assert np.array_equal(par.camera_param_offsets, ref["offsets"]) and par.n_camera_params == int(ref["n_camera_params"])
    rot = np.zeros(seen["x0"].size, dtype=bool)
    for off in par.camera_param_offsets:
        rot[off:off + 3] = True
    assert seen["x0"].shape == ref["x0"].shape and np.array_equal(seen["x0"][~rot], ref["x0"][~rot]) and np.allclose(seen["x0"][rot], ref["x0"][rot], rtol=0, atol=1e-12)
    assert np.array_equal(np.asarray(cam_idx, dtype=np.int64), ref["camera_indices"]) and np.array_equal(np.asarray(uv, dtype=np.float64), ref["image_coords"])
    assert np.array_equal(np.asarray(obj_idx, dtype=np.int64), ref["obj_indices"])
    assert (ga is not None) == bool(ref["has_rows"])
    if ga is not None:
        # (the reference walks a set of sync indices: the order of the rows is not defined there; a row keeps its distance and its weight)
        assert np.array_equal(_sorted_rows(ga, gb, cd, cw), _sorted_rows(ref["groups_a"], ref["groups_b"], ref["row_distance"], ref["row_weight"]))
    lb, ub = seen["kwargs"]["bounds"]
    assert np.array_equal(np.asarray(lb), ref["lb"]) and np.array_equal(np.asarray(ub), ref["ub"])
    theirs = dict(zip((str(k) for k in ref["kwargs_keys"]), (str(v) for v in ref["kwargs_values"])))
    mine = {k: str(v) for k, v in seen["kwargs"].items() if k not in ("bounds", "engine_factory")}
    assert mine == theirs, (mine, theirs)
    # -- what was made of the result
    assert error == str(ref["error"]) and (out is not None) == bool(ref["returned"])
    if out is None:
        return
    for k, cid in enumerate(ref["cam_ids"]):
        cam, src = out.camera_array.cameras[int(cid)], vol.camera_array.cameras[int(cid)]
        assert cam is not src and np.array_equal(src.matrix, ref["K"][k])  # the source volume's cameras are not touched
        if np.isnan(ref["out_R"][k]).any():
            assert cam.rotation is None and cam.translation is None
        else:
            assert np.allclose(cam.rotation, ref["out_R"][k], rtol=0, atol=1e-12) and np.array_equal(np.ravel(cam.translation), ref["out_t"][k])
        assert np.array_equal(cam.matrix, ref["out_K"][k])
        d = np.ravel(cam.distortions)
        assert np.array_equal(d, ref["out_dist"][k][: d.size]) and np.all(np.isnan(ref["out_dist"][k][d.size:]))
    assert np.array_equal(out.world_points.df[WORLD_COLS].to_numpy(dtype=np.float64), ref["out_world"], equal_nan=True)
    assert np.array_equal(out.img_to_obj_map, ref["out_map"]) and out.image_points is vol.image_points
    st = out.optimization_status
    assert bool(st.converged) == bool(ref["status_fields"][0]) and st.termination_reason == str(ref["status_reason"])
    assert st.iterations == 17 and st.final_cost == 1.25
    got = np.array([[w.cam_id, {"f": 0, "k1": 1, "k2": 2}[w.parameter], {"lower": 0, "upper": 1}[w.bound], w.value] for w in st.bound_warnings], dtype=np.float64).reshape(-1, 4)
    assert np.array_equal(_sorted_rows(got), _sorted_rows(ref["status_warnings"]))


# ---- the constraint rows of the reference's own joint_residuals / joint_jacobian (core/reprojection.py:112-117, :207-226) ----------------------
CONROWS = sorted((Path(__file__).parent / "golden" / "reference_host").glob("conrows_*.npz"))


def _conrow_parameterization(ref):
    from caliscope_amd.bundle_parameterization import BundleParameterization

    K = np.array([[400.0, 0.0, 200.0], [0.0, 400.0, 200.0], [0.0, 0.0, 1.0]])
    cams = CameraArray({c: CameraData(cam_id=c, size=(400, 400), matrix=K.copy(), distortions=np.zeros(5), rotation=np.eye(3), translation=np.zeros(3))
                        for c in range(int(ref["n_cams"]))})
    return BundleParameterization.from_camera_array(cams, int(ref["n_points"]), refine_intrinsics=bool(ref["refine"]))


def test_the_constraint_row_fixtures_are_there():
    assert len(CONROWS) == 6


@pytest.mark.parametrize("path", CONROWS, ids=lambda p: p.stem)
def test_oracle_constraint_rows_equal_the_reference_s_own(path):
    """The fixtures hold what the reference's OWN ``joint_residuals`` / ``joint_jacobian`` returned when called without observations (their
    per-camera loops, the only place they reach OpenCV, then do nothing): the constraint rows and their Jacobian from the reference's numpy code.
    The oracle — what the device rows are compared with — must reproduce them."""
    from oracle.residuals import joint_jacobian, joint_residuals

    ref = np.load(path)
    par = _conrow_parameterization(ref)
    none_i, none_uv = np.zeros(0, dtype=np.int32), np.zeros((0, 2))
    args = (ref["x"], par, none_i, none_uv, none_i, ref["groups_a"], ref["groups_b"], ref["distances"], ref["weights"])
    r = joint_residuals(*args)
    scale = np.abs(ref["residuals"]).max()
    assert r.shape == ref["residuals"].shape and np.allclose(r, ref["residuals"], rtol=0, atol=1e-13 * scale)
    J = np.asarray(joint_jacobian(*args).todense())
    assert J.shape == ref["jacobian"].shape and np.allclose(J, ref["jacobian"], rtol=0, atol=1e-13 * np.abs(ref["jacobian"]).max())
    assert np.array_equal(J != 0, ref["jacobian"] != 0)  # the same pattern: zero subgradient at coincident endpoints, shared rows folded


@pytest.mark.gpu
@pytest.mark.parametrize("path", CONROWS, ids=lambda p: p.stem)
def test_device_constraint_rows_equal_the_reference_s_own(path):
    """The same stored rows against the device: the residual hook appends the constraint rows behind the reprojection rows of a (here: two-row)
    observation list; rows in the caller's order."""
    from caliscope_amd.engine import BAProblem
    from caliscope_amd.hip_engine import HipEngine

    ref = np.load(path)
    par = _conrow_parameterization(ref)
    cam_idx, uv, obj = np.array([0, 1], dtype=np.int32), np.array([[200.0, 200.0], [210.0, 190.0]]), np.array([0, 1], dtype=np.int32)
    prob = BAProblem(par, cam_idx, uv, obj, constraint_groups_a=ref["groups_a"], constraint_groups_b=ref["groups_b"], constraint_distances=ref["distances"],
                     constraint_weights=ref["weights"])
    with HipEngine(prob, evaluation_only=True) as eng:
        r, _ = eng.residuals(ref["x"])
    assert r.size == 4 + ref["residuals"].size
    assert np.allclose(r[4:], ref["residuals"], rtol=0, atol=1e-12 * np.abs(ref["residuals"]).max())


# ---- the stage driver calibrate_extrinsics (core/calibrate_extrinsics.py:44-261) with its three heavy calls scripted ---------------------------
DRIVERS = sorted((Path(__file__).parent / "golden" / "reference_host").glob("driver_*.npz"))


def test_the_driver_fixtures_are_there():
    assert len(DRIVERS) == 11


@pytest.mark.parametrize("path", DRIVERS, ids=lambda p: p.stem)
def test_stage_driver_does_what_the_reference_s_does(path):
    """``CaptureVolume.bootstrap`` hands back a stored "triangulation", ``optimize`` and the filter record their arguments (tests/driver_script.py,
    the same stand-ins the generator put on the reference's class); the rest of the driver is real.  The trace — progress marks, what bootstrap
    received (again after a dropped static marker), every ``optimize`` call's keywords, the filter's percentile — and the ``CalibrationRun`` must be the
    reference's: blind intrinsics, the static-marker guard on the rigidity report, the depth-ratio gate on intrinsic refinement, the estimates."""
    import ast

    from caliscope_amd.calibrate_extrinsics import calibrate_extrinsics
    from caliscope_amd.constraints import PointRemap
    from caliscope_amd.exceptions import CalibrationError
    from tests.driver_script import scripted

    ref = np.load(path)
    wdf = pd.DataFrame(ref["world"], columns=WORLD_COLS).astype({"sync_index": "int64", "object_id": "int64", "keypoint_id": "int64"})
    idf = pd.DataFrame(ref["image"], columns=IMG_COLS + ["obj_loc_x", "obj_loc_y", "obj_loc_z"]).astype({c: "int64" for c in IMG_COLS[:4]})
    cams = CameraArray({int(c): CameraData(cam_id=int(c), size=(int(sz[0]), int(sz[1])), matrix=K.copy() if has else None, distortions=d.copy() if has else None,
                                           ignore=bool(ig), rotation=np.eye(3), translation=t.copy())
                        for c, sz, K, d, ig, has, t in zip(ref["cam_ids"], ref["sizes"], ref["K"], ref["dist"], ref["ignore"], ref["has_intrinsics"], ref["t"])})
    cs = None
    if bool(ref["has_constraints"]):
        cs = ConstraintSet(tuple(DistanceConstraint(int(a), int(b), int(c), int(d), float(e), float(f)) for a, b, c, d, e, f in ref["distances"]),
                           frozenset(int(o) for o in ref["static_ids"]),
                           point_remaps=tuple(PointRemap(int(r[0]), int(r[1]), int(r[2]), int(r[3]), float(r[4]), float(r[5]), float(r[6])) for r in ref["remaps"]),
                           back_face_thickness_m=float(ref["thickness"]) if "thickness" in ref.files else None)
    if "firing" in ref.files:  # two-sided board: the count of cross-face rows that fire on the stored triangulation
        from caliscope_amd.calibrate_extrinsics import _count_firing_cross_face_rows

        assert _count_firing_cross_face_rows(wdf, cs.distances) == int(ref["firing"])
    trace, run, error = [], None, None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with scripted(CaptureVolume, WorldPoints, wdf, trace):
            try:
                run = calibrate_extrinsics(ImagePoints(idf), cams, cs, refine_intrinsics=bool(ref["refine"]), filter_percentile=float(ref["filter_percentile"]),
                                           progress=lambda pct, msg: trace.append(("progress", int(pct), str(msg))))
            except CalibrationError as exc:
                error = exc
    want = ast.literal_eval(str(ref["trace"]))
    assert trace == want, "\n".join(f"{a}\n{b}" for a, b in zip(trace, want) if a != b)
    assert (run is not None) == bool(ref["returned"]) and ("CalibrationError" if error is not None else "") == str(ref["error_type"])
    if run is None:
        assert all(str(int(c)) in str(error) for c in ref["error_mentions"])  # (the message names the cameras that fell back to blind intrinsics; its wording is this package's)
        if "error_words" in ref.files:  # (two-sided board: which of the two guards spoke)
            assert all(str(w) in str(error).lower() for w in ref["error_words"]) and len(ref["error_words"])
        return
    assert sorted(run.synthesized_cam_ids) == ref["synthesized"].tolist() and list(run.dropped_static_markers) == ref["dropped"].tolist()
    assert bool(run.intrinsic_refinement_gated) == bool(ref["gated"])
    est = np.array([[e.cam_id, e.f_recovered, e.k1_recovered, e.k2_recovered, e.f_initial, e.k1_initial, e.k2_initial] for e in run.intrinsic_estimates],
                   dtype=np.float64).reshape(-1, 7)
    assert np.array_equal(est, ref["estimates"])
    assert [len(run.capture_volume.image_points.df), len(run.capture_volume.world_points.df)] == ref["final_counts"].tolist()
    assert np.array_equal(run.capture_volume.image_points.df[IMG_COLS[:4]].to_numpy(dtype=np.int64), ref["final_image_keys"])


# ---- the reference's batched SVD triangulation (core/point_data.py:121-232), a plain numpy function ---------------------------------------------
DLTS = sorted((Path(__file__).parent / "golden" / "reference_host").glob("dlt_*.npz"))


def test_the_triangulation_fixtures_are_there():
    assert len(DLTS) == 6


@pytest.mark.parametrize("path", DLTS, ids=lambda p: p.stem)
def test_oracle_triangulation_equals_the_reference_s_own(path):
    """``triangulate_image_points`` of the reference needs no OpenCV: its output on random rigs (points seen by one camera are left out, the rest
    solved by SVD per camera set) against the oracle's restatement — the function the device triangulation is compared with on the GPU
    (tests/test_triangulation.py).  Same points (set of keys), coordinates to 1e-9."""
    from oracle import triangulation as otri

    ref = np.load(path)
    P = {int(c): p for c, p in zip(ref["cam_ids"], ref["P"])}
    sync, obj, kp, xyz = otri.triangulate_image_points(P, ref["sync"], ref["cam"], ref["obj"], ref["kp"], ref["xy"])
    mine = _sorted_rows(np.column_stack([sync, obj, kp]), xyz)
    want = _sorted_rows(np.column_stack([ref["out_sync"], ref["out_obj"], ref["out_kp"]]), ref["out_xyz"])
    assert mine.shape == want.shape and np.array_equal(mine[:, :3], want[:, :3])
    assert np.allclose(mine[:, 3:], want[:, 3:], rtol=0, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("path", DLTS, ids=lambda p: p.stem)
def test_device_triangulation_equals_the_reference_s_own(path):
    """The same stored output against ``cba_triangulate`` (no undistortion: the fixture's coordinates are normalised already)."""
    from caliscope_amd.triangulation import triangulate_image_points

    ref = np.load(path)
    P = {int(c): p for c, p in zip(ref["cam_ids"], ref["P"])}
    sync, obj, kp, xyz = triangulate_image_points(P, ref["sync"], ref["cam"], ref["obj"], ref["kp"], ref["xy"])
    mine = _sorted_rows(np.column_stack([sync, obj, kp]), xyz)
    want = _sorted_rows(np.column_stack([ref["out_sync"], ref["out_obj"], ref["out_kp"]]), ref["out_xyz"])
    assert mine.shape == want.shape and np.array_equal(mine[:, :3], want[:, :3])
    assert np.allclose(mine[:, 3:], want[:, 3:], rtol=0, atol=1e-9)


# ---- ImagePoints.triangulate (core/point_data.py:416-560) for cameras without lens distortion ------------------------------------------------------
TRIANGULATES = sorted((Path(__file__).parent / "golden" / "reference_host").glob("triangulate_*.npz"))


def _oracle_run(cam_P, starts, cam_index, xy, *, cam_model=None, cam_intr=None, float32_io=True, device_id=0, want_undistorted=False):
    """What ``caliscope_amd.triangulation._run`` asks of the device (undistort every observation with its camera's table, one DLT per group), from the
    oracle (oracle/triangulation.py): the CPU stand-in for ``cba_triangulate``, which has its own device-against-oracle tests."""
    from oracle import triangulation as otri

    xy = np.asarray(xy, dtype=np.float64).reshape(-1, 2)
    und = xy.copy()
    if cam_intr is not None:
        for c in np.unique(cam_index):
            m = cam_index == c
            fx, fy, cx, cy = cam_intr[c, :4]
            K = np.array([[fx, 0.0, cx], [0.0, fy, cy], [0.0, 0.0, 1.0]])
            und[m] = otri.undistort_points(xy[m], K, cam_intr[c, 4:9] if not cam_model[c] else cam_intr[c, 4:8], bool(cam_model[c]), float32_io=float32_io)
    xyz = np.full((len(starts) - 1, 3), np.nan)
    for g in range(len(starts) - 1):
        a, b = int(starts[g]), int(starts[g + 1])
        if b - a >= 2:
            xyz[g] = otri.triangulate_point([np.asarray(cam_P[cam_index[i]]).reshape(3, 4) for i in range(a, b)], und[a:b])
    return xyz, (und if want_undistorted else None)


def test_the_high_level_triangulation_fixtures_are_there():
    assert len(TRIANGULATES) == 6


def _triangulate_fixture(path, monkeypatch=None):
    import caliscope_amd.triangulation as tri
    from caliscope_amd.cameras import rvec_to_matrix

    ref = np.load(path)
    idf = pd.DataFrame(ref["image"], columns=IMG_COLS + ["frame_time"]).astype({c: "int64" for c in IMG_COLS[:4]})
    cams = CameraArray({int(c): CameraData(cam_id=int(c), size=(1280, 720), matrix=K.copy(), distortions=np.zeros(5), ignore=bool(ig),
                                           rotation=rvec_to_matrix(rv) if p else None, translation=t.copy() if p else None)
                        for c, K, rv, t, p, ig in zip(ref["cam_ids"], ref["K"], ref["rvec"], ref["t"], ref["posed"], ref["ignore"])})
    if monkeypatch is not None:
        monkeypatch.setattr(tri, "_run", _oracle_run)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wp = ImagePoints(idf).triangulate(cams, static_object_ids=frozenset(int(o) for o in ref["static_ids"]))
    out = wp.df
    assert [c for c in ref["world_columns"] if c in WORLD_COLS] == [c for c in out.columns if c in WORLD_COLS]
    mine, want = _sorted_rows(out[WORLD_COLS].to_numpy(dtype=np.float64)), _sorted_rows(ref["world"])
    assert mine.shape == want.shape and np.array_equal(mine[:, :3], want[:, :3])
    assert np.allclose(mine[:, 3:6], want[:, 3:6], rtol=0, atol=1e-8)
    assert np.array_equal(np.isnan(mine[:, 6]), np.isnan(want[:, 6])) and np.allclose(mine[:, 6], want[:, 6], rtol=0, atol=1e-12, equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("path", TRIANGULATES, ids=lambda p: p.stem)
def test_triangulate_on_the_device_equals_the_reference_s_own_output(path):
    """The same fixtures through ``cba_triangulate`` itself (undistortion of float32 pixel input and DLT on the device)."""
    _triangulate_fixture(path)


@pytest.mark.parametrize("path", TRIANGULATES, ids=lambda p: p.stem)
def test_triangulate_host_logic_equals_the_reference_s_own_output(path, monkeypatch):
    """Which cameras take part (posed and not ignored), static objects pooled into one point at STATIC_SYNC_INDEX, points seen once left out,
    frame times, float32 pixel input, the table that comes back — against the reference's own ``ImagePoints.triangulate`` (zero-distortion
    cameras: the generator says what stood in for ``cv2.undistortPoints``).  The device call is replaced by the oracle here; the device has its own
    tests against the same oracle and against the reference's numpy triangulation (above)."""
    _triangulate_fixture(path, monkeypatch)


# ---- ConstraintSet.remap_image_points (core/constraints.py:192-214) ------------------------------------------------------------------------------
REMAPS = sorted((Path(__file__).parent / "golden" / "reference_host").glob("remap_*.npz"))


def test_the_remap_fixtures_are_there():
    assert len(REMAPS) == 6


@pytest.mark.parametrize("path", REMAPS, ids=lambda p: p.stem)
def test_observation_remaps_equal_the_reference_s_own_output(path):
    """Arbitrary remap tuples — some chained (the target of one is the source of a later one: the reference applies them in turn on the same frame),
    some matching nothing; identity and board coordinates rewritten, everything else of the row kept; no remaps: the input object itself."""
    from caliscope_amd.constraints import PointRemap

    ref = np.load(path)
    cols = IMG_COLS + ["obj_loc_x", "obj_loc_y", "obj_loc_z"]
    idf = pd.DataFrame(ref["image"], columns=cols).astype({c: "int64" for c in IMG_COLS[:4]})
    cs = ConstraintSet((), frozenset(), point_remaps=tuple(PointRemap(int(r[0]), int(r[1]), int(r[2]), int(r[3]), float(r[4]), float(r[5]), float(r[6])) for r in ref["remaps"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = cs.remap_image_points(ImagePoints(idf.copy())).df
        ip = ImagePoints(idf.copy())
        assert (ConstraintSet((), frozenset()).remap_image_points(ip) is ip) == bool(ref["no_remaps_returns_the_input"])
    _same_table(out, ref["out"], [str(c) for c in ref["out_columns"]])


# ---- CameraArray.from_toml (cameras/camera_array.py:377-441) on camera files of the reference's side ---------------------------------------------
def test_camera_files_load_as_the_reference_loads_them(tmp_path):
    """The real session's ``camera_array.toml`` (3 x 3 rotations, a key this package does not know), and the shapes the reference's loader accepts:
    rotation as a 3-vector or a 3 x 1 column, the string "null" for a missing value, absent optional keys, a camera with nothing but its size, a
    fisheye camera, an empty file."""
    ref = np.load(Path(__file__).parent / "golden" / "reference_host" / "camtoml_00.npz")

    def opt(v):
        return np.nan if v is None else float(v)

    for name in (str(n) for n in ref["names"]):
        path = tmp_path / f"{name}.toml"
        path.write_text(str(ref[f"{name}_text"]))
        arr = CameraArray.from_toml(path)
        ids = sorted(arr.cameras)
        assert ids == ref[f"{name}_ids"].tolist(), name
        for k, c in enumerate(ids):
            cam = arr.cameras[c]
            mine = np.array([cam.size[0], cam.size[1], cam.rotation_count, opt(cam.error), opt(cam.exposure), opt(cam.grid_count), float(bool(cam.ignore)), float(bool(cam.fisheye))])
            assert np.array_equal(mine, ref[f"{name}_scalars"][k], equal_nan=True), (name, c, mine, ref[f"{name}_scalars"][k])
            n_dist = int(ref[f"{name}_n_dist"][k])
            assert (cam.distortions is None) == (n_dist < 0) and (cam.matrix is None) == bool(np.isnan(ref[f"{name}_K"][k]).all())
            if cam.matrix is not None:
                assert np.array_equal(cam.matrix, ref[f"{name}_K"][k]) and np.array_equal(np.ravel(cam.distortions), ref[f"{name}_dist"][k][:n_dist])
            posed = not np.isnan(ref[f"{name}_R"][k]).any()
            assert (cam.rotation is not None) == posed and (cam.translation is not None) == (not np.isnan(ref[f"{name}_t"][k]).any())
            if posed:
                assert np.allclose(cam.rotation, ref[f"{name}_R"][k], rtol=0, atol=1e-12) and np.array_equal(np.ravel(cam.translation), ref[f"{name}_t"][k])
