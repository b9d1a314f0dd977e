"""fill_gaps / filter_to_objects / smooth of the point tables against a track-by-track restatement of what the reference does
(per track: reindex to every sync index, drop the rows beyond max_gap_size inside a hole, pandas linear interpolation)."""
import numpy as np
import pandas as pd

from caliscope_amd.point_data import ImagePoints, WorldPoints


def _track_by_track(df, keys, cols, max_gap):
    out = []
    for vals, group in df.groupby(keys):
        group = group.sort_values("sync_index")
        full = pd.DataFrame({"sync_index": np.arange(group["sync_index"].min(), group["sync_index"].max() + 1)})
        for k, v in zip(keys, vals):
            full[k] = int(v)
        m = pd.merge(full, group, on=keys + ["sync_index"], how="left")
        gap = m[cols[0]].isnull().astype(int).groupby(m[cols[0]].notnull().cumsum()).cumsum()
        m = m[gap <= max_gap]
        for c in cols:
            m[c] = m[c].interpolate(method="linear", limit=max_gap)
        out.append(m)
    return pd.concat(out).dropna(subset=[cols[0]])


def _image_table(rng):
    rows = []
    for cam in range(3):
        for kp in range(4):
            frames = np.flatnonzero(rng.random(60) < 0.7) + 5
            for f in frames:
                rows.append(dict(sync_index=int(f), cam_id=cam, object_id=kp // 2, keypoint_id=kp, img_loc_x=100 + 3.0 * f + cam + rng.normal(),
                                 img_loc_y=50 + 0.5 * f * f / 10 + kp, frame_time=f / 30.0, obj_loc_x=0.1 * kp, obj_loc_y=0.0, obj_loc_z=0.0))
    df = pd.DataFrame(rows)
    return df.sample(frac=1.0, random_state=1).reset_index(drop=True)


def test_image_point_gaps_are_filled_like_the_reference_does():
    df = _image_table(np.random.default_rng(0))
    keys, cols = ["cam_id", "object_id", "keypoint_id"], ["img_loc_x", "img_loc_y", "frame_time"]
    for max_gap in (1, 3, 6):
        got = ImagePoints(df).fill_gaps(max_gap).df.sort_values(keys + ["sync_index"]).reset_index(drop=True)
        ref = _track_by_track(ImagePoints(df).df, keys, cols, max_gap).sort_values(keys + ["sync_index"]).reset_index(drop=True)
        assert len(got) == len(ref) > len(df)
        for c in keys + ["sync_index"]:
            assert np.array_equal(got[c].to_numpy(), ref[c].to_numpy()), c
        for c in cols:
            assert np.allclose(got[c].to_numpy(), ref[c].to_numpy(), rtol=0, atol=1e-12), c
        assert got["obj_loc_x"].isna().sum() == len(got) - len(df)  # inserted rows carry no board coordinates
    assert len(ImagePoints(df).fill_gaps(0)) == len(df)
    only = ImagePoints(df).filter_to_objects([1])
    assert set(only.df["object_id"]) == {1} and len(only) == int((df["object_id"] == 1).sum())


def test_world_point_gaps_and_smoothing():
    rng = np.random.default_rng(2)
    rows = []
    for kp in range(5):
        frames = np.flatnonzero(rng.random(80) < 0.75)
        for f in frames:
            rows.append(dict(sync_index=int(f), object_id=0, keypoint_id=kp, x_coord=np.sin(f / 9.0) + 0.02 * rng.normal(), y_coord=f / 40.0,
                             z_coord=0.1 * kp, frame_time=f / 30.0))
    df = pd.DataFrame(rows).sample(frac=1.0, random_state=3).reset_index(drop=True)
    keys, cols = ["object_id", "keypoint_id"], ["x_coord", "y_coord", "z_coord", "frame_time"]
    got = WorldPoints(df).fill_gaps(3).df.sort_values(keys + ["sync_index"]).reset_index(drop=True)
    ref = _track_by_track(WorldPoints(df).df, keys, cols, 3).sort_values(keys + ["sync_index"]).reset_index(drop=True)
    assert len(got) == len(ref) and np.array_equal(got["sync_index"].to_numpy(), ref["sync_index"].to_numpy())
    for c in cols:
        assert np.allclose(got[c].to_numpy(), ref[c].to_numpy(), rtol=0, atol=1e-12), c
    # smoothing: the reference filters every trajectory in table order with filtfilt
    from scipy.signal import butter, filtfilt

    wp = WorldPoints(df)
    sm = wp.smooth(fps=30.0, cutoff_freq=3.0, order=2).df
    b, a = butter(2, 3.0, btype="low", fs=30.0, output="ba")
    base = wp.df
    for _, group in base.groupby(["object_id", "keypoint_id"]):
        for c in ("x_coord", "y_coord", "z_coord"):
            assert np.allclose(sm.loc[group.index, c].to_numpy(), filtfilt(b, a, group[c].to_numpy()), atol=1e-12)
    short = WorldPoints(df[df["keypoint_id"] == 0].head(5))
    assert np.array_equal(short.smooth(30.0, 3.0).points, short.points)  # 5 samples <= 3 * order: untouched


def test_world_point_index_range_ignores_static_rows():
    """Reference tests/test_constraints.py:390-424."""
    from caliscope_amd.point_data import STATIC_SYNC_INDEX

    base = dict(object_id=0, keypoint_id=0, x_coord=0.0, y_coord=0.0, z_coord=0.0)
    wp = WorldPoints(pd.DataFrame([dict(base, sync_index=STATIC_SYNC_INDEX), dict(base, sync_index=5, keypoint_id=1), dict(base, sync_index=9, keypoint_id=1)]))
    assert (wp.min_index, wp.max_index) == (5, 9)
    static_only = WorldPoints(pd.DataFrame([dict(base, sync_index=STATIC_SYNC_INDEX), dict(base, sync_index=STATIC_SYNC_INDEX, keypoint_id=1)]))
    assert (static_only.min_index, static_only.max_index) == (0, 0)


def test_world_points_with_points_and_cached_columns():
    """The fast paths of ``CaptureVolume.optimize``: new coordinates on validated keys, numpy views of the observation table."""
    import pandas as pd
    import pytest

    from caliscope_amd.point_data import ImagePoints, WorldPoints

    world = WorldPoints(pd.DataFrame({"sync_index": [0, 0, 1], "object_id": 0, "keypoint_id": [0, 1, 0], "x_coord": [0.0, 1.0, 2.0],
                                      "y_coord": 0.5, "z_coord": [1.0, 1.0, 1.5], "frame_time": [0.0, 0.0, 0.1]}))
    xyz = np.arange(9.0).reshape(3, 3)
    moved = world.with_points(xyz)
    assert np.array_equal(moved.points, xyz) and np.array_equal(moved.df[["x_coord", "y_coord", "z_coord"]].to_numpy(), xyz)
    assert np.array_equal(world.points[:, 0], [0.0, 1.0, 2.0])  # the source table is untouched
    assert moved.df[["sync_index", "object_id", "keypoint_id", "frame_time"]].equals(world.df[["sync_index", "object_id", "keypoint_id", "frame_time"]])
    assert (moved.min_index, moved.max_index) == (world.min_index, world.max_index) == (0, 1)
    pts = moved.points
    pts[:] = -1.0
    assert np.array_equal(moved.points, xyz)  # `points` hands out copies
    xyz[:] = 7.0  # the caller's array is not what the new table keeps
    assert np.array_equal(moved.points, np.arange(9.0).reshape(3, 3)) and np.array_equal(moved.df["x_coord"].to_numpy(), [0.0, 3.0, 6.0])
    again = moved.with_points(moved.points + 1.0).take(np.array([True, False, True]))  # tables built over shared key columns behave like any other
    assert list(again.df["keypoint_id"]) == [0, 0] and np.array_equal(again.points, [[1.0, 2.0, 3.0], [7.0, 8.0, 9.0]])
    with pytest.raises(ValueError):
        world.with_points(np.zeros((2, 3)))
    with pytest.raises(ValueError, match="non-finite"):
        world.with_points(np.full((3, 3), np.nan))

    img = ImagePoints(pd.DataFrame({"sync_index": [0, 0], "cam_id": [1, 2], "object_id": 0, "keypoint_id": [0, 0], "img_loc_x": [10.0, 20.0],
                                    "img_loc_y": [1.0, 2.0]}))
    cols = img.arrays()
    assert cols is img.arrays() and set(cols) == {"sync_index", "cam_id", "object_id", "keypoint_id", "img_loc_x", "img_loc_y"}
    assert cols["cam_id"].dtype == np.int64 and cols["img_loc_x"].dtype == np.float64
    with pytest.raises(ValueError):
        cols["img_loc_x"][0] = 0.0


def test_marshalling_with_sparse_and_unposed_camera_ids():
    """Observation -> camera-position mapping: cameras without a pose and ids that do not appear in the array drop out, ids
    far apart take the search path instead of the dense table."""
    import pandas as pd

    from caliscope_amd.cameras import CameraArray, CameraData
    from caliscope_amd.capture_volume import CaptureVolume
    from caliscope_amd.point_data import ImagePoints, WorldPoints

    K = np.array([[800.0, 0, 320], [0, 800.0, 240], [0, 0, 1]])

    def cam(cid, posed=True):
        return CameraData(cam_id=cid, size=(640, 480), matrix=K.copy(), distortions=np.zeros(5), rotation=np.eye(3) if posed else None,
                          translation=np.array([0.0, 0.0, 2.0]) if posed else None)

    for ids in ((3, 5, 9), (3, 5, 4_000_000)):
        array = CameraArray({ids[0]: cam(ids[0]), ids[1]: cam(ids[1], posed=False), ids[2]: cam(ids[2])})
        img = ImagePoints(pd.DataFrame({"sync_index": 0, "cam_id": [ids[0], ids[1], ids[2], ids[2], 77], "object_id": 0,
                                        "keypoint_id": [0, 0, 0, 1, 0], "img_loc_x": [1.0, 2.0, 3.0, 4.0, 5.0], "img_loc_y": 0.0}))
        world = WorldPoints(pd.DataFrame({"sync_index": 0, "object_id": 0, "keypoint_id": [0, 1], "x_coord": 0.0, "y_coord": 0.0, "z_coord": 0.0}))
        vol = CaptureVolume(array, img, world)
        mask, cam_idx, uv, obj = vol._matched_arrays()
        assert mask.tolist() == [True, False, True, True, False]
        assert cam_idx.tolist() == [0, 1, 1] and obj.tolist() == [0, 0, 1] and uv[:, 0].tolist() == [1.0, 3.0, 4.0]
        assert cam_idx.dtype == np.int32 and obj.dtype == np.int32


def test_row_subsets_column_by_column_equal_the_frame_s_own():
    """ImagePoints.take / WorldPoints.take build the subset column by column (round 4): same rows, columns, order, dtypes and a fresh 0..n-1 index as
    ``df[mask].reset_index(drop=True)`` / ``df.iloc[idx].reset_index(drop=True)``, NaN columns included."""
    import pandas as pd

    from caliscope_amd.point_data import ImagePoints, WorldPoints

    rng = np.random.default_rng(5)
    n = 500
    img = ImagePoints(pd.DataFrame({"sync_index": rng.integers(0, 40, n), "cam_id": rng.integers(0, 6, n), "object_id": rng.integers(0, 3, n),
                                    "keypoint_id": rng.integers(0, 20, n), "img_loc_x": rng.random(n) * 640, "img_loc_y": rng.random(n) * 480,
                                    "frame_time": np.where(rng.random(n) < 0.3, np.nan, rng.random(n))}))
    mask = rng.random(n) < 0.6
    idx = rng.permutation(n)[:123]
    for rows in (mask, idx, np.zeros(n, dtype=bool), np.ones(n, dtype=bool)):
        got = img.take(rows)
        want = img._df[rows].reset_index(drop=True) if rows.dtype == bool else img._df.iloc[rows].reset_index(drop=True)
        pd.testing.assert_frame_equal(got._df, want)
        assert len(got) == len(want) and set(got.arrays()) == set(img.arrays())
        assert all(np.array_equal(got.arrays()[c], want[c].to_numpy()) for c in got.arrays())
    world = WorldPoints(pd.DataFrame({"sync_index": np.r_[np.full(5, -1), rng.integers(3, 30, 95)], "object_id": rng.integers(0, 3, 100), "keypoint_id": np.arange(100),
                                      "x_coord": rng.random(100), "y_coord": rng.random(100), "z_coord": rng.random(100)}))
    keep = rng.random(100) < 0.5
    sub = world.take(keep)
    pd.testing.assert_frame_equal(sub._df, world._df[keep].reset_index(drop=True))
    moving = sub._df["sync_index"].to_numpy()
    moving = moving[moving != -1]
    assert (sub.min_index, sub.max_index) == (int(moving.min()), int(moving.max()))
    assert np.array_equal(sub.points, world.points[keep])


def test_matched_arrays_are_kept_per_posed_camera_set():
    """CaptureVolume._matched_arrays marshals once per volume (round 4) — and again when the set of posed cameras changes under it; the arrays it
    hands out are read-only (they are shared between calls), the volume's own observation -> point map stays writable."""
    import pandas as pd
    import pytest

    from caliscope_amd.cameras import CameraArray, CameraData
    from caliscope_amd.capture_volume import CaptureVolume
    from caliscope_amd.point_data import ImagePoints, WorldPoints

    K = np.array([[800.0, 0, 320], [0, 800.0, 240], [0, 0, 1]])

    def cam(cid, posed=True):
        return CameraData(cam_id=cid, size=(640, 480), matrix=K.copy(), distortions=np.zeros(5), rotation=np.eye(3) if posed else None,
                          translation=np.array([0.0, 0.0, 2.0]) if posed else None)

    array = CameraArray({0: cam(0), 1: cam(1), 2: cam(2, posed=False)})
    img = ImagePoints(pd.DataFrame({"sync_index": 0, "cam_id": [0, 1, 2, 0], "object_id": 0, "keypoint_id": [0, 0, 0, 1], "img_loc_x": [1.0, 2.0, 3.0, 4.0], "img_loc_y": 0.5}))
    world = WorldPoints(pd.DataFrame({"sync_index": 0, "object_id": 0, "keypoint_id": [0, 1], "x_coord": 0.0, "y_coord": 0.0, "z_coord": 0.0}))
    vol = CaptureVolume(array, img, world)
    first = vol._matched_arrays()
    assert vol._matched_arrays() is first and first[0].tolist() == [True, True, False, True]
    with pytest.raises(ValueError):
        first[2][0, 0] = 9.0
    vol.img_to_obj_map[0] = vol.img_to_obj_map[0]  # (still writable)
    array.cameras[2].rotation, array.cameras[2].translation = np.eye(3), np.array([0.0, 0.0, 2.0])  # the third camera gets a pose
    again = vol._matched_arrays()
    assert again is not first and again[0].tolist() == [True, True, True, True] and again[1].tolist() == [0, 1, 2, 0]
    assert again[2].shape == (4, 2) and again[2][:, 0].tolist() == [1.0, 2.0, 3.0, 4.0] and again[3].dtype == np.int32


def test_observation_to_point_map_by_table_equals_the_merge():
    """CaptureVolume._compute_img_to_obj_map: the dense-table lookup (round 4) against the reference's left merge — duplicate world keys (the last row
    wins), observations without a world point, static objects looked up at STATIC_SYNC_INDEX, negative and offset keys; widely spread keys take the merge."""
    import pandas as pd

    from caliscope_amd.cameras import CameraArray, CameraData
    from caliscope_amd.capture_volume import CaptureVolume
    from caliscope_amd.constraints import ConstraintSet
    from caliscope_amd.point_data import STATIC_SYNC_INDEX, ImagePoints, WorldPoints

    K = np.array([[800.0, 0, 320], [0, 800.0, 240], [0, 0, 1]])
    array = CameraArray({c: CameraData(cam_id=c, size=(640, 480), matrix=K.copy(), distortions=np.zeros(5), rotation=np.eye(3), translation=np.array([0.0, 0.0, 2.0]))
                         for c in range(3)})
    rng = np.random.default_rng(9)
    for spread, static in ((1, False), (1, True), (10_000_000, False)):
        nw, ni = 400, 3000
        wsync = rng.integers(5, 60, nw) * spread
        wobj, wkp = rng.integers(2, 5, nw), rng.integers(0, 12, nw)
        if static:
            wsync = np.where(wobj == 4, STATIC_SYNC_INDEX, wsync)
        world = WorldPoints(pd.DataFrame({"sync_index": wsync, "object_id": wobj, "keypoint_id": wkp, "x_coord": rng.random(nw), "y_coord": 0.0, "z_coord": 1.0}))
        img = ImagePoints(pd.DataFrame({"sync_index": rng.integers(3, 64, ni) * spread, "cam_id": rng.integers(0, 3, ni), "object_id": rng.integers(1, 6, ni),
                                        "keypoint_id": rng.integers(0, 14, ni), "img_loc_x": rng.random(ni), "img_loc_y": rng.random(ni)}))
        con = ConstraintSet(distances=(), static_object_ids=frozenset({4})) if static else None
        vol = CaptureVolume(array, img, world, con)
        by_merge = vol._img_to_obj_map_by_merge(con.static_object_ids if con else frozenset())
        assert np.array_equal(vol.img_to_obj_map, by_merge) and vol.img_to_obj_map.dtype == np.int32
        assert (by_merge >= 0).sum() > 100 and (by_merge < 0).sum() > 100  # both kinds of row are present
        if static:
            rows = np.flatnonzero((img._df["object_id"].to_numpy() == 4) & (by_merge >= 0))
            assert rows.size and np.all(world._df["sync_index"].to_numpy()[by_merge[rows]] == STATIC_SYNC_INDEX)


def test_camera_copies_are_deep():
    """``CaptureVolume.optimize`` deep-copies the camera array on every call: the field-wise ``CameraData.__deepcopy__`` must give what the generic
    walk gave — equal values, no shared arrays, attributes a caller hung on the object included."""
    from copy import deepcopy

    from caliscope_amd.cameras import CameraArray, CameraData

    cam = CameraData.from_intrinsics(3, (640, 480), 500.0, distortions=[0.1, 0.01, 0.0, 0.0, 0.0])
    cam.rotation, cam.translation, cam.error, cam.fisheye = np.eye(3), np.array([0.1, 0.2, 0.3]), 0.25, False
    cam.note = {"tags": ["a"]}
    arr = CameraArray({3: cam, 4: CameraData(cam_id=4, size=(320, 240))})
    twin = deepcopy(arr)
    a, b = arr.cameras[3], twin.cameras[3]
    assert a is not b  # (dataclass equality would compare arrays elementwise: checked field by field)
    for name, value in a.__dict__.items():
        other = b.__dict__[name]
        if isinstance(value, np.ndarray):
            assert np.array_equal(value, other) and not np.shares_memory(value, other), name
        else:
            assert value == other, name
    assert b.note is not a.note and b.note["tags"] is not a.note["tags"] and b.size == (640, 480)
    assert twin.cameras[4].matrix is None and twin.cameras[4].rotation is None and set(twin.cameras) == {3, 4}
    b.translation[0] = 9.0
    b.matrix[0, 0] = 1.0
    assert a.translation[0] == 0.1 and a.matrix[0, 0] == 500.0
