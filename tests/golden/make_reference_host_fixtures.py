"""Outputs of the REFERENCE's own host logic on random tables, stored as fixtures ("outputs of the reference itself run here").

    python tests/golden/make_reference_host_fixtures.py          (build container only: imports /root/reference/src)

What is run is the reference's code, unmodified: ``CaptureVolume.__post_init__`` (-> ``img_to_obj_map``, core/capture_volume.py:119-139),
``_build_constraint_arrays`` (:446-516), ``rigidity_report`` (:532-605), ``unique_sync_indices`` and the index range of ``WorldPoints`` — the
pure pandas / numpy part of the path's callers.  The reference imports ``cv2`` and ``rtoml`` at module level; neither is installed here and
none of the functions above calls them, so two stub modules satisfy the imports (a ``cv2`` whose every attribute raises, an ``rtoml`` over tomli).
Nothing of the reference is copied: the fixtures hold the random INPUT tables this script made and the arrays the reference returned for them.

Cases: two to five objects (some static), frames with holes, shuffled rows, now and then a DUPLICATE world key, observations without a world
point, distance constraints between random keypoints (some mixing a static with a moving object, some naming keypoints that never appear)
and centroid constraints.  Consumer: tests/test_reference_host_fixtures.py."""
import sys
import tempfile
import types
import warnings
from pathlib import Path

import numpy as np
import pandas as pd

HERE = Path(__file__).parent
OUT = HERE / "reference_host"
N_CASES = 10


def _stub_modules():
    cv2 = types.ModuleType("cv2")

    def _missing(name):
        raise AttributeError(f"cv2 stub: {name} (the fixture generator must not reach OpenCV)")

    cv2.__getattr__ = _missing
    import tomli

    rtoml = types.ModuleType("rtoml")
    rtoml.load = lambda f: tomli.loads(f.read() if hasattr(f, "read") else Path(f).read_text())
    rtoml.loads = tomli.loads
    sys.modules.setdefault("cv2", cv2)
    sys.modules.setdefault("rtoml", rtoml)


def random_tables(seed):
    """(world rows, image rows, distance constraints, centroid constraints, static object ids) — plain lists and tuples."""
    rng = np.random.default_rng(1000 + seed)
    static_index = -1  # STATIC_SYNC_INDEX of the reference (core/point_data.py), asserted against the import below
    n_obj = int(rng.integers(2, 6))
    static = sorted(o for o in range(n_obj) if rng.random() < 0.3)
    frames = sorted(rng.choice(60, size=int(rng.integers(3, 14)), replace=False).tolist())
    world, img = [], []
    for o in range(n_obj):
        for k in range(4):
            for si in ([static_index] if o in static else frames):
                if rng.random() < 0.2:
                    continue
                for _ in range(2 if rng.random() < 0.05 else 1):
                    world.append((si, o, k, float(rng.normal()), float(rng.normal()), float(rng.normal() + 4.0), float("nan") if o in static else si * 0.1))
        for si in frames:
            for k in range(4):
                for cam in (0, 1):
                    if rng.random() < 0.85:
                        img.append((si, cam, o, k, float(200 + rng.normal(0, 30)), float(200 + rng.normal(0, 30))))
    world = [world[i] for i in rng.permutation(len(world))]
    img = [img[i] for i in rng.permutation(len(img))]
    dist = [(int(rng.integers(0, n_obj)), int(rng.integers(0, 5)), int(rng.integers(0, n_obj)), int(rng.integers(0, 5)),
             float(rng.uniform(0.1, 2.0)), float(rng.uniform(0.001, 0.01))) for _ in range(int(rng.integers(1, 12)))]
    cent = [(int(a), int(b), float(rng.uniform(0.1, 2.0)), 0.005) for a, b in rng.integers(0, n_obj, size=(int(rng.integers(0, 4)), 2)) if a != b]
    return world, img, dist, cent, static


WORLD_COLS = ["sync_index", "object_id", "keypoint_id", "x_coord", "y_coord", "z_coord", "frame_time"]
IMG_COLS = ["sync_index", "cam_id", "object_id", "keypoint_id", "img_loc_x", "img_loc_y"]


def main():
    _stub_modules()
    sys.path.insert(0, "/root/reference/src")
    from caliscope.cameras.camera_array import CameraArray, CameraData
    from caliscope.core.capture_volume import CaptureVolume
    from caliscope.core.constraints import CentroidDistanceConstraint, ConstraintSet, DistanceConstraint
    from caliscope.core.point_data import STATIC_SYNC_INDEX, ImagePoints, WorldPoints

    assert STATIC_SYNC_INDEX == -1
    OUT.mkdir(exist_ok=True)
    K = np.array([[400.0, 0.0, 200.0], [0.0, 400.0, 200.0], [0.0, 0.0, 1.0]])
    for case in range(N_CASES):
        world, img, dist, cent, static = random_tables(case)
        wdf = pd.DataFrame(world, columns=WORLD_COLS).astype({"sync_index": "int64", "object_id": "int64", "keypoint_id": "int64"})
        idf = pd.DataFrame(img, columns=IMG_COLS).astype({c: "int64" for c in IMG_COLS[:4]})
        cams = CameraArray({c: CameraData(cam_id=c, size=(400, 400), matrix=K.copy(), distortions=np.zeros(5), rotation=np.eye(3),
                                          translation=np.array([0.1 * c, 0.0, 0.0])) for c in (0, 1)})
        cs = ConstraintSet(tuple(DistanceConstraint(*d) for d in dist), frozenset(static), centroid_distances=tuple(CentroidDistanceConstraint(*c) for c in cent))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            wp = WorldPoints(wdf)
            vol = CaptureVolume(cams, ImagePoints(idf), wp, cs)
        wtab, itab = vol.world_points.df, vol.image_points.df  # (the tables as the reference keeps them: row order is what the map refers to)
        arrays = vol._build_constraint_arrays()
        rep = vol.rigidity_report()
        viol = np.array([[v.object_id_a, v.keypoint_id_a, v.object_id_b, v.keypoint_id_b, v.sync_index, 1 if v.kind == "centroid" else 0] for v in rep.violations],
                        dtype=np.int64).reshape(-1, 6)
        out = dict(
            world=wtab[WORLD_COLS].to_numpy(dtype=np.float64), image=itab[IMG_COLS].to_numpy(dtype=np.float64),
            distances=np.array(dist, dtype=np.float64).reshape(-1, 6), centroids=np.array(cent, dtype=np.float64).reshape(-1, 4),
            static_ids=np.array(static, dtype=np.int64),
            img_to_obj_map=np.asarray(vol.img_to_obj_map, dtype=np.int64),
            has_rows=np.array(arrays is not None),
            groups_a=(arrays[0] if arrays else np.zeros((0, 4), np.int32)), groups_b=(arrays[1] if arrays else np.zeros((0, 4), np.int32)),
            row_distance=(arrays[2] if arrays else np.zeros(0)), row_sigma=(arrays[3] if arrays else np.zeros(0)),
            violations=viol, violation_expected=np.array([v.expected for v in rep.violations]), violation_actual=np.array([v.actual for v in rep.violations]),
            rmse_mm=np.array(rep.rmse_mm), max_violation_mm=np.array(rep.max_violation_mm),
            unique_sync_indices=np.asarray(vol.unique_sync_indices, dtype=np.int64),
            world_min_max=np.array([wp.min_index if wp.min_index is not None else 0, wp.max_index if wp.max_index is not None else 0], dtype=np.int64),
        )
        np.savez_compressed(OUT / f"case_{case:02d}.npz", **out)
        print(f"case {case}: {len(wtab)} world rows ({int(wtab.duplicated(subset=WORLD_COLS[:3]).sum())} duplicate keys), {len(itab)} observations "
              f"({int((out['img_to_obj_map'] < 0).sum())} unmatched), {len(dist)} + {len(cent)} constraints -> {len(out['row_distance'])} rows, static {static}")


if __name__ == "__main__":
    main()
