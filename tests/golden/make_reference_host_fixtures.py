"""Outputs of the REFERENCE's own host logic on random tables, stored as fixtures ("outputs of the reference itself run here").

    python tests/golden/make_reference_host_fixtures.py          (build container only: imports /root/reference/src)

What is run is the reference's code, unmodified: ``CaptureVolume.__post_init__`` (-> ``img_to_obj_map``, core/capture_volume.py:119-139),
``_build_constraint_arrays`` (:446-516), ``rigidity_report`` (:532-605), ``unique_sync_indices`` and the index range of ``WorldPoints`` — the
pure pandas / numpy part of the path's callers.  The reference imports ``cv2`` and ``rtoml`` at module level; neither is installed here and
none of the functions above calls them, so two stub modules satisfy the imports (a ``cv2`` whose every attribute raises, an ``rtoml`` over tomli).
Nothing of the reference is copied: the fixtures hold the random INPUT tables this script made and the arrays the reference returned for them.

Cases: two to five objects (some static), frames with holes, shuffled rows, now and then a DUPLICATE world key, observations without a world
point, distance constraints between random keypoints (some mixing a static with a moving object, some naming keypoints that never appear)
and centroid constraints.  Consumer: tests/test_reference_host_fixtures.py.

Second family (``bundle_*.npz``): the reference's ``BundleParameterization`` (core/bundle_parameterization.py) on random camera arrays — sparse
camera ids, unposed and ignored cameras, fisheye cameras beside pinhole ones, intrinsics locked or free: block table, offsets, ``pack``,
``bounds``, ``unpack_into`` of a perturbed vector, ``bound_warnings``, ``intrinsic_estimates``, ``trial_projection_inputs``, the non-zero
pattern of ``sparsity``.  ``CameraData.extrinsics_to_vector / extrinsics_from_vector`` (cameras/camera_array.py:115-133) call ``cv2.Rodrigues``;
for this family the stub's ``Rodrigues`` is scipy's ``Rotation`` (rotation vector <-> matrix: the same map to ~1e-16, NOT OpenCV's code), so the
rotation entries pin the LAYOUT of the vector and are compared at 1e-12, everything else exactly.

Third family (``tables_*.npz``): the reference's point tables (core/point_data.py) on random tracks with holes: validation (column set and order,
optional columns), ``fill_gaps`` at three gap sizes for image and world points, ``WorldPoints.smooth``, the CSV round trip, ``filter_to_objects``.

Fourth family (``interop_*.npz``): on-disk formats in the direction a user migrates — directories written by THIS package's ``CaptureVolume.save()``
(camera_array.toml, image_points.csv, world_points.csv, constraints.toml) read by the reference's ``CaptureVolume.load()``; the fixture holds the
file texts and every field the reference's loaders returned (rotation vectors through the same scipy ``Rodrigues`` as above).

Fifth family (``compilers_*.npz``): the constraint compilers ``ConstraintSet.from_marker_set`` and ``from_chessboard`` on random marker sets and boards
(see ``compiler_cases``).

Sixth family (``filter_*.npz``): the outlier filters between the solver passes, on an injected report (see ``filter_cases``).

Seventh family (``report_*.npz``): the bookkeeping of the reprojection report around injected pixel errors (see ``report_cases``).

Eighth family (``seam_*.npz``): the seam itself — what the reference's ``optimize()`` hands to ``least_squares`` and what it makes of the result
(see ``seam_cases``).

Ninth family (``conrows_*.npz``): the constraint rows of the reference's own ``joint_residuals`` / ``joint_jacobian`` (see ``constraint_row_cases``).

Tenth family (``driver_*.npz``): the stage driver ``calibrate_extrinsics`` with its three heavy calls scripted (see ``driver_cases``).

Eleventh family (``dlt_*.npz``): the reference's batched SVD triangulation, a plain numpy function (see ``dlt_cases``).

Twelfth family (``triangulate_*.npz``): ``ImagePoints.triangulate`` for cameras without lens distortion (see ``triangulate_cases``).

Thirteenth family (``remap_*.npz``): ``ConstraintSet.remap_image_points`` with arbitrary, also chained, remaps (see ``remap_cases``).

Fourteenth (``camtoml_00.npz``): ``CameraArray.from_toml`` on camera files of the reference's side (see ``camera_toml_cases``)."""
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

    def rodrigues(a):  # (bundle_* family only: scipy's rotation-vector map in the place of OpenCV's, see the docstring)
        from scipy.spatial.transform import Rotation

        a = np.asarray(a, dtype=np.float64)
        if a.shape == (3, 3):
            return Rotation.from_matrix(a).as_rotvec().reshape(3, 1), None
        return Rotation.from_rotvec(a.reshape(3)).as_matrix(), None

    cv2.Rodrigues = rodrigues
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


def random_camera_array(seed):
    """Plain description of a camera array: list of dicts (cam_id, size, K, dist, fisheye, ignore, rvec / t or None)."""
    rng = np.random.default_rng(5000 + seed)
    ids = sorted(rng.choice(40, size=int(rng.integers(3, 8)), replace=False).tolist())
    cams = []
    for c in ids:
        fisheye = bool(rng.random() < 0.25)
        posed = bool(rng.random() < 0.85)
        f = float(rng.uniform(300, 900))
        cams.append(dict(cam_id=int(c), size=(int(rng.integers(320, 1920)), int(rng.integers(240, 1080))),
                         K=[[f, 0.0, float(rng.uniform(100, 600))], [0.0, f * float(rng.uniform(0.98, 1.02)), float(rng.uniform(100, 400))], [0.0, 0.0, 1.0]],
                         dist=rng.normal(0, 0.05, 4 if fisheye else 5).tolist(), fisheye=fisheye, ignore=bool(rng.random() < 0.15),
                         rvec=rng.normal(0, 0.8, 3).tolist() if posed else None, t=rng.normal(0, 1.0, 3).tolist() if posed else None))
    if sum(1 for c in cams if c["rvec"] is not None and not c["ignore"]) < 2:  # at least two cameras to optimise
        for c in cams[:2]:
            c["ignore"] = False
            c["rvec"], c["t"] = rng.normal(0, 0.8, 3).tolist(), rng.normal(0, 1.0, 3).tolist()
    return cams, bool(seed % 2), int(rng.integers(5, 30))


def bundle_cases():
    from scipy.spatial.transform import Rotation

    from caliscope.cameras.camera_array import CameraArray, CameraData
    from caliscope.core.bundle_parameterization import BundleParameterization

    for case in range(8):
        desc, refine, n_points = random_camera_array(case)
        rng = np.random.default_rng(9000 + case)

        def build():
            return CameraArray({d["cam_id"]: CameraData(
                cam_id=d["cam_id"], size=tuple(d["size"]), matrix=np.array(d["K"]), distortions=np.array(d["dist"]), fisheye=d["fisheye"], ignore=d["ignore"],
                rotation=None if d["rvec"] is None else Rotation.from_rotvec(d["rvec"]).as_matrix(),
                translation=None if d["t"] is None else np.array(d["t"])) for d in desc})

        arr = build()
        par = BundleParameterization.from_camera_array(arr, n_points, refine_intrinsics=refine)
        pts = rng.normal(0, 1, (n_points, 3))
        x0 = par.pack(arr, pts)
        lb, ub = par.bounds()
        x1 = x0 + rng.normal(0, 0.01, x0.size)
        edge = []  # some free intrinsics onto / next to their bounds for bound_warnings
        for i, b in enumerate(par.blocks):
            if b.free_intrinsics:
                off = par.camera_param_offsets[i] + 6
                x1[off:off + 3] = [[0.5, 0.502, 1.0, 1.99, 2.0][int(rng.integers(0, 5))], [-1.0, -0.995, 0.3, 0.992, 1.0][int(rng.integers(0, 5))],
                                   [-2.0, -1.991, 0.0, 1.995, 2.0][int(rng.integers(0, 5))]]
                edge.append(i)
        warns = par.bound_warnings(x1)
        arr2 = build()
        pts_back = par.unpack_into(arr2, x1.copy())
        est = par.intrinsic_estimates(arr2)
        n_obs = int(rng.integers(20, 60))
        cam_idx = rng.integers(0, len(par.blocks), n_obs)
        obj_idx = rng.integers(0, n_points, n_obs)
        ga, gb = rng.integers(0, n_points, (4, 4)), rng.integers(0, n_points, (4, 4))
        sp = par.sparsity(cam_idx, obj_idx, 4, ga, gb).tocoo()
        trial = [par.trial_projection_inputs(x1, i) for i in range(len(par.blocks))]
        blocks = np.array([[b.cam_id, int(b.free_intrinsics), b.fx_initial, b.fy_initial, b.cx, b.cy, int(b.fisheye), b.k1_initial, b.k2_initial, len(b.dist_fixed),
                            *(list(b.dist_fixed) + [0.0] * (4 - len(b.dist_fixed)))] for b in par.blocks], dtype=np.float64)
        opt_ids = [b.cam_id for b in par.blocks]
        # the flattening this package hands to the C ABI, computed from the REFERENCE'S OWN parameterization object (the seam is
        # duck-typed — the one-line patch passes the reference's class, not this package's)
        sys.path.insert(0, str(HERE.parent.parent))
        from caliscope_amd.bundle_parameterization import device_tables, n_params_of

        tabs = device_tables(par)
        np.savez_compressed(
            OUT / f"bundle_{case:02d}.npz", n_params_of=np.array(n_params_of(par)), **{f"device_{k}": np.asarray(v) for k, v in tabs.items()},
            cam_ids=np.array([d["cam_id"] for d in desc]), sizes=np.array([d["size"] for d in desc]), K=np.array([d["K"] for d in desc]),
            dist=np.array([d["dist"] + [np.nan] * (5 - len(d["dist"])) for d in desc]), fisheye=np.array([d["fisheye"] for d in desc]),
            ignore=np.array([d["ignore"] for d in desc]), posed=np.array([d["rvec"] is not None for d in desc]),
            rvec=np.array([d["rvec"] if d["rvec"] is not None else [np.nan] * 3 for d in desc]), t=np.array([d["t"] if d["t"] is not None else [np.nan] * 3 for d in desc]),
            refine=np.array(refine), n_points=np.array(n_points), points=pts,
            blocks=blocks, offsets=np.array(par.camera_param_offsets), n_camera_params=np.array(par.n_camera_params), x0=x0, lb=lb, ub=ub, x1=x1,
            warnings=np.array([[w.cam_id, {"f": 0, "k1": 1, "k2": 2}[w.parameter], {"lower": 0, "upper": 1}[w.bound], w.value] for w in warns], dtype=np.float64).reshape(-1, 4),
            unpacked_R=np.array([arr2.cameras[c].rotation for c in opt_ids]), unpacked_t=np.array([np.ravel(arr2.cameras[c].translation) for c in opt_ids]),
            unpacked_K=np.array([arr2.cameras[c].matrix for c in opt_ids]),
            unpacked_dist=np.array([list(np.ravel(arr2.cameras[c].distortions)) + [np.nan] * (5 - np.size(arr2.cameras[c].distortions)) for c in opt_ids]),
            points_back=np.asarray(pts_back),
            estimates=np.array([[e.cam_id, e.f_recovered, e.k1_recovered, e.k2_recovered, e.f_initial, e.k1_initial, e.k2_initial] for e in est], dtype=np.float64).reshape(-1, 7),
            cam_idx=cam_idx, obj_idx=obj_idx, groups_a=ga, groups_b=gb, sparsity_shape=np.array(sp.shape), sparsity_rows=sp.row[sp.data != 0], sparsity_cols=sp.col[sp.data != 0],
            trial_rvec=np.array([t[0] for t in trial]), trial_tvec=np.array([t[1] for t in trial]), trial_K=np.array([t[2] for t in trial]),
            trial_dist=np.array([list(t[3]) + [np.nan] * (5 - len(t[3])) for t in trial]),
        )
        print(f"bundle {case}: cameras {[d['cam_id'] for d in desc]} -> optimised {opt_ids}, refine {refine}, {par.n_camera_params} camera parameters, "
              f"{len(warns)} bound warnings, {len(est)} estimates, sparsity {sp.shape} with {int((sp.data != 0).sum())} non-zeros")


def random_tracks(seed):
    """(image rows, world rows) with tracks that have holes of one to six frames and single stray samples."""
    rng = np.random.default_rng(7000 + seed)
    frames = np.arange(int(rng.integers(25, 60)))
    img, world = [], []
    for obj in range(int(rng.integers(1, 3))):
        for kp in range(int(rng.integers(2, 6))):
            present = np.ones(frames.size, dtype=bool)
            for _ in range(int(rng.integers(1, 5))):
                a = int(rng.integers(0, frames.size - 1))
                present[a:a + int(rng.integers(1, 7))] = False
            track = np.cumsum(rng.normal(0, 0.02, (frames.size, 3)), axis=0) + rng.normal(0, 1, 3)
            for f in frames[present]:
                world.append((int(f), obj, kp, *track[f].tolist(), f / 30.0))
            for cam in range(int(rng.integers(1, 4))):
                seen = present & (rng.random(frames.size) < 0.9)
                uv = np.cumsum(rng.normal(0, 1.5, (frames.size, 2)), axis=0) + rng.uniform(100, 500, 2)
                for f in frames[seen]:
                    img.append((int(f), cam, obj, kp, *uv[f].tolist()))
    return [img[i] for i in rng.permutation(len(img))], [world[i] for i in rng.permutation(len(world))]


def table_cases():
    from caliscope.core.point_data import ImagePoints, WorldPoints

    for case in range(6):
        img, world = random_tracks(case)
        idf = pd.DataFrame(img, columns=IMG_COLS).astype({c: "int64" for c in IMG_COLS[:4]})
        wdf = pd.DataFrame(world, columns=WORLD_COLS).astype({"sync_index": "int64", "object_id": "int64", "keypoint_id": "int64"})
        ip, wp = ImagePoints(idf), WorldPoints(wdf)
        out = dict(image=idf.to_numpy(dtype=np.float64), world=wdf.to_numpy(dtype=np.float64))
        out["image_columns"] = np.array(list(ip.df.columns))
        out["world_columns"] = np.array(list(wp.df.columns))
        out["image_validated"] = ip.df.to_numpy(dtype=np.float64)
        out["world_validated"] = wp.df.to_numpy(dtype=np.float64)
        for gap in (1, 3, 5):
            fi, fw = ip.fill_gaps(gap).df, wp.fill_gaps(gap).df  # (the reference's filled frames carry its helper column `gap_size` along)
            out[f"image_filled_{gap}"], out[f"image_filled_{gap}_columns"] = fi.to_numpy(dtype=np.float64), np.array(list(fi.columns))
            out[f"world_filled_{gap}"], out[f"world_filled_{gap}_columns"] = fw.to_numpy(dtype=np.float64), np.array(list(fw.columns))
        out["world_smoothed"] = wp.smooth(fps=30.0, cutoff_freq=6.0, order=2).df.to_numpy(dtype=np.float64)
        out["world_smoothed_o3"] = wp.smooth(fps=60.0, cutoff_freq=4.0, order=3).df.to_numpy(dtype=np.float64)
        # (smooth() of a gap-filled table raises inside the reference — its filled frame keeps a repeated index —, so that chain is not a fixture)
        with tempfile.TemporaryDirectory() as tmp:
            ip.to_csv(Path(tmp) / "xy.csv")
            wp.to_csv(Path(tmp) / "xyz.csv")
            out["image_csv"] = np.array((Path(tmp) / "xy.csv").read_text())
            out["world_csv"] = np.array((Path(tmp) / "xyz.csv").read_text())
            out["image_csv_back"] = ImagePoints.from_csv(Path(tmp) / "xy.csv").df.to_numpy(dtype=np.float64)
            out["world_csv_back"] = WorldPoints.from_csv(Path(tmp) / "xyz.csv").df.to_numpy(dtype=np.float64)
        keep = sorted(set(int(o) for o in idf["object_id"]))[:1]
        out["kept_objects"] = np.array(keep)
        out["image_filtered"] = ip.filter_to_objects(keep).df.to_numpy(dtype=np.float64)
        np.savez_compressed(OUT / f"tables_{case:02d}.npz", **out)
        print(f"tables {case}: {len(idf)} observations -> {len(out['image_filled_3'])} filled (gap 3), {len(wdf)} world rows -> {len(out['world_filled_3'])}; "
              f"columns {list(ip.df.columns)} / {list(wp.df.columns)}")


def interop_cases():
    """A directory written by THIS package's ``CaptureVolume.save()`` and read by the REFERENCE's ``CaptureVolume.load()`` (core/capture_volume.py:253-267:
    ``CameraArray.from_toml``, the two CSV readers, ``ConstraintSet.from_toml``): the file texts and what the reference made of them."""
    sys.path.insert(0, str(HERE.parent.parent))
    from scipy.spatial.transform import Rotation

    import caliscope_amd.cameras as my_cam
    import caliscope_amd.capture_volume as my_cv
    import caliscope_amd.constraints as my_con
    import caliscope_amd.point_data as my_pd
    from caliscope.core.capture_volume import CaptureVolume as RefVolume

    for case in range(6):
        desc, _, _ = random_camera_array(100 + case)
        rng = np.random.default_rng(11000 + case)
        for d in desc[:2]:  # the volume needs posed cameras 0 / 1 of the random tables: rename the first two, posed and active
            d["ignore"] = False
            if d["rvec"] is None:
                d["rvec"], d["t"] = rng.normal(0, 0.5, 3).tolist(), rng.normal(0, 1, 3).tolist()
        desc[0]["cam_id"], desc[1]["cam_id"] = 0, 1
        extras = {d["cam_id"]: dict(rotation_count=int(rng.integers(0, 4)), error=[None, float(rng.uniform(0.1, 2))][int(rng.integers(0, 2))],
                                    exposure=[None, int(rng.integers(-8, 0))][int(rng.integers(0, 2))], grid_count=[None, int(rng.integers(5, 60))][int(rng.integers(0, 2))])
                  for d in desc}
        cams = my_cam.CameraArray({d["cam_id"]: my_cam.CameraData(
            cam_id=d["cam_id"], size=tuple(d["size"]), matrix=np.array(d["K"]), distortions=np.array(d["dist"]), fisheye=d["fisheye"], ignore=d["ignore"],
            rotation=None if d["rvec"] is None else Rotation.from_rotvec(d["rvec"]).as_matrix(), translation=None if d["t"] is None else np.array(d["t"]),
            **extras[d["cam_id"]]) for d in desc})
        world, img, dist, cent, static = random_tables(200 + case)
        wdf = pd.DataFrame(world, columns=WORLD_COLS).astype({"sync_index": "int64", "object_id": "int64", "keypoint_id": "int64"})
        idf = pd.DataFrame(img, columns=IMG_COLS).astype({c: "int64" for c in IMG_COLS[:4]})
        remaps = tuple(my_con.PointRemap(int(rng.integers(0, 3)), int(rng.integers(0, 4)), int(rng.integers(3, 6)), int(rng.integers(0, 4)),
                                         float(rng.normal()), float(rng.normal()), 0.0) for _ in range(int(rng.integers(0, 3))))
        cs = my_con.ConstraintSet(tuple(my_con.DistanceConstraint(*d) for d in dist), frozenset(static),
                                  centroid_distances=tuple(my_con.CentroidDistanceConstraint(*c) for c in cent), point_remaps=remaps,
                                  back_face_thickness_m=[None, 0.004][case % 2]) if case != 3 else None  # (one volume without constraints: no file)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mine = my_cv.CaptureVolume(cams, my_pd.ImagePoints(idf), my_pd.WorldPoints(wdf), cs)
            with tempfile.TemporaryDirectory() as tmp:
                mine.save(tmp)
                files = {f.name: f.read_text() for f in sorted(Path(tmp).iterdir())}
                ref = RefVolume.load(tmp)
        ids = sorted(ref.camera_array.cameras)
        nan3, nan5 = [np.nan] * 3, [np.nan] * 5

        def opt(v):
            return np.nan if v is None else float(v)

        out = dict(
            file_names=np.array(list(files)), file_texts=np.array(list(files.values())),
            cam_ids=np.array(ids), sizes=np.array([ref.camera_array.cameras[c].size for c in ids]),
            rotation_count=np.array([ref.camera_array.cameras[c].rotation_count for c in ids]),
            error=np.array([opt(ref.camera_array.cameras[c].error) for c in ids]), exposure=np.array([opt(ref.camera_array.cameras[c].exposure) for c in ids]),
            grid_count=np.array([opt(ref.camera_array.cameras[c].grid_count) for c in ids]),
            ignore=np.array([bool(ref.camera_array.cameras[c].ignore) for c in ids]), fisheye=np.array([bool(ref.camera_array.cameras[c].fisheye) for c in ids]),
            K=np.array([ref.camera_array.cameras[c].matrix for c in ids]),
            dist=np.array([list(np.ravel(ref.camera_array.cameras[c].distortions)) + [np.nan] * (5 - np.size(ref.camera_array.cameras[c].distortions)) for c in ids]),
            posed=np.array([ref.camera_array.cameras[c].rotation is not None for c in ids]),
            R=np.array([ref.camera_array.cameras[c].rotation if ref.camera_array.cameras[c].rotation is not None else np.full((3, 3), np.nan) for c in ids]),
            t=np.array([np.ravel(ref.camera_array.cameras[c].translation) if ref.camera_array.cameras[c].translation is not None else nan3 for c in ids]),
            image=ref.image_points.df.to_numpy(dtype=np.float64), image_columns=np.array(list(ref.image_points.df.columns)),
            world=ref.world_points.df.to_numpy(dtype=np.float64), world_columns=np.array(list(ref.world_points.df.columns)),
            img_to_obj_map=np.asarray(ref.img_to_obj_map, dtype=np.int64), has_constraints=np.array(ref.constraints is not None),
        )
        if ref.constraints is not None:
            rc = ref.constraints
            out.update(
                distances=np.array([[d.object_id_a, d.keypoint_id_a, d.object_id_b, d.keypoint_id_b, d.distance, d.sigma] for d in rc.distances], dtype=np.float64).reshape(-1, 6),
                centroids=np.array([[c.object_id_a, c.object_id_b, c.distance, c.sigma] for c in rc.centroid_distances], dtype=np.float64).reshape(-1, 4),
                static_ids=np.array(sorted(rc.static_object_ids), dtype=np.int64),
                remaps=np.array([[r.object_id_from, r.keypoint_id_from, r.object_id_to, r.keypoint_id_to, r.obj_loc_x, r.obj_loc_y, r.obj_loc_z] for r in rc.point_remaps],
                                dtype=np.float64).reshape(-1, 7),
                thickness=np.array(np.nan if rc.back_face_thickness_m is None else rc.back_face_thickness_m))
        np.savez_compressed(OUT / f"interop_{case:02d}.npz", **out)
        print(f"interop {case}: files {list(files)}, cameras {ids} ({int(out['posed'].sum())} posed), {len(out['image'])} observations, "
              f"{'no constraints' if ref.constraints is None else str(len(out['distances'])) + ' + ' + str(len(out['centroids'])) + ' constraints, ' + str(len(out['remaps'])) + ' remaps'}")


def _rows_sorted(a):
    return a[np.lexsort(a.T[::-1])] if len(a) else a


def _constraint_set_arrays(cs):
    """The set as arrays, rows sorted (the order of the rows of a constraint set carries no meaning; the reference and this package list a grid's
    diagonals in different orders)."""
    out = _constraint_set_arrays_unsorted(cs)
    return {k: (_rows_sorted(v) if v.ndim == 2 else v) for k, v in out.items()}


def _constraint_set_arrays_unsorted(cs):
    return dict(
        distances=np.array([[d.object_id_a, d.keypoint_id_a, d.object_id_b, d.keypoint_id_b, d.distance, d.sigma] for d in cs.distances], dtype=np.float64).reshape(-1, 6),
        centroids=np.array([[c.object_id_a, c.object_id_b, c.distance, c.sigma] for c in cs.centroid_distances], dtype=np.float64).reshape(-1, 4),
        static_ids=np.array(sorted(cs.static_object_ids), dtype=np.int64),
        remaps=np.array([[r.object_id_from, r.keypoint_id_from, r.object_id_to, r.keypoint_id_to, r.obj_loc_x, r.obj_loc_y, r.obj_loc_z] for r in cs.point_remaps],
                        dtype=np.float64).reshape(-1, 7),
        thickness=np.array(np.nan if cs.back_face_thickness_m is None else cs.back_face_thickness_m))


def compiler_cases():
    """``ConstraintSet.from_marker_set`` (core/constraints.py:84-190) on random marker sets — sizes, static markers, centre and corner links, mirror
    pairs of zero and of positive thickness — and ``from_chessboard`` (:397-418).  ``ArucoMarkerSet`` checks its ids against the capacity of an OpenCV
    dictionary: the stub's ``cv2.aruco.getPredefinedDictionary`` answers with 250 entries and nothing else.  The fixture stores what the compilers read
    from the reference's objects (corners, flags, the derived ``is_center`` / ``corner_mapping`` / ``is_zero_thickness``) and the compiled sets; at
    generation time this package's compiler is also run on the reference's OWN objects and must return the same set."""
    sys.path.insert(0, str(HERE.parent.parent))
    import cv2

    cv2.aruco = types.SimpleNamespace(getPredefinedDictionary=lambda d: types.SimpleNamespace(bytesList=[None] * 250))
    import caliscope_amd.constraints as my_con
    from caliscope.core.aruco_marker import ArucoMarker, ArucoMarkerSet, DistanceLink, MirrorPair
    from caliscope.core.chessboard import Chessboard
    from caliscope.core.constraints import ConstraintSet

    for case in range(8):
        rng = np.random.default_rng(13000 + case)
        ids = sorted(rng.choice(40, size=int(rng.integers(2, 8)), replace=False).tolist())
        static = {m for m in ids if rng.random() < 0.3}
        size = {m: float(rng.uniform(0.03, 0.3)) for m in ids}
        pairs = []  # mirror pairs first: the two markers of a pair share their size and are both static or both mobile
        free = list(ids)
        for _ in range(int(rng.integers(0, 3))):
            if len(free) < 2:
                break
            a, b = (int(v) for v in rng.choice(free, size=2, replace=False))
            free.remove(a); free.remove(b)
            size[b] = size[a]
            if (a in static) != (b in static):
                static.discard(a); static.discard(b)
            pairs.append((a, b))
        markers = {m: ArucoMarker(int(m), size[m], static=m in static) for m in ids}
        links, seen = [], set()
        for _ in range(int(rng.integers(0, 6))):
            a, b = (int(v) for v in rng.choice(ids, size=2, replace=False))
            if (a in static) != (b in static):
                continue
            corner = None if rng.random() < 0.5 else (int(rng.integers(0, 4)), int(rng.integers(0, 4)))
            key = frozenset([(a, None if corner is None else corner[0]), (b, None if corner is None else corner[1])])
            if key in seen:
                continue
            seen.add(key)
            links.append(DistanceLink(a, b, float(rng.uniform(0.1, 2.0)), *(corner if corner else (None, None)), sigma_m=[None, 0.003][int(rng.integers(0, 2))]))
        mirrors = [MirrorPair(a, b, int(rng.integers(0, 4)), int(rng.integers(0, 4)), [0.0, 0.006][int(rng.integers(0, 2))], sigma_m=[None, 0.001][int(rng.integers(0, 2))])
                   for a, b in pairs]
        try:
            ms = ArucoMarkerSet(0, markers, tuple(links), tuple(mirrors))
        except ValueError as exc:  # (a random combination the reference's own validation refuses: no mirror pairs then)
            print(f"compilers {case}: marker set refused by the reference ({exc}); mirror pairs dropped")
            mirrors = []
            ms = ArucoMarkerSet(0, markers, tuple(links), ())
        sig, csig = float(rng.uniform(0.001, 0.004)), float(rng.uniform(0.003, 0.008))
        ref = ConstraintSet.from_marker_set(ms, sigma_m=sig, center_sigma_m=csig)
        mine = my_con.ConstraintSet.from_marker_set(ms, sigma_m=sig, center_sigma_m=csig)  # this package's compiler on the reference's own objects
        a, b = _constraint_set_arrays(ref), _constraint_set_arrays(mine)
        same = all(np.array_equal(a[k], b[k], equal_nan=True) for k in a)
        rows, cols = int(rng.integers(3, 9)), int(rng.integers(3, 9))
        board = Chessboard(rows, cols, float(rng.uniform(1.0, 6.0)))
        bsig = float(rng.uniform(0.001, 0.004))
        ref_b = ConstraintSet.from_chessboard(board, sigma_m=bsig)
        same_b = all(np.array_equal(v, _constraint_set_arrays(my_con.ConstraintSet.from_chessboard(board, sigma_m=bsig))[k], equal_nan=True)
                     for k, v in _constraint_set_arrays(ref_b).items())
        # from_charuco (:359-395) reads charuco.board.getChessboardCorners() / .getSquareLength() and charuco.thickness_m: a stand-in with a float32
        # corner grid (interior corners of a cols x rows board, some cases shuffled: the compiler recovers the grid from coordinates, not ids) goes
        # through the REFERENCE's compiler — thin boards and two-sided ones with a substrate (back face = object 1, ties and braces)
        ccols, crows, sq = int(rng.integers(3, 8)), int(rng.integers(3, 8)), float(rng.uniform(0.02, 0.06))
        grid = np.array([[(i + 1) * sq, (j + 1) * sq, 0.0] for j in range(crows) for i in range(ccols)], dtype=np.float32)
        if case % 3 == 0:
            grid = grid[rng.permutation(len(grid))]
        thick = [0.0, 0.005, 0.012][case % 3]
        stand_in = types.SimpleNamespace(board=types.SimpleNamespace(getChessboardCorners=lambda g=grid: g, getSquareLength=lambda q=sq: q), thickness_m=thick)
        ch_sig, ch_tsig = float(rng.uniform(0.001, 0.004)), float(rng.uniform(0.0003, 0.001))
        ref_c = ConstraintSet.from_charuco(stand_in, sigma_m=ch_sig, thickness_sigma_m=ch_tsig)
        same_c = all(np.array_equal(v, _constraint_set_arrays(my_con.ConstraintSet.from_charuco(stand_in, sigma_m=ch_sig, thickness_sigma_m=ch_tsig))[k], equal_nan=True)
                     for k, v in _constraint_set_arrays(ref_c).items())
        same_b = same_b and same_c
        out = {f"set_{k}": v for k, v in a.items()}
        out.update({f"board_{k}": v for k, v in _constraint_set_arrays(ref_b).items()})
        out.update({f"charuco_{k}": v for k, v in _constraint_set_arrays(ref_c).items()})
        out.update(charuco_corners=grid, charuco=np.array([sq, thick, ch_sig, ch_tsig]))
        out.update(
            marker_ids=np.array(ids), marker_static=np.array([m in static for m in ids]), marker_size=np.array([markers[m].size_m for m in ids]),
            marker_corners=np.array([markers[m].corners for m in ids], dtype=np.float64),
            links=np.array([[lk.marker_a, lk.marker_b, lk.distance_m, -1 if lk.corner_a is None else lk.corner_a, -1 if lk.corner_b is None else lk.corner_b,
                             np.nan if lk.sigma_m is None else lk.sigma_m, int(lk.is_center)] for lk in links], dtype=np.float64).reshape(-1, 7),
            mirrors=np.array([[mp.marker_a, mp.marker_b, mp.anchor_corner_a, mp.anchor_corner_b, mp.thickness_m, np.nan if mp.sigma_m is None else mp.sigma_m,
                               int(mp.is_zero_thickness), *np.ravel(mp.corner_mapping)] for mp in mirrors], dtype=np.float64).reshape(-1, 15),
            sigma=np.array([sig, csig]), board=np.array([rows, cols, board.square_size_cm, bsig]), board_points=np.asarray(board.get_object_points(), dtype=np.float64),
            same_on_reference_objects=np.array([same, same_b]))
        np.savez_compressed(OUT / f"compilers_{case:02d}.npz", **out)
        print(f"compilers {case}: markers {ids} (static {sorted(static)}), {len(links)} links, {len(mirrors)} mirror pairs -> {len(a['distances'])} + {len(a['centroids'])} "
              f"constraints, {len(a['remaps'])} remaps; board {rows} x {cols} -> {len(out['board_distances'])}; this package's compilers on the reference's objects: "
              f"{'identical' if same and same_b else 'DIFFERENT'}")


def filter_cases():
    """``filter_by_percentile_error`` / ``filter_by_absolute_error`` (core/capture_volume.py:607-753) with an INJECTED report: the reference computes
    its reprojection report through ``cv2.projectPoints``, which is not available here, and the filters read nothing of it but ``raw_errors`` — so a
    report object of the reference's own class is put in the place of the cached property, with random errors for the matched observations.  What is
    pinned is the logic between the solver passes of ``calibrate_extrinsics``: per-camera / overall percentiles, the safety floor, which
    observations survive, which world points are pruned (static ones re-attached)."""
    from caliscope.cameras.camera_array import CameraArray, CameraData
    from caliscope.core.capture_volume import CaptureVolume
    from caliscope.core.constraints import ConstraintSet
    from caliscope.core.point_data import ImagePoints, WorldPoints
    from caliscope.core.reprojection_report import ReprojectionReport

    K = np.array([[400.0, 0.0, 200.0], [0.0, 400.0, 200.0], [0.0, 0.0, 1.0]])
    for case in range(6):
        world, img, _, _, static = random_tables(300 + case)
        rng = np.random.default_rng(17000 + case)
        wdf = pd.DataFrame(world, columns=WORLD_COLS).astype({"sync_index": "int64", "object_id": "int64", "keypoint_id": "int64"})
        wdf = wdf.drop_duplicates(subset=WORLD_COLS[:3]).reset_index(drop=True)  # (duplicate keys multiply rows in the reference's merges: not what is pinned here)
        idf = pd.DataFrame(img, columns=IMG_COLS).astype({c: "int64" for c in IMG_COLS[:4]})
        cams = CameraArray({c: CameraData(cam_id=c, size=(400, 400), matrix=K.copy(), distortions=np.zeros(5), rotation=np.eye(3),
                                          translation=np.array([0.1 * c, 0.0, 0.0])) for c in (0, 1)})
        cs = ConstraintSet((), frozenset(static)) if static else None
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            vol = CaptureVolume(cams, ImagePoints(idf), WorldPoints(wdf), cs)
        matched = vol.img_to_obj_map >= 0
        idf_kept = vol.image_points.df[matched]
        err = np.abs(rng.normal(0, 1.0, (int(matched.sum()), 2))) * rng.choice([1.0, 1.0, 1.0, 8.0], size=(int(matched.sum()), 1))
        err[idf_kept["cam_id"].to_numpy() == 1] *= 6.0  # camera 1 is the bad one: thresholds taken from camera 0 leave it below the floor alone
        raw = pd.DataFrame({"sync_index": idf_kept["sync_index"].to_numpy(), "cam_id": idf_kept["cam_id"].to_numpy(), "object_id": idf_kept["object_id"].to_numpy(),
                            "keypoint_id": idf_kept["keypoint_id"].to_numpy(), "error_x": err[:, 0], "error_y": err[:, 1], "euclidean_error": np.hypot(err[:, 0], err[:, 1])})
        report = ReprojectionReport(overall_rmse=0.0, by_camera={}, by_point={}, n_unmatched_observations=int((~matched).sum()), unmatched_rate=0.0, unmatched_by_camera={},
                                    raw_errors=raw, n_observations_matched=int(matched.sum()), n_observations_total=len(matched), n_cameras=2, n_points=len(wdf))
        vol.__dict__["reprojection_report"] = report  # (the cached property's slot)
        out = dict(world=vol.world_points.df[WORLD_COLS].to_numpy(dtype=np.float64), image=vol.image_points.df[IMG_COLS].to_numpy(dtype=np.float64),
                   static_ids=np.array(static, dtype=np.int64), raw_errors=raw.to_numpy(dtype=np.float64), n_runs=np.array(0))
        runs = [("percentile", 2.5, "per_camera", 10), ("percentile", 30.0, "per_camera", 10), ("percentile", 20.0, "overall", 10),
                ("percentile", 90.0, "per_camera", int(rng.integers(20, 60))), ("percentile", 100.0, "overall", 5),
                ("absolute", float(np.percentile(raw["euclidean_error"], 60)), "", 10), ("absolute", 0.05, "", int(rng.integers(15, 40))),
                ("absolute", float(np.percentile(raw["euclidean_error"][raw["cam_id"] == 0], 80)), "", int(0.5 * min((raw["cam_id"] == 0).sum(), (raw["cam_id"] == 1).sum()))),
                ("percentile", 40.0, "overall", int(0.45 * min((raw["cam_id"] == 0).sum(), (raw["cam_id"] == 1).sum())))]
        for n, (kind, value, scope, floor) in enumerate(runs):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                f = vol.filter_by_percentile_error(value, scope=scope, min_per_camera=floor) if kind == "percentile" else vol.filter_by_absolute_error(value, min_per_camera=floor)
            out[f"run{n}_args"] = np.array([0.0 if kind == "percentile" else 1.0, value, {"per_camera": 0.0, "overall": 1.0, "": -1.0}[scope], float(floor)])
            # how many cameras the safety floor has to top up in this run (plain numpy on the injected errors): with two or more, the reference under
            # pandas 2.3.3 (its lock file's version, and this container's) tops up only the FIRST as documented — its `keep_mask[camera_idx] = ...`
            # turns the boolean mask into an object column ("Setting an item of incompatible dtype" FutureWarning), `~keep_mask` is then true
            # everywhere, and every later camera gets the n-th smallest of ALL its errors as threshold instead of the n-th smallest of the dropped
            # ones: it ends below the floor it was asked for.  The consumer compares those runs for what they are.
            e, c = raw["euclidean_error"].to_numpy(), raw["cam_id"].to_numpy()
            if kind == "percentile" and scope == "per_camera":
                thr = {k: float(np.percentile(e[c == k], 100 - value)) for k in (0, 1) if (c == k).any()}
            else:
                thr = {k: (float(np.percentile(e, 100 - value)) if kind == "percentile" else value) for k in (0, 1)}
            out[f"run{n}_floor_cameras"] = np.array(sum(1 for k in thr if ((e[c == k] <= thr[k]).sum() < min(floor, (c == k).sum()))))
            out[f"run{n}_image"] = f.image_points.df[IMG_COLS].to_numpy(dtype=np.float64)
            out[f"run{n}_world"] = f.world_points.df[WORLD_COLS].to_numpy(dtype=np.float64)
            out[f"run{n}_map"] = np.asarray(f.img_to_obj_map, dtype=np.int64)
        out["n_runs"] = np.array(len(runs))
        np.savez_compressed(OUT / f"filter_{case:02d}.npz", **out)
        print(f"filter {case}: {len(idf)} observations ({int(matched.sum())} matched), static {static}: kept "
              + ", ".join(str(len(out[f'run{n}_image'])) + '/' + str(len(out[f'run{n}_world'])) for n in range(len(runs))))


def report_cases():
    """The bookkeeping of ``CaptureVolume.reprojection_report`` (core/capture_volume.py:151-236) around the pixel errors: which observations count
    (matched AND seen by a posed, non-ignored camera), the per-camera / per-point / overall RMS, the unmatched counts per camera.  The errors
    themselves come from ``cv2.projectPoints`` in the reference; here the module-level ``reprojection_errors`` the property calls is replaced by a
    function that hands back pre-drawn random errors (one row per counted observation), so what is pinned is everything the reference does in
    pandas around them.  Camera arrays with an unposed and an ignored camera that both have observations."""
    import caliscope.core.capture_volume as ref_cv
    from caliscope.cameras.camera_array import CameraArray, CameraData
    from caliscope.core.constraints import ConstraintSet
    from caliscope.core.point_data import ImagePoints, WorldPoints

    K = np.array([[400.0, 0.0, 200.0], [0.0, 400.0, 200.0], [0.0, 0.0, 1.0]])
    for case in range(6):
        world, img, _, _, static = random_tables(400 + case)
        rng = np.random.default_rng(19000 + case)
        wdf = pd.DataFrame(world, columns=WORLD_COLS).astype({"sync_index": "int64", "object_id": "int64", "keypoint_id": "int64"})
        idf = pd.DataFrame(img, columns=IMG_COLS).astype({c: "int64" for c in IMG_COLS[:4]})
        other = rng.random(len(idf))  # a fifth of the rows each go to an unposed camera (5) and to an ignored one (9); camera 12 has no observations
        idf.loc[other < 0.2, "cam_id"] = 5
        idf.loc[other > 0.8, "cam_id"] = 9
        idf = idf.drop_duplicates(subset=IMG_COLS[:4]).reset_index(drop=True)

        def cam(c, posed=True, ignore=False):
            return CameraData(cam_id=c, size=(400, 400), matrix=K.copy(), distortions=np.zeros(5), ignore=ignore,
                              rotation=np.eye(3) if posed else None, translation=np.array([0.1 * c, 0.0, 0.0]) if posed else None)

        cams = CameraArray({0: cam(0), 1: cam(1), 5: cam(5, posed=False), 9: cam(9, ignore=True), 12: cam(12)})
        cs = ConstraintSet((), frozenset(static)) if static else None
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            vol = ref_cv.CaptureVolume(cams, ImagePoints(idf), WorldPoints(wdf), cs)
        drawn = {}

        def fake_errors(camera_array, camera_indices, image_coords, world_coords):
            e = rng.normal(0, 1.0, (len(camera_indices), 2)) * (1.0 + np.asarray(camera_indices)[:, None])
            drawn["errors"], drawn["camera_indices"] = e, np.asarray(camera_indices).copy()
            return e

        real, ref_cv.reprojection_errors = ref_cv.reprojection_errors, fake_errors
        try:
            rep = vol.reprojection_report
        finally:
            ref_cv.reprojection_errors = real
        out = dict(
            world=vol.world_points.df[WORLD_COLS].to_numpy(dtype=np.float64), image=vol.image_points.df[IMG_COLS].to_numpy(dtype=np.float64),
            static_ids=np.array(static, dtype=np.int64), errors=drawn["errors"], camera_indices=drawn["camera_indices"].astype(np.int64),
            overall_rmse=np.array(rep.overall_rmse), by_camera=np.array(sorted(rep.by_camera.items()), dtype=np.float64).reshape(-1, 2),
            by_point=np.array(sorted((o, k, v) for (o, k), v in rep.by_point.items()), dtype=np.float64).reshape(-1, 3),
            n_unmatched=np.array(rep.n_unmatched_observations), unmatched_rate=np.array(rep.unmatched_rate),
            unmatched_by_camera=np.array(sorted(rep.unmatched_by_camera.items()), dtype=np.int64).reshape(-1, 2),
            raw_errors=rep.raw_errors.to_numpy(dtype=np.float64), raw_columns=np.array(list(rep.raw_errors.columns)),
            counts=np.array([rep.n_observations_matched, rep.n_observations_total, rep.n_cameras, rep.n_points], dtype=np.int64))
        np.savez_compressed(OUT / f"report_{case:02d}.npz", **out)
        print(f"report {case}: {len(idf)} observations, {rep.n_observations_matched} counted, unmatched by camera {dict(sorted(rep.unmatched_by_camera.items()))}, "
              f"{len(rep.by_point)} points, cameras in by_camera {sorted(rep.by_camera)}")


def seam_cases():
    """THE SEAM: the reference's ``CaptureVolume.optimize()`` (core/capture_volume.py:322-444) run with ``least_squares`` replaced by a
    recorder that returns a scripted result.  Stored: everything the reference hands to scipy at :387-411 — ``x0``, the eight ``args``, ``bounds``
    and the keyword arguments — and everything it makes of the result — the new cameras, the new points, ``OptimizationStatus`` (reason strings,
    bound warnings), the ``CalibrationError`` of a strict call that did not converge.  Camera arrays with sparse ids, an unposed and an ignored
    camera that have observations, fisheye cameras, locked and free intrinsics; tables with static objects; distance and centroid constraints;
    every termination status.  (Rotation vectors through the scipy ``Rodrigues`` stub, as in the bundle_* family.)"""
    from scipy.spatial.transform import Rotation

    import caliscope.core.capture_volume as ref_cv
    from caliscope.cameras.camera_array import CameraArray, CameraData
    from caliscope.core.constraints import CentroidDistanceConstraint, ConstraintSet, DistanceConstraint
    from caliscope.core.point_data import ImagePoints, WorldPoints
    from caliscope.exceptions import CalibrationError

    for case in range(10):
        rng = np.random.default_rng(23000 + case)
        world, img, dist, cent, static = random_tables(500 + case)
        wdf = pd.DataFrame(world, columns=WORLD_COLS).astype({"sync_index": "int64", "object_id": "int64", "keypoint_id": "int64"})
        wdf = wdf.drop_duplicates(subset=WORLD_COLS[:3]).reset_index(drop=True)
        idf = pd.DataFrame(img, columns=IMG_COLS).astype({c: "int64" for c in IMG_COLS[:4]})
        ids = [0, 1, 4, 7, 11]  # 4 unposed, 7 ignored, 11 fisheye; observations spread over all of them
        idf["cam_id"] = rng.choice(ids, size=len(idf), p=[0.3, 0.3, 0.1, 0.1, 0.2])
        idf = idf.drop_duplicates(subset=IMG_COLS[:4]).reset_index(drop=True)
        desc = []
        for c in ids:
            f = float(rng.uniform(300, 900))
            desc.append(dict(cam_id=c, size=(int(rng.integers(320, 1920)), int(rng.integers(240, 1080))), fisheye=c == 11, ignore=c == 7,
                             K=[[f, 0.0, float(rng.uniform(100, 600))], [0.0, f * 1.01, float(rng.uniform(100, 400))], [0.0, 0.0, 1.0]],
                             dist=rng.normal(0, 0.05, 4 if c == 11 else 5).tolist(), rvec=None if c == 4 else rng.normal(0, 0.6, 3).tolist(),
                             t=None if c == 4 else rng.normal(0, 1.0, 3).tolist()))
        cams = CameraArray({d["cam_id"]: CameraData(cam_id=d["cam_id"], size=tuple(d["size"]), matrix=np.array(d["K"]), distortions=np.array(d["dist"]), fisheye=d["fisheye"],
                                                    ignore=d["ignore"], rotation=None if d["rvec"] is None else Rotation.from_rotvec(d["rvec"]).as_matrix(),
                                                    translation=None if d["t"] is None else np.array(d["t"])) for d in desc})
        cs = ConstraintSet(tuple(DistanceConstraint(*d) for d in dist), frozenset(static), centroid_distances=tuple(CentroidDistanceConstraint(*c) for c in cent)) if case % 4 != 3 else None
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            vol = ref_cv.CaptureVolume(cams, ImagePoints(idf), WorldPoints(wdf), cs)
        status = [2, 1, 3, 4, 0, 2, 0, 2, 3, 2][case]
        kw = dict(ftol=[1e-8, 1e-10, 1e-6][case % 3], max_nfev=[None, 50, 7][case % 3], verbose=[0, 0, 1][case % 3], strict=case != 4,
                  use_constraints=case % 5 != 2, pixel_sigma=[1.0, 0.5, 2.0][case % 3], refine_intrinsics=bool(case % 2),
                  loss=["linear", "huber", "soft_l1", "cauchy"][case % 4], f_scale=[1.0, 0.002, 0.01][case % 3])
        seen = {}

        def recorder(fun, x0, args=(), jac=None, **kwargs):
            seen.update(fun=getattr(fun, "__name__", str(fun)), jac=getattr(jac, "__name__", str(jac)), x0=np.array(x0, dtype=np.float64), args=args, kwargs=kwargs)
            x = np.array(x0, dtype=np.float64) + 0.01 * np.sin(np.arange(len(x0)) * 0.7)
            par = args[0]
            for i, b in enumerate(par.blocks):  # free intrinsics now and then next to a bound: the status must carry the warning
                if b.free_intrinsics and i % 2 == 0:
                    x[par.camera_param_offsets[i] + 6] = 0.5025
                    x[par.camera_param_offsets[i] + 8] = 1.995
            seen["x"] = x
            return types.SimpleNamespace(x=x, status=status, nfev=int(rng.integers(3, 40)), cost=float(rng.uniform(0.1, 5.0)), optimality=1e-9, success=status > 0)

        real, ref_cv.least_squares = ref_cv.least_squares, recorder
        error = ""
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                out = vol.optimize(**kw)
        except CalibrationError as exc:
            out, error = None, str(exc)
        finally:
            ref_cv.least_squares = real
        par, cam_idx, uv, obj_idx, ga, gb, cd, cw = seen["args"]
        lb, ub = seen["kwargs"]["bounds"]
        fix = dict(
            world=vol.world_points.df[WORLD_COLS].to_numpy(dtype=np.float64), image=vol.image_points.df[IMG_COLS].to_numpy(dtype=np.float64),
            static_ids=np.array(static, dtype=np.int64), has_constraints=np.array(cs is not None),
            distances=np.array(dist, dtype=np.float64).reshape(-1, 6), centroids=np.array(cent, dtype=np.float64).reshape(-1, 4),
            cam_ids=np.array(ids), sizes=np.array([d["size"] for d in desc]), K=np.array([d["K"] for d in desc]),
            dist=np.array([d["dist"] + [np.nan] * (5 - len(d["dist"])) for d in desc]), fisheye=np.array([d["fisheye"] for d in desc]), ignore=np.array([d["ignore"] for d in desc]),
            rvec=np.array([d["rvec"] if d["rvec"] is not None else [np.nan] * 3 for d in desc]), t=np.array([d["t"] if d["t"] is not None else [np.nan] * 3 for d in desc]),
            call=np.array([kw["ftol"], -1 if kw["max_nfev"] is None else kw["max_nfev"], kw["verbose"], kw["strict"], kw["use_constraints"], kw["pixel_sigma"], kw["refine_intrinsics"],
                           kw["f_scale"]], dtype=np.float64), loss=np.array(kw["loss"]),
            fun=np.array(seen["fun"]), jac=np.array(seen["jac"]), x0=seen["x0"], camera_indices=np.asarray(cam_idx, dtype=np.int64), image_coords=np.asarray(uv, dtype=np.float64),
            obj_indices=np.asarray(obj_idx, dtype=np.int64), has_rows=np.array(ga is not None),
            groups_a=np.zeros((0, 4), np.int64) if ga is None else np.asarray(ga, dtype=np.int64), groups_b=np.zeros((0, 4), np.int64) if gb is None else np.asarray(gb, dtype=np.int64),
            row_distance=np.zeros(0) if cd is None else np.asarray(cd, dtype=np.float64), row_weight=np.zeros(0) if cw is None else np.asarray(cw, dtype=np.float64),
            lb=np.asarray(lb), ub=np.asarray(ub), offsets=np.array(par.camera_param_offsets), n_camera_params=np.array(par.n_camera_params),
            kwargs_keys=np.array(sorted(k for k in seen["kwargs"] if k != "bounds")),
            kwargs_values=np.array([str(seen["kwargs"][k]) for k in sorted(seen["kwargs"]) if k != "bounds"]),
            x_result=seen["x"], result_status=np.array(status), error=np.array(error), returned=np.array(out is not None))
        if out is not None:
            st = out.optimization_status
            fix.update(
                out_R=np.array([out.camera_array.cameras[c].rotation if out.camera_array.cameras[c].rotation is not None else np.full((3, 3), np.nan) for c in ids]),
                out_t=np.array([np.ravel(out.camera_array.cameras[c].translation) if out.camera_array.cameras[c].translation is not None else [np.nan] * 3 for c in ids]),
                out_K=np.array([out.camera_array.cameras[c].matrix for c in ids]),
                out_dist=np.array([list(np.ravel(out.camera_array.cameras[c].distortions)) + [np.nan] * (5 - np.size(out.camera_array.cameras[c].distortions)) for c in ids]),
                out_world=out.world_points.df[WORLD_COLS].to_numpy(dtype=np.float64), out_map=np.asarray(out.img_to_obj_map, dtype=np.int64),
                status_fields=np.array([st.converged, st.iterations, st.final_cost], dtype=np.float64), status_reason=np.array(st.termination_reason),
                status_warnings=np.array([[w.cam_id, {"f": 0, "k1": 1, "k2": 2}[w.parameter], {"lower": 0, "upper": 1}[w.bound], w.value] for w in st.bound_warnings],
                                         dtype=np.float64).reshape(-1, 4),
                source_untouched=np.array(all(np.array_equal(vol.camera_array.cameras[d["cam_id"]].matrix, np.array(d["K"])) for d in desc)))
        np.savez_compressed(OUT / f"seam_{case:02d}.npz", **fix)
        print(f"seam {case}: {len(cam_idx)} of {len(idf)} observations handed over, {par.n_camera_params} camera parameters (refine {kw['refine_intrinsics']}), "
              f"{0 if ga is None else len(ga)} constraint rows, loss {kw['loss']}, status {status} -> "
              + (f"error: {error.splitlines()[0]}" if out is None else f"{out.optimization_status.termination_reason}, {len(out.optimization_status.bound_warnings)} bound warnings"))


def constraint_row_cases():
    """The constraint rows of the reference's OWN ``joint_residuals`` / ``joint_jacobian`` (core/reprojection.py:112-117, :207-226).  Those two functions
    reach OpenCV only inside their per-camera loops, which skip a camera without observations — so called with EMPTY observation arrays they return
    exactly the constraint rows, computed by the reference's numpy code: corner endpoints (a row repeated four times), centroid endpoints (four
    distinct rows), endpoints that share rows, coincident endpoints (the zero subgradient), random weights.  Consumer: the oracle
    (oracle/residuals.py), which the device rows are compared with."""
    from caliscope.cameras.camera_array import CameraArray, CameraData
    from caliscope.core.bundle_parameterization import BundleParameterization
    from caliscope.core.reprojection import joint_jacobian, joint_residuals

    K = np.array([[400.0, 0.0, 200.0], [0.0, 400.0, 200.0], [0.0, 0.0, 1.0]])
    for case in range(6):
        rng = np.random.default_rng(29000 + case)
        n_cams, refine = int(rng.integers(2, 5)), bool(case % 2)
        cams = CameraArray({c: CameraData(cam_id=c, size=(400, 400), matrix=K.copy(), distortions=np.zeros(5), rotation=np.eye(3), translation=np.zeros(3)) for c in range(n_cams)})
        n_points = int(rng.integers(12, 40))
        par = BundleParameterization.from_camera_array(cams, n_points, refine_intrinsics=refine)
        x = np.concatenate([rng.normal(0, 0.3, par.n_camera_params), rng.normal(0, 1.0, 3 * n_points)])
        m = int(rng.integers(6, 20))
        ga, gb = np.empty((m, 4), dtype=np.int32), np.empty((m, 4), dtype=np.int32)
        for i in range(m):
            for g in (ga, gb):
                g[i] = rng.integers(0, n_points) if rng.random() < 0.6 else rng.choice(n_points, size=4, replace=False)
        ga[0], gb[0] = 3, 3                      # coincident endpoints: the norm is not differentiable there
        gb[1] = ga[1]                            # the same group on both sides
        if m > 3:
            gb[2, :2] = ga[2, :2]                # groups that share rows
        dist, wgt = rng.uniform(0.05, 2.0, m), rng.uniform(10.0, 2000.0, m)
        none_i, none_uv = np.zeros(0, dtype=np.int16), np.zeros((0, 2))
        r = joint_residuals(x, par, none_i, none_uv, np.zeros(0, dtype=np.int32), ga, gb, dist, wgt)
        J = joint_jacobian(x, par, none_i, none_uv, np.zeros(0, dtype=np.int32), ga, gb, dist, wgt)
        np.savez_compressed(OUT / f"conrows_{case:02d}.npz", n_cams=np.array(n_cams), refine=np.array(refine), n_points=np.array(n_points), x=x, groups_a=ga, groups_b=gb,
                            distances=dist, weights=wgt, residuals=np.asarray(r), jacobian=np.asarray(J.todense()))
        print(f"conrows {case}: {m} rows over {n_points} points, {par.n_camera_params} camera parameters: |r| {np.linalg.norm(r):.3f}, {J.nnz} non-zeros in {J.shape}")


def driver_cases():
    """``calibrate_extrinsics`` (core/calibrate_extrinsics.py:44-261) with its three heavy calls scripted (tests/driver_script.py: bootstrap hands back
    a given "triangulation", optimize and the filter record their arguments) and everything else real: blind intrinsics, anchors, the static-marker
    guard, the depth-ratio gate, the stage sequence, the progress marks, the ``CalibrationRun``."""
    sys.path.insert(0, str(HERE.parent.parent))
    import caliscope.core.calibrate_extrinsics as ref_drv
    from caliscope.cameras.camera_array import CameraArray, CameraData
    from caliscope.core.capture_volume import CaptureVolume
    from caliscope.core.constraints import ConstraintSet, DistanceConstraint, PointRemap
    from caliscope.core.point_data import ImagePoints, WorldPoints
    from caliscope.exceptions import CalibrationError
    from tests.driver_script import scripted

    corners = np.array([[-0.5, 0.5, 0.0], [0.5, 0.5, 0.0], [0.5, -0.5, 0.0], [-0.5, -0.5, 0.0]])
    for case in range(8, 11):
        # two-sided charuco board with a substrate (object 0 = front face, 1 = back face): the extraction check (:328-370), the count of firing
        # cross-face rows (:373-391) and the error when the two faces are never triangulated at the same instant.  8: all well; 9: the faces are never
        # seen together; 10: the extraction carries another thickness than the configuration.
        rng = np.random.default_rng(31000 + case)
        sq, thick = 0.04, 0.006
        grid = np.array([[(i + 1) * sq, (j + 1) * sq, 0.0] for j in range(3) for i in range(4)], dtype=np.float32)
        stand_in = types.SimpleNamespace(board=types.SimpleNamespace(getChessboardCorners=lambda g=grid: g, getSquareLength=lambda q=sq: q), thickness_m=thick)
        cs = ConstraintSet.from_charuco(stand_in)
        frames = list(range(8))
        world, img = [], []
        for si in frames:
            centre = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), rng.uniform(1.0, 6.0)])
            for o in (0, 1):
                if case == 9 and (si % 2) != o:
                    continue  # front face on even frames only, back face on odd ones: no common instant
                for k in range(len(grid)):
                    p = centre + grid[k].astype(np.float64) + [0.0, 0.0, thick * o]
                    world.append((si, o, k, *p.tolist(), si / 30.0))
                    for cam in (0, 1, 3):
                        if rng.random() < 0.9:
                            img.append((si, cam, o, k, float(200 + rng.normal(0, 40)), float(200 + rng.normal(0, 40)), float(grid[k][0]), float(grid[k][1]),
                                        (0.004 if case == 10 else thick) * o))
        wdf = pd.DataFrame(world, columns=WORLD_COLS).astype({"sync_index": "int64", "object_id": "int64", "keypoint_id": "int64"})
        idf = pd.DataFrame(img, columns=IMG_COLS + ["obj_loc_x", "obj_loc_y", "obj_loc_z"]).astype({c: "int64" for c in IMG_COLS[:4]})
        desc = []
        for c in (0, 1, 3):
            f = float(rng.uniform(300, 900))
            desc.append(dict(cam_id=c, size=(1280, 720), ignore=False, has_intrinsics=True, K=[[f, 0.0, 320.0], [0.0, f, 240.0], [0.0, 0.0, 1.0]], dist=rng.normal(0, 0.05, 5).tolist(),
                             t=[0.2 * c, 0.0, 0.0]))
        cams = CameraArray({d["cam_id"]: CameraData(cam_id=d["cam_id"], size=tuple(d["size"]), matrix=np.array(d["K"]), distortions=np.array(d["dist"]), rotation=np.eye(3),
                                                    translation=np.array(d["t"])) for d in desc})
        trace, error, run = [], "", None
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with scripted(CaptureVolume, WorldPoints, wdf, trace):
                try:
                    run = ref_drv.calibrate_extrinsics(ImagePoints(idf), cams, cs, refine_intrinsics=True, filter_percentile=2.5,
                                                       progress=lambda pct, msg: trace.append(("progress", int(pct), str(msg))))
                except CalibrationError as exc:
                    error = str(exc)
        dist = [(d.object_id_a, d.keypoint_id_a, d.object_id_b, d.keypoint_id_b, d.distance, d.sigma) for d in cs.distances]
        out = dict(world=wdf.to_numpy(dtype=np.float64), image=idf.to_numpy(dtype=np.float64), static_ids=np.zeros(0, dtype=np.int64), has_constraints=np.array(True),
                   distances=np.array(dist, dtype=np.float64).reshape(-1, 6), remaps=np.zeros((0, 7)), thickness=np.array(thick),
                   cam_ids=np.array([d["cam_id"] for d in desc]), sizes=np.array([d["size"] for d in desc]), K=np.array([d["K"] for d in desc]), dist=np.array([d["dist"] for d in desc]),
                   ignore=np.array([d["ignore"] for d in desc]), has_intrinsics=np.array([d["has_intrinsics"] for d in desc]), t=np.array([d["t"] for d in desc]),
                   refine=np.array(True), filter_percentile=np.array(2.5), trace=np.array(repr(trace)), error_type=np.array("CalibrationError" if error else ""),
                   error_mentions=np.zeros(0, dtype=np.int64), error_words=np.array([w for w in ("cross-face", "thickness") if w in error.lower()]), returned=np.array(run is not None),
                   firing=np.array(ref_drv._count_firing_cross_face_rows(wdf, cs.distances)))
        if run is not None:
            out.update(synthesized=np.array(sorted(run.synthesized_cam_ids), dtype=np.int64), dropped=np.array(run.dropped_static_markers, dtype=np.int64),
                       gated=np.array(run.intrinsic_refinement_gated),
                       estimates=np.array([[e.cam_id, e.f_recovered, e.k1_recovered, e.k2_recovered, e.f_initial, e.k1_initial, e.k2_initial] for e in run.intrinsic_estimates],
                                          dtype=np.float64).reshape(-1, 7),
                       final_counts=np.array([len(run.capture_volume.image_points.df), len(run.capture_volume.world_points.df)], dtype=np.int64),
                       final_image_keys=run.capture_volume.image_points.df[IMG_COLS[:4]].to_numpy(dtype=np.int64))
        np.savez_compressed(OUT / f"driver_{case:02d}.npz", **out)
        print(f"driver {case} (two-sided board, {int(out['firing'])} cross-face rows firing): " + (f"error ({error.splitlines()[0][:80]}...)" if run is None else "ran")
              + f"; {len(trace)} trace entries")
    for case in range(8):
        rng = np.random.default_rng(31000 + case)
        shallow, deformed, blind, no_obj_loc, refine, remap = case == 1, case == 3, case in (4, 5), case == 5, case != 6, case == 7
        n_obj = 3 if not remap else 4
        static = [2] if case in (2, 3) else []
        size = {o: float(rng.uniform(0.1, 0.3)) for o in range(n_obj)}
        frames = list(range(int(rng.integers(6, 12))))
        world, img = [], []
        for o in range(n_obj):
            for si in ([-1] if o in static else frames):
                centre = np.array([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(4.0, 5.0) if shallow else rng.uniform(1.0, 6.0)])
                for k in range(4):
                    p = centre + size[o] * corners[k] + (rng.normal(0, 0.6 * size[o], 3) if (deformed and o in static) else 0.0)
                    world.append((si, o, k, *p.tolist(), float("nan") if o in static else si / 30.0))
            for si in frames:
                for k in range(4):
                    for cam in (0, 1, 3):
                        if rng.random() < 0.9:
                            loc = [float("nan")] * 3 if no_obj_loc else (size[o] * corners[k]).tolist()
                            img.append((si, cam, o, k, float(200 + rng.normal(0, 40)), float(200 + rng.normal(0, 40)), *loc))
        wdf = pd.DataFrame(world, columns=WORLD_COLS).astype({"sync_index": "int64", "object_id": "int64", "keypoint_id": "int64"})
        idf = pd.DataFrame(img, columns=IMG_COLS + ["obj_loc_x", "obj_loc_y", "obj_loc_z"]).astype({c: "int64" for c in IMG_COLS[:4]})
        desc = []
        for c in (0, 1, 3, 6):
            f = float(rng.uniform(300, 900))
            desc.append(dict(cam_id=c, size=(int(rng.integers(320, 1920)), int(rng.integers(240, 1080))), ignore=c == 6, has_intrinsics=not (blind and c == 3),
                             K=[[f, 0.0, 320.0], [0.0, f, 240.0], [0.0, 0.0, 1.0]], dist=rng.normal(0, 0.05, 5).tolist(), t=[0.2 * c, 0.0, 0.0]))
        cams = CameraArray({d["cam_id"]: CameraData(cam_id=d["cam_id"], size=tuple(d["size"]), matrix=np.array(d["K"]) if d["has_intrinsics"] else None,
                                                    distortions=np.array(d["dist"]) if d["has_intrinsics"] else None, ignore=d["ignore"], rotation=np.eye(3),
                                                    translation=np.array(d["t"])) for d in desc})
        dist = [(o, i, o, j, float(size[o] * np.linalg.norm(corners[i] - corners[j])), 0.002) for o in range(n_obj if not remap else 3) for i in range(4) for j in range(i + 1, 4)]
        remaps = [(3, k, 0, (k + 1) % 4, *(size[0] * corners[(k + 1) % 4]).tolist()) for k in range(4)] if remap else []
        cs = None if case == 0 else ConstraintSet(tuple(DistanceConstraint(*d) for d in dist), frozenset(static), point_remaps=tuple(PointRemap(*r) for r in remaps))
        trace, error, run = [], "", None
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with scripted(CaptureVolume, WorldPoints, wdf, trace):
                try:
                    run = ref_drv.calibrate_extrinsics(ImagePoints(idf), cams, cs, refine_intrinsics=refine, filter_percentile=[2.5, 5.0][case % 2],
                                                       progress=lambda pct, msg: trace.append(("progress", int(pct), str(msg))))
                except CalibrationError as exc:
                    error = str(exc)
        out = dict(world=wdf.to_numpy(dtype=np.float64), image=idf.to_numpy(dtype=np.float64), static_ids=np.array(static, dtype=np.int64), has_constraints=np.array(cs is not None),
                   distances=np.array(dist, dtype=np.float64).reshape(-1, 6), remaps=np.array(remaps, dtype=np.float64).reshape(-1, 7),
                   cam_ids=np.array([d["cam_id"] for d in desc]), sizes=np.array([d["size"] for d in desc]), K=np.array([d["K"] for d in desc]), dist=np.array([d["dist"] for d in desc]),
                   ignore=np.array([d["ignore"] for d in desc]), has_intrinsics=np.array([d["has_intrinsics"] for d in desc]), t=np.array([d["t"] for d in desc]),
                   refine=np.array(refine), filter_percentile=np.array([2.5, 5.0][case % 2]), trace=np.array(repr(trace)), error_type=np.array("CalibrationError" if error else ""),
                   error_mentions=np.array(sorted(c for c in (0, 1, 3, 6) if error and str(c) in error.split("cameras")[1].split("have")[0]) if error else [], dtype=np.int64),
                   returned=np.array(run is not None))
        if run is not None:
            out.update(synthesized=np.array(sorted(run.synthesized_cam_ids), dtype=np.int64), dropped=np.array(run.dropped_static_markers, dtype=np.int64),
                       gated=np.array(run.intrinsic_refinement_gated),
                       estimates=np.array([[e.cam_id, e.f_recovered, e.k1_recovered, e.k2_recovered, e.f_initial, e.k1_initial, e.k2_initial] for e in run.intrinsic_estimates],
                                          dtype=np.float64).reshape(-1, 7),
                       final_counts=np.array([len(run.capture_volume.image_points.df), len(run.capture_volume.world_points.df)], dtype=np.int64),
                       final_image_keys=run.capture_volume.image_points.df[IMG_COLS[:4]].to_numpy(dtype=np.int64))
        np.savez_compressed(OUT / f"driver_{case:02d}.npz", **out)
        print(f"driver {case}: " + (f"error ({error.splitlines()[0][:70]}...)" if run is None else
                                   f"synthesized {sorted(run.synthesized_cam_ids)}, dropped {list(run.dropped_static_markers)}, gated {run.intrinsic_refinement_gated}, "
                                   f"{len(run.intrinsic_estimates)} estimates") + f"; {len(trace)} trace entries: " + " ".join(t[0][0] + (str(t[1]) if t[0] == "progress" else "") for t in trace))


def dlt_cases():
    """``triangulate_image_points`` (core/point_data.py:121-232): the reference's batched SVD triangulation is a module-level numpy function — no OpenCV,
    no stub.  Random rigs, points seen by two to all cameras (and by one: not triangulated), noisy normalised coordinates, several objects and
    frames.  Consumer: the oracle's restatement (oracle/triangulation.py), which the device triangulation is compared with on the GPU."""
    from scipy.spatial.transform import Rotation

    from caliscope.core.point_data import triangulate_image_points

    for case in range(6):
        rng = np.random.default_rng(37000 + case)
        n_cams = int(rng.integers(2, 8))
        cam_ids = sorted(rng.choice(30, size=n_cams, replace=False).tolist())
        P = {}
        for c in cam_ids:
            R = Rotation.from_rotvec(rng.normal(0, 0.4, 3)).as_matrix()
            P[int(c)] = np.hstack([R, (rng.normal(0, 0.5, 3) + [0, 0, 4.0]).reshape(3, 1)])
        sync, cam, obj, kp, xy = [], [], [], [], []
        for f in range(int(rng.integers(2, 6))):
            for o in range(int(rng.integers(1, 4))):
                for k in range(int(rng.integers(1, 6))):
                    X = np.append(rng.normal(0, 0.6, 3), 1.0)
                    for c in rng.choice(cam_ids, size=int(rng.integers(1, n_cams + 1)), replace=False):
                        h = P[int(c)] @ X
                        sync.append(f); cam.append(int(c)); obj.append(o); kp.append(k)
                        xy.append(h[:2] / h[2] + rng.normal(0, 1e-3, 2))
        order = rng.permutation(len(sync))
        sync, cam, obj, kp, xy = (np.asarray(a)[order] for a in (sync, cam, obj, kp, xy))
        o_sync, o_obj, o_kp, o_xyz = triangulate_image_points(P, sync, cam, obj, kp, xy)
        np.savez_compressed(OUT / f"dlt_{case:02d}.npz", cam_ids=np.array(cam_ids), P=np.array([P[int(c)] for c in cam_ids]), sync=sync, cam=cam, obj=obj, kp=kp, xy=xy,
                            out_sync=o_sync, out_obj=o_obj, out_kp=o_kp, out_xyz=o_xyz)
        print(f"dlt {case}: {n_cams} cameras, {len(sync)} observations -> {len(o_xyz)} points")


def triangulate_cases():
    """``ImagePoints.triangulate(camera_array, static_object_ids)`` (core/point_data.py:416-560) for cameras WITHOUT lens distortion: which cameras take
    part (posed, not ignored), static objects pooled over all frames into one point at STATIC_SYNC_INDEX, points seen once left out, the mean frame
    time per sync index, the table that comes back.  The reference undistorts through ``cv2.undistortPoints`` on float32 copies of the pixel
    coordinates; for zero distortion coefficients that function is the linear map (u - cx) / fx, (v - cy) / fy evaluated in double and returned as
    float32 — which is what the stub does here (it refuses any non-zero coefficient).  The triangulation itself is the reference's numpy code."""
    import cv2
    from scipy.spatial.transform import Rotation

    from caliscope.cameras.camera_array import CameraArray, CameraData
    from caliscope.core.point_data import ImagePoints

    def undistort_zero(points, K, dist, P=None):
        assert not np.any(np.asarray(dist)), "the stub stands for OpenCV only where OpenCV is the identity"
        assert P is not None and np.array_equal(np.asarray(P), np.identity(3))
        pts = np.asarray(points)
        assert pts.dtype == np.float32
        out = np.empty_like(pts)
        out[..., 0] = ((pts[..., 0].astype(np.float64) - K[0, 2]) / K[0, 0]).astype(np.float32)
        out[..., 1] = ((pts[..., 1].astype(np.float64) - K[1, 2]) / K[1, 1]).astype(np.float32)
        return out

    cv2.undistortPoints = undistort_zero
    for case in range(6):
        rng = np.random.default_rng(41000 + case)
        ids = sorted(rng.choice(20, size=int(rng.integers(3, 7)), replace=False).tolist())
        desc = []
        for n, c in enumerate(ids):
            f = float(rng.uniform(500, 1200))
            desc.append(dict(cam_id=int(c), K=[[f, 0.0, float(rng.uniform(300, 700))], [0.0, f * 1.01, float(rng.uniform(200, 500))], [0.0, 0.0, 1.0]],
                             rvec=rng.normal(0, 0.3, 3).tolist(), t=(rng.normal(0, 0.4, 3) + [0, 0, 4.0]).tolist(),
                             posed=not (n == len(ids) - 1 and case % 2 == 0), ignore=(n == 0 and case % 3 == 0)))
        cams = CameraArray({d["cam_id"]: CameraData(cam_id=d["cam_id"], size=(1280, 720), matrix=np.array(d["K"]), distortions=np.zeros(5), ignore=d["ignore"],
                                                    rotation=Rotation.from_rotvec(d["rvec"]).as_matrix() if d["posed"] else None,
                                                    translation=np.array(d["t"]) if d["posed"] else None) for d in desc})
        static = [5] if case % 2 else []
        rows = []
        frames = list(range(int(rng.integers(2, 6))))
        for o in (0, 3, 5):
            fixed = {k: rng.normal(0, 0.5, 3) for k in range(4)}
            for f in frames:
                for k in range(4):
                    X = fixed[k] if o in static else rng.normal(0, 0.5, 3)
                    for d in desc:
                        if rng.random() < 0.7:
                            R = Rotation.from_rotvec(d["rvec"]).as_matrix()
                            h = R @ X + np.array(d["t"])
                            K = np.array(d["K"])
                            rows.append((f, d["cam_id"], o, k, float(K[0, 0] * h[0] / h[2] + K[0, 2] + rng.normal(0, 0.3)), float(K[1, 1] * h[1] / h[2] + K[1, 2] + rng.normal(0, 0.3)),
                                         f / 30.0 + 0.001 * d["cam_id"]))
        rows = [rows[i] for i in rng.permutation(len(rows))]
        idf = pd.DataFrame(rows, columns=IMG_COLS + ["frame_time"]).astype({c: "int64" for c in IMG_COLS[:4]})
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            wp = ImagePoints(idf).triangulate(cams, static_object_ids=frozenset(static))
        out = wp.df
        np.savez_compressed(OUT / f"triangulate_{case:02d}.npz", image=idf.to_numpy(dtype=np.float64), cam_ids=np.array(ids), K=np.array([d["K"] for d in desc]),
                            rvec=np.array([d["rvec"] for d in desc]), t=np.array([d["t"] for d in desc]), posed=np.array([d["posed"] for d in desc]),
                            ignore=np.array([d["ignore"] for d in desc]), static_ids=np.array(static, dtype=np.int64),
                            world=out[WORLD_COLS].to_numpy(dtype=np.float64), world_columns=np.array(list(out.columns)))
        print(f"triangulate {case}: cameras {ids} (unposed {[d['cam_id'] for d in desc if not d['posed']]}, ignored {[d['cam_id'] for d in desc if d['ignore']]}), "
              f"{len(idf)} observations, static {static} -> {len(out)} world points ({int((out['sync_index'] == -1).sum())} static)")


def remap_cases():
    """``ConstraintSet.remap_image_points`` (core/constraints.py:192-214) with arbitrary ``PointRemap`` tuples — also ones whose target is the source of
    a LATER remap (the reference applies them one after the other on the same frame, so such observations move twice) and ones nothing matches."""
    from caliscope.core.constraints import ConstraintSet, PointRemap
    from caliscope.core.point_data import ImagePoints

    for case in range(6):
        rng = np.random.default_rng(43000 + case)
        rows = []
        for f in range(int(rng.integers(3, 8))):
            for o in range(5):
                for k in range(4):
                    for cam in range(3):
                        if rng.random() < 0.6:
                            rows.append((f, cam, o, k, float(rng.normal(300, 50)), float(rng.normal(300, 50)), *(rng.normal(0, 0.1, 3).tolist() if rng.random() < 0.8 else [float("nan")] * 3)))
        rows = [rows[i] for i in rng.permutation(len(rows))]
        idf = pd.DataFrame(rows, columns=IMG_COLS + ["obj_loc_x", "obj_loc_y", "obj_loc_z"]).astype({c: "int64" for c in IMG_COLS[:4]})
        remaps = [(int(rng.integers(0, 6)), int(rng.integers(0, 4)), int(rng.integers(0, 5)), int(rng.integers(0, 4)), *rng.normal(0, 0.1, 3).tolist()) for _ in range(int(rng.integers(1, 9)))]
        if case % 2 == 0 and len(remaps) >= 2:  # a chain: the second remap picks up what the first produced
            a = remaps[0]
            remaps[1] = (a[2], a[3], int(rng.integers(0, 5)), int(rng.integers(0, 4)), *rng.normal(0, 0.1, 3).tolist())
        cs = ConstraintSet((), frozenset(), point_remaps=tuple(PointRemap(*r) for r in remaps))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = cs.remap_image_points(ImagePoints(idf.copy())).df
            same_object = ConstraintSet((), frozenset()).remap_image_points(ip := ImagePoints(idf.copy())) is ip
        np.savez_compressed(OUT / f"remap_{case:02d}.npz", image=idf.to_numpy(dtype=np.float64), remaps=np.array(remaps, dtype=np.float64).reshape(-1, 7),
                            out=out.to_numpy(dtype=np.float64), out_columns=np.array(list(out.columns)), no_remaps_returns_the_input=np.array(same_object))
        moved = int(((out[["object_id", "keypoint_id"]].to_numpy() != idf[["object_id", "keypoint_id"]].to_numpy()).any(axis=1)).sum()) if len(out) == len(idf) else -1
        print(f"remap {case}: {len(idf)} observations, {len(remaps)} remaps -> {moved} observations renamed")


def camera_toml_cases():
    """``CameraArray.from_toml`` (cameras/camera_array.py:377-441) on files the reference's side wrote: the real session's ``camera_array.toml`` (3 x 3
    rotation matrices, an extra key) and hand-made variants of the shapes its loader accepts — rotation as a 3-vector (through the scipy
    ``Rodrigues`` stub), the string "null" for a missing value, absent optional keys, an unposed camera, a fisheye camera, an empty file."""
    from caliscope.cameras.camera_array import CameraArray

    texts = {
        "session": (HERE / "post_optimization" / "camera_array.toml").read_text(),
        "variants": """
[cameras.3]
cam_id = 3
size = [640, 480]
matrix = [[500.0, 0.0, 320.0], [0.0, 501.0, 240.0], [0.0, 0.0, 1.0]]
distortions = [0.01, -0.02, 0.0, 0.0, 0.003]
rotation = [0.1, -0.2, 0.3]
translation = [0.5, 0.25, 2.0]
error = "null"
exposure = "null"

[cameras.11]
cam_id = 11
size = [1920, 1080]
rotation_count = 2
fisheye = true
ignore = true
grid_count = 17
matrix = [[700.0, 0.0, 960.0], [0.0, 700.0, 540.0], [0.0, 0.0, 1.0]]
distortions = [0.05, -0.01, 0.003, -0.001]
rotation = [[0.0], [0.0], [0.5]]
translation = [0.0, 0.0, 1.0]

[cameras.7]
cam_id = 7
size = [800, 600]
""",
        "empty": "",
    }
    out = {}
    for name, text in texts.items():
        with tempfile.TemporaryDirectory() as tmp:
            path = Path(tmp) / "camera_array.toml"
            path.write_text(text)
            arr = CameraArray.from_toml(path)
        ids = sorted(arr.cameras)

        def opt(v):
            return np.nan if v is None else float(v)

        out[f"{name}_text"] = np.array(text)
        out[f"{name}_ids"] = np.array(ids, dtype=np.int64)
        out[f"{name}_scalars"] = np.array([[arr.cameras[c].size[0], arr.cameras[c].size[1], arr.cameras[c].rotation_count, opt(arr.cameras[c].error), opt(arr.cameras[c].exposure),
                                             opt(arr.cameras[c].grid_count), float(bool(arr.cameras[c].ignore)), float(bool(arr.cameras[c].fisheye))] for c in ids], dtype=np.float64).reshape(-1, 8)
        out[f"{name}_K"] = np.array([arr.cameras[c].matrix if arr.cameras[c].matrix is not None else np.full((3, 3), np.nan) for c in ids]).reshape(-1, 3, 3)
        out[f"{name}_dist"] = np.array([list(np.ravel(arr.cameras[c].distortions)) + [np.nan] * (5 - np.size(arr.cameras[c].distortions)) if arr.cameras[c].distortions is not None
                                        else [np.nan] * 5 for c in ids]).reshape(-1, 5)
        out[f"{name}_n_dist"] = np.array([-1 if arr.cameras[c].distortions is None else np.size(arr.cameras[c].distortions) for c in ids], dtype=np.int64)
        out[f"{name}_R"] = np.array([arr.cameras[c].rotation if arr.cameras[c].rotation is not None else np.full((3, 3), np.nan) for c in ids]).reshape(-1, 3, 3)
        out[f"{name}_t"] = np.array([np.ravel(arr.cameras[c].translation) if arr.cameras[c].translation is not None else [np.nan] * 3 for c in ids]).reshape(-1, 3)
        print(f"camera toml '{name}': cameras {ids}, posed {[c for c in ids if arr.cameras[c].rotation is not None]}")
    np.savez_compressed(OUT / "camtoml_00.npz", names=np.array(list(texts)), **out)


if __name__ == "__main__":
    main()
    camera_toml_cases()
    remap_cases()
    triangulate_cases()
    dlt_cases()
    driver_cases()
    constraint_row_cases()
    bundle_cases()
    table_cases()
    interop_cases()
    compiler_cases()
    filter_cases()
    report_cases()
    seam_cases()
