"""A small rigid-board scene with distance constraints (test data for the constraint rows)."""

import numpy as np

from caliscope_amd.bundle_parameterization import BundleParameterization
from caliscope_amd.cameras import matrix_to_rvec, rvec_to_matrix
from caliscope_amd.synthetic import project_pinhole_bc5, ring_camera_array
from caliscope_amd.cameras import CameraArray, CameraData


def board_scene(n_cams=5, n_frames=6, rows=3, cols=4, spacing=0.06, seed=3, noise_px=0.4, sigma_m=0.002, pixel_sigma=1.0,
                perturb=True):
    """Cameras on a ring, a rows x cols board moved rigidly through n_frames poses; every camera sees every corner.
    Constraints per frame: horizontal / vertical neighbours and both cell diagonals (corner endpoints = one point
    repeated four times, as the reference encodes them) plus one centroid-to-centroid row between the first and the
    last cell (four distinct points per endpoint)."""
    rng = np.random.default_rng(seed)
    cams = ring_camera_array(n_cams)
    grid = np.array([[c * spacing, r * spacing, 0.0] for r in range(rows) for c in range(cols)])
    grid -= grid.mean(axis=0)
    pts, frame_of = [], []
    for f in range(n_frames):
        R = rvec_to_matrix(rng.normal(0, 0.5, 3))
        t = np.array([rng.uniform(-0.25, 0.25), rng.uniform(-0.25, 0.25), rng.uniform(0.35, 0.85)])
        pts.append(grid @ R.T + t)
        frame_of += [f] * len(grid)
    pts = np.vstack(pts)
    n_per = rows * cols
    cam_idx, uv, obj = [], [], []
    for c, cam in sorted(cams.cameras.items()):
        K = cam.matrix
        p, z = project_pinhole_bc5(pts, cam.rotation, cam.translation, K[0, 0], K[1, 1], K[0, 2], K[1, 2], cam.distortions)
        ok = z > 0
        cam_idx.append(np.full(ok.sum(), c)); uv.append(p[ok] + rng.normal(0, noise_px, (ok.sum(), 2))); obj.append(np.flatnonzero(ok))
    cam_idx, uv, obj = np.concatenate(cam_idx).astype(np.int32), np.vstack(uv), np.concatenate(obj).astype(np.int32)
    ga, gb, dist = [], [], []
    idx = lambda r, c: r * cols + c
    for f in range(n_frames):
        base = f * n_per
        edges = []
        for r in range(rows):
            for c in range(cols):
                if c + 1 < cols: edges.append((idx(r, c), idx(r, c + 1)))
                if r + 1 < rows: edges.append((idx(r, c), idx(r + 1, c)))
                if c + 1 < cols and r + 1 < rows:
                    edges.append((idx(r, c), idx(r + 1, c + 1))); edges.append((idx(r, c + 1), idx(r + 1, c)))
        for a, b in edges:
            ga.append([base + a] * 4); gb.append([base + b] * 4); dist.append(float(np.linalg.norm(grid[a] - grid[b])))
        cell_a = [idx(0, 0), idx(0, 1), idx(1, 0), idx(1, 1)]
        cell_b = [idx(rows - 2, cols - 2), idx(rows - 2, cols - 1), idx(rows - 1, cols - 2), idx(rows - 1, cols - 1)]
        ga.append([base + i for i in cell_a]); gb.append([base + i for i in cell_b])
        dist.append(float(np.linalg.norm(grid[cell_a].mean(axis=0) - grid[cell_b].mean(axis=0))))
    ga, gb, dist = np.array(ga, dtype=np.int32), np.array(gb, dtype=np.int32), np.array(dist)
    f_median = float(np.median([cam.matrix[0, 0] for cam in cams.cameras.values()]))
    weights = np.full(len(dist), (pixel_sigma / f_median) / sigma_m)  # reference capture_volume.py:381
    init = CameraArray({c: CameraData(cam_id=c, size=cam.size, matrix=cam.matrix.copy(), distortions=cam.distortions.copy(),
                                      rotation=rvec_to_matrix(matrix_to_rvec(cam.rotation) + (rng.normal(0, 0.01, 3) if perturb else 0)),
                                      translation=cam.translation + (rng.normal(0, 0.02, 3) if perturb else 0))
                        for c, cam in cams.cameras.items()})
    pts0 = pts + (rng.normal(0, 0.01, pts.shape) if perturb else 0)
    par = BundleParameterization.from_camera_array(init, n_points=len(pts), refine_intrinsics=False)
    x0 = par.pack(init, pts0)
    return dict(par=par, x0=x0, cam=cam_idx, uv=uv, obj=obj, constraints=(ga, gb, dist, weights), points_true=pts, cameras_true=cams,
                cameras_init=init, n_frames=n_frames, n_per=n_per)


def board_volume(**kw):
    """The same scene as a CaptureVolume: the board is object 0, keypoint = corner index, one sync index per pose; the
    ConstraintSet comes from the grid compiler (truss + braces)."""
    import pandas as pd

    from caliscope_amd.capture_volume import CaptureVolume
    from caliscope_amd.constraints import ConstraintSet
    from caliscope_amd.point_data import ImagePoints, WorldPoints

    rows, cols, spacing = kw.get("rows", 3), kw.get("cols", 4), kw.get("spacing", 0.06)
    sc = board_scene(**kw)
    n_per = sc["n_per"]
    pts0 = sc["x0"][sc["par"].n_camera_params:].reshape(-1, 3)
    pt = np.arange(len(pts0))
    world = pd.DataFrame({"sync_index": pt // n_per, "object_id": 0, "keypoint_id": pt % n_per, "x_coord": pts0[:, 0], "y_coord": pts0[:, 1],
                          "z_coord": pts0[:, 2], "frame_time": (pt // n_per) * 0.1})
    img = pd.DataFrame({"sync_index": sc["obj"] // n_per, "cam_id": sc["cam"], "object_id": 0, "keypoint_id": sc["obj"] % n_per,
                        "img_loc_x": sc["uv"][:, 0], "img_loc_y": sc["uv"][:, 1]})
    grid = np.array([[c * spacing, r * spacing, 0.0] for r in range(rows) for c in range(cols)], dtype=np.float32)
    cs = ConstraintSet.from_grid(grid, spacing)
    return CaptureVolume(sc["cameras_init"], ImagePoints(img), WorldPoints(world), cs), sc


def marker_volume(n_cams=5, n_frames=12, n_markers=3, size=0.12, seed=5, noise_px=0.4):
    """Static square markers on the floor (objects 10, 11, ...: one world point per corner for the whole recording, every
    frame observes it again) plus one mobile marker (object 0) carried through the volume; the static markers are tied
    by a corner link (10 -> 11) and a centre link (11 -> 12)."""
    import pandas as pd

    from caliscope_amd.capture_volume import CaptureVolume
    from caliscope_amd.constraints import CentroidDistanceConstraint, ConstraintSet, DistanceConstraint
    from caliscope_amd.point_data import STATIC_SYNC_INDEX, ImagePoints, WorldPoints

    rng = np.random.default_rng(seed)
    cams = ring_camera_array(n_cams)
    h = size / 2
    square = np.array([[-h, h, 0], [h, h, 0], [h, -h, 0], [-h, -h, 0.0]])
    static_xyz = {10 + m: square @ rvec_to_matrix([0, 0, rng.uniform(0, 6)]).T + [rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), 0.3 + 0.1 * m]
                  for m in range(n_markers)}
    world, img = [], []

    def observe(si, oid, xyz):
        for c, cam in sorted(cams.cameras.items()):
            K = cam.matrix
            p, z = project_pinhole_bc5(xyz, cam.rotation, cam.translation, K[0, 0], K[1, 1], K[0, 2], K[1, 2], cam.distortions)
            for k in range(4):
                if z[k] > 0 and rng.random() < 0.9:
                    u = p[k] + rng.normal(0, noise_px, 2)
                    img.append(dict(sync_index=si, cam_id=c, object_id=oid, keypoint_id=k, img_loc_x=u[0], img_loc_y=u[1]))

    for oid, xyz in static_xyz.items():
        for k in range(4):
            w = xyz[k] + rng.normal(0, 0.01, 3)
            world.append(dict(sync_index=STATIC_SYNC_INDEX, object_id=oid, keypoint_id=k, x_coord=w[0], y_coord=w[1], z_coord=w[2], frame_time=np.nan))
    for si in range(n_frames):
        xyz = square @ rvec_to_matrix(rng.normal(0, 0.6, 3)).T + [rng.uniform(-0.4, 0.4), rng.uniform(-0.4, 0.4), rng.uniform(0.4, 1.0)]
        for k in range(4):
            w = xyz[k] + rng.normal(0, 0.01, 3)
            world.append(dict(sync_index=si, object_id=0, keypoint_id=k, x_coord=w[0], y_coord=w[1], z_coord=w[2], frame_time=si * 0.1))
        observe(si, 0, xyz)
        for oid, sxyz in static_xyz.items():
            observe(si, oid, sxyz)
    rows = [DistanceConstraint(o, i, o, j, float(np.linalg.norm(square[i] - square[j])), 0.002)
            for o in [0, *static_xyz] for i in range(4) for j in range(i + 1, 4)]
    rows.append(DistanceConstraint(10, 0, 11, 2, float(np.linalg.norm(static_xyz[10][0] - static_xyz[11][2])), 0.002))
    cent = (CentroidDistanceConstraint(11, 12, float(np.linalg.norm(static_xyz[11].mean(0) - static_xyz[12].mean(0))), 0.005),)
    cs = ConstraintSet(tuple(rows), frozenset(static_xyz), cent)
    init = CameraArray({c: CameraData(cam_id=c, size=cam.size, matrix=cam.matrix.copy(), distortions=cam.distortions.copy(),
                                      rotation=rvec_to_matrix(matrix_to_rvec(cam.rotation) + rng.normal(0, 0.01, 3)),
                                      translation=cam.translation + rng.normal(0, 0.02, 3)) for c, cam in cams.cameras.items()})
    vol = CaptureVolume(init, ImagePoints(pd.DataFrame(img)), WorldPoints(pd.DataFrame(world)), cs)
    par = BundleParameterization.from_camera_array(init, n_points=len(world), refine_intrinsics=False)
    return vol, par
