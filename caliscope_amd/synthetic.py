"""Vectorised synthetic BA scenes at the sizes of BASELINE.json's configs.

The reference's generator (``synthetic/``: ``CameraSynthesizer.add_ring``, ``SyntheticScene``)
walks frame x camera x point in Python and calls ``cv2.projectPoints``; it cannot produce
millions of observations.  This module follows its *recipe* with numpy array code
(SURVEY.md §8d):

* cameras: inward-facing rings of <= 16 (``camera_synthesizer.py:134-199``), radius 3 m, heights
  0.5 + 0.75*ring, rings staggered by half an angular step, looking at the centre of the point
  cloud, OpenCV convention ``t = -R @ position`` (``:304``), WEBCAM lens (``:23-27``);
* points: uniform in a cylinder r <= 0.6 m, z in [0, 1.2] m;
* each point is observed by exactly ``k = N/P`` cameras drawn without replacement among those that
  see it in front and in frame; pixel noise N(0, 0.5) (``synthetic_scene.py:47``);
* optional gross outliers, 10-50 px in a random direction (``synthetic/outliers.py:14-54``);
* initial guess: ground truth perturbed (rvec 0.01 rad, tvec 0.02 m, points 0.01 m), optional
  intrinsic perturbation ``f*1.03, k1+0.02, k2+0.05`` (``camera_synthesizer.py:42-45``).

Seeds: data 42, init 43, outliers 44.
"""

from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from caliscope_amd.cameras import CameraArray, CameraData, matrix_to_rvec, rvec_to_matrix

WEBCAM_FOCAL = 1394.6
WEBCAM_DIST = (0.115, -0.219, 0.0012, 0.0086, 0.113)
WEBCAM_SIZE = (1920, 1080)

CONFIGS = {
    # name: cameras, points, observations, loss, outlier fraction, refine intrinsics
    "cfg2": dict(n_cams=8, n_points=5_000, n_obs=40_000, loss="linear", outliers=0.0, refine=False),
    "cfg3": dict(n_cams=32, n_points=50_000, n_obs=400_000, loss="huber", outliers=0.05, refine=False),
    "cfg4": dict(n_cams=64, n_points=200_000, n_obs=2_000_000, loss="linear", outliers=0.0, refine=False),
    "cfg5": dict(n_cams=128, n_points=1_000_000, n_obs=10_000_000, loss="linear", outliers=0.0, refine=True),
    # not a BASELINE config: the size the CPU test build of the ABI can solve, so that bench.py's own code paths run in the CPU suite
    "tiny": dict(n_cams=5, n_points=90, n_obs=450, loss="linear", outliers=0.0, refine=False),
}


def project_pinhole_bc5(X, R, t, fx, fy, cx, cy, dist):
    """Forward pinhole + Brown-Conrady projection of (n,3) world points (data generation only)."""
    Xc = X @ R.T + t
    z = Xc[:, 2]
    x, y = Xc[:, 0] / z, Xc[:, 1] / z
    k1, k2, p1, p2, k3 = dist
    r2 = x * x + y * y
    radial = 1.0 + r2 * (k1 + r2 * (k2 + r2 * k3))
    xd = x * radial + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
    yd = y * radial + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
    return np.stack([fx * xd + cx, fy * yd + cy], axis=1), z


def _look_at(position, target):
    fwd = target - position
    fwd = fwd / np.linalg.norm(fwd)
    right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
    right = right / np.linalg.norm(right)
    down = np.cross(fwd, right)
    return np.vstack([right, down / np.linalg.norm(down), fwd])


def ring_camera_array(n_cams: int, radius: float = 3.0, target=(0.0, 0.0, 0.6)) -> CameraArray:
    cams = {}
    w, h = WEBCAM_SIZE
    K = np.array([[WEBCAM_FOCAL, 0, w / 2.0], [0, WEBCAM_FOCAL, h / 2.0], [0, 0, 1.0]])
    n_rings = -(-n_cams // 16)
    cam_id = 0
    for ring in range(n_rings):
        in_ring = min(16, n_cams - 16 * ring)
        height = 0.5 + 0.75 * ring
        for i in range(in_ring):
            ang = 2 * np.pi * (i + 0.5 * (ring % 2)) / in_ring
            pos = np.array([radius * np.cos(ang), radius * np.sin(ang), height])
            R = _look_at(pos, np.asarray(target, dtype=np.float64))
            cams[cam_id] = CameraData(
                cam_id=cam_id, size=WEBCAM_SIZE, matrix=K.copy(), distortions=np.array(WEBCAM_DIST),
                rotation=R, translation=-R @ pos,
            )
            cam_id += 1
    return CameraArray(cams)


@dataclass
class SyntheticBA:
    name: str
    cameras_true: CameraArray
    cameras_init: CameraArray
    points_true: np.ndarray  # (P, 3)
    points_init: np.ndarray  # (P, 3)
    camera_indices: np.ndarray  # (N,) int32
    image_coords: np.ndarray  # (N, 2) float64
    obj_indices: np.ndarray  # (N,) int32
    loss: str
    refine_intrinsics: bool
    outlier_rows: np.ndarray

    @property
    def n_obs(self) -> int:
        return int(self.camera_indices.shape[0])

    def f_scale_1px(self) -> float:
        """``CaptureVolume.pixel_f_scale(1.0)``: one pixel in the normalised residual units."""
        fl = [c.matrix[0, 0] for c in self.cameras_init.posed_cameras.values()]
        return 1.0 / float(np.median(fl))


def make_scene(
    name: str = "custom",
    *,
    n_cams: int,
    n_points: int,
    n_obs: int,
    loss: str = "linear",
    outliers: float = 0.0,
    refine: bool = False,
    pixel_sigma: float = 0.5,
    seed: int = 42,
    chunk: int = 131072,
    shard: int = 0,
) -> SyntheticBA:
    """``shard`` > 0 draws a different set of points / observations (and their initial perturbation) for the
    same cameras and the same camera perturbation: rank ``shard`` of a weak-scaling run owns that set."""
    if n_obs % n_points:
        raise ValueError("n_obs must be a multiple of n_points (each point is seen by exactly n_obs/n_points cameras)")
    k = n_obs // n_points
    cams_true = ring_camera_array(n_cams)
    rng = np.random.default_rng(seed if shard == 0 else [seed, shard])
    rad = 0.6 * np.sqrt(rng.uniform(0, 1, n_points))
    ang = rng.uniform(0, 2 * np.pi, n_points)
    pts = np.stack([rad * np.cos(ang), rad * np.sin(ang), rng.uniform(0, 1.2, n_points)], axis=1)

    w, h = WEBCAM_SIZE
    cam_list = [cams_true.cameras[c] for c in sorted(cams_true.cameras)]
    cam_sel = np.empty((n_points, k), dtype=np.int32)
    uv_sel = np.empty((n_points, k, 2))
    for lo in range(0, n_points, chunk):
        hi = min(n_points, lo + chunk)
        m = hi - lo
        keys = rng.random((m, n_cams), dtype=np.float32)
        uv_all = np.empty((m, n_cams, 2))
        for ci, cam in enumerate(cam_list):
            K = cam.matrix
            uv, z = project_pinhole_bc5(pts[lo:hi], cam.rotation, cam.translation, K[0, 0], K[1, 1], K[0, 2], K[1, 2],
                                        cam.distortions)
            uv_all[:, ci] = uv
            vis = (z > 0) & (uv[:, 0] >= 0) & (uv[:, 0] < w) & (uv[:, 1] >= 0) & (uv[:, 1] < h)
            keys[~vis, ci] = 2.0
        pick = np.argpartition(keys, k - 1, axis=1)[:, :k]
        pick.sort(axis=1)
        if np.any(np.take_along_axis(keys, pick, axis=1) >= 2.0):
            raise ValueError(f"some points are visible in fewer than k={k} cameras")
        cam_sel[lo:hi] = pick
        uv_sel[lo:hi] = np.take_along_axis(uv_all, pick[:, :, None], axis=1)
    image_coords = uv_sel.reshape(-1, 2) + rng.normal(0, pixel_sigma, (n_obs, 2))
    camera_indices = cam_sel.reshape(-1).astype(np.int32)
    obj_indices = np.repeat(np.arange(n_points, dtype=np.int32), k)

    outlier_rows = np.array([], dtype=np.int64)
    if outliers > 0:
        orng = np.random.default_rng(seed + 2 if shard == 0 else [seed + 2, shard])
        n_bad = round(outliers * n_obs)
        outlier_rows = np.sort(orng.choice(n_obs, size=n_bad, replace=False))
        mag = orng.uniform(10.0, 50.0, n_bad)
        th = orng.uniform(0, 2 * np.pi, n_bad)
        image_coords[outlier_rows, 0] += mag * np.cos(th)
        image_coords[outlier_rows, 1] += mag * np.sin(th)

    irng = np.random.default_rng(seed + 1)
    cams_init = {}
    for cid in sorted(cams_true.cameras):
        c = cams_true.cameras[cid]
        rvec = matrix_to_rvec(c.rotation) + irng.normal(0, 0.01, 3)
        K = c.matrix.copy()
        dist = c.distortions.copy()
        if refine:
            K[0, 0] *= 1.03
            K[1, 1] *= 1.03
            dist[0] += 0.02
            dist[1] += 0.05
        cams_init[cid] = CameraData(
            cam_id=cid, size=c.size, matrix=K, distortions=dist, rotation=rvec_to_matrix(rvec),
            translation=c.translation + irng.normal(0, 0.02, 3),
        )
    prng = irng if shard == 0 else np.random.default_rng([seed + 1, shard])
    pts_init = pts + prng.normal(0, 0.01, pts.shape)
    return SyntheticBA(
        name=name, cameras_true=cams_true, cameras_init=CameraArray(cams_init), points_true=pts, points_init=pts_init,
        camera_indices=camera_indices, image_coords=image_coords, obj_indices=obj_indices, loss=loss,
        refine_intrinsics=refine, outlier_rows=outlier_rows,
    )


def make_config(name: str, **overrides) -> SyntheticBA:
    cfg = dict(CONFIGS[name])
    cfg.update(overrides)
    return make_scene(name, **cfg)
