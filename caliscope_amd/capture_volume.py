"""``CaptureVolume.optimize()`` on the MI355X — host-side mirror of the reference's hot-path entry point.

Same public surface as the reference's ``core/capture_volume.py`` for the BA path:

* ``OptimizationStatus``                    <- ``capture_volume.py:45-67``
* ``CaptureVolume.optimize(ftol, max_nfev, verbose, strict, use_constraints, pixel_sigma, *,
  refine_intrinsics, loss, f_scale)``       <- ``:322-444`` (same kwargs, same defaults, immutable self,
  ``CalibrationError`` when ``strict`` and not converged)
* ``CaptureVolume.pixel_f_scale``           <- ``:141-148``
* ``CaptureVolume.reprojection_report``     <- ``:150-235`` (``overall_rmse = sqrt(mean(ex^2+ey^2))`` in pixels)
* ``CaptureVolume.filter_by_percentile_error`` (the stage between the product's passes,
  ``calibrate_extrinsics.py:244``)

The marshalling of DataFrames into flat arrays follows ``:346-358`` but is vectorised (the reference uses a
Python list comprehension over every observation, and a Python loop to build ``img_to_obj_map``).  The solver
call goes through :func:`caliscope_amd.least_squares.least_squares`, i.e. the MI355X engine; a volume's
``ConstraintSet`` becomes the constraint rows of the solve (``_build_constraint_arrays`` <- ``:446-516``, weights
``(pixel_sigma / f_median) / sigma`` <- ``:373-383``), ``rigidity_report`` <- ``:532-605``.
"""

from __future__ import annotations

import logging
from copy import deepcopy
from dataclasses import dataclass, field
from functools import cached_property

import numpy as np
import pandas as pd

from caliscope_amd.bundle_parameterization import BundleParameterization
from caliscope_amd.cameras import CameraArray
from caliscope_amd.engine import BAProblem
from caliscope_amd.exceptions import CalibrationError
from caliscope_amd.least_squares import least_squares
from caliscope_amd.constraints import ConstraintSet, ConstraintViolation, DistanceConstraint, RigidityReport
from caliscope_amd.point_data import STATIC_SYNC_INDEX, ImagePoints, WorldPoints
from caliscope_amd.engine import STATUS_REASONS

logger = logging.getLogger(__name__)

_KEY = ["sync_index", "object_id", "keypoint_id"]


@dataclass(frozen=True)
class OptimizationStatus:
    converged: bool
    termination_reason: str
    iterations: int  # number of residual evaluations (scipy's nfev)
    final_cost: float
    bound_warnings: tuple = ()


def _group_index(keys: np.ndarray):
    """``(unique_sorted_keys, inverse)`` like ``np.unique(keys, return_inverse=True)`` for integer keys.  Ids here
    (cameras, points, keypoints) are small and dense, so a presence table + prefix sum replaces the sort of every
    observation; widely spread keys fall back to ``np.unique``."""
    keys = np.asarray(keys, dtype=np.int64)
    if keys.size == 0:
        return keys, np.zeros(0, dtype=np.int64)
    lo, hi = int(keys.min()), int(keys.max())
    if hi - lo > 4 * keys.size + 1024:
        return np.unique(keys, return_inverse=True)
    shifted = keys - lo
    present = np.zeros(hi - lo + 1, dtype=bool)
    present[shifted] = True
    rank = np.cumsum(present, dtype=np.int64) - 1
    return np.flatnonzero(present) + lo, rank[shifted]


@dataclass(frozen=True)
class ReprojectionReport:
    overall_rmse: float
    by_camera: dict
    by_point: dict
    n_unmatched_observations: int
    unmatched_rate: float
    unmatched_by_camera: dict
    raw_errors: pd.DataFrame
    n_observations_matched: int
    n_observations_total: int
    n_cameras: int
    n_points: int


@dataclass(frozen=True)
class CaptureVolume:
    camera_array: CameraArray
    image_points: ImagePoints
    world_points: WorldPoints
    constraints: ConstraintSet | None = None
    img_to_obj_map: np.ndarray = field(init=False)
    _optimization_status: OptimizationStatus | None = field(default=None, compare=False)
    # optimize() changes coordinates only: the observation -> world-point map of the source volume stays valid (at 2M
    # observations rebuilding it is 30 % of the call)
    _known_map: np.ndarray | None = field(default=None, compare=False, repr=False)

    @property
    def optimization_status(self) -> OptimizationStatus | None:
        return self._optimization_status

    def __post_init__(self):
        known = self._known_map
        if known is not None and len(known) == len(self.image_points):
            object.__setattr__(self, "img_to_obj_map", known)
        else:
            object.__setattr__(self, "img_to_obj_map", self._compute_img_to_obj_map())
        object.__setattr__(self, "_known_map", None)
        n_img, n_world = len(self.image_points), len(self.world_points)
        if n_img == 0:
            raise ValueError("No image observations provided")
        if n_world == 0:
            raise ValueError("No world points provided")
        if len(self.camera_array.posed_cameras) == 0:
            raise ValueError("No posed cameras in array")
        n_matched = int(np.sum(self.img_to_obj_map >= 0))
        if n_matched == 0:
            raise ValueError("No image observations have corresponding world points")
        if n_matched < 2 * n_world:
            logger.warning(f"Suspicious geometry: {n_matched} matched observations for {n_world} world points. "
                           f"Expected at least {n_world * 2} for multi-view geometry.")
        if int(self.img_to_obj_map.max()) >= n_world:
            raise ValueError(f"obj_indices contains out-of-bounds index: {int(self.img_to_obj_map.max())} >= {n_world}")

    def _compute_img_to_obj_map(self) -> np.ndarray:
        """Row of ``world_points`` for every image observation, -1 when unmatched.  Observations of a static object look their
        point up at ``STATIC_SYNC_INDEX`` (reference :119-139, a left merge on (sync_index, object_id, keypoint_id) that keeps the LAST of duplicate
        world keys).  The three integer keys are folded into one and looked up in a dense table when their ranges are small (they are: frame
        numbers, a few objects, keypoint ids) — a fifth of the merge's time on 2M observations; spread-out keys take the merge."""
        wdf, idf = self.world_points._df, self.image_points._df
        static_ids = self.constraints.static_object_ids if self.constraints else frozenset()
        if len(wdf) and len(idf):
            w = [wdf[c].to_numpy() for c in _KEY]
            i = [idf[c].to_numpy() for c in _KEY]
            if static_ids:
                i[0] = np.where(np.isin(i[1], list(static_ids)), STATIC_SYNC_INDEX, i[0])
            lo = [min(int(a.min()), int(b.min())) for a, b in zip(w, i)]
            span = [max(int(a.max()), int(b.max())) - l + 1 for a, b, l in zip(w, i, lo)]
            if span[0] * span[1] * span[2] <= max(1 << 22, 16 * len(wdf)) and span[0] * span[1] * span[2] <= 1 << 25:  # (128 MB of int32 at most; wider key ranges take the merge)
                fold = lambda k: ((k[0] - lo[0]) * span[1] + (k[1] - lo[1])) * span[2] + (k[2] - lo[2])  # noqa: E731
                table = np.full(span[0] * span[1] * span[2], -1, dtype=np.int32)
                table[fold(w)] = np.arange(len(wdf), dtype=np.int32)  # (a repeated key keeps the last row written: numpy assigns in order)
                return table[fold(i)]
        return self._img_to_obj_map_by_merge(static_ids)

    def _img_to_obj_map_by_merge(self, static_ids) -> np.ndarray:
        world = self.world_points._df[_KEY].copy()
        world["world_idx"] = np.arange(len(world), dtype=np.int64)
        world = world.drop_duplicates(subset=_KEY, keep="last")
        keys = self.image_points._df[_KEY]
        if static_ids:
            keys = keys.copy()
            keys.loc[keys["object_id"].isin(list(static_ids)), "sync_index"] = STATIC_SYNC_INDEX
        merged = keys.merge(world, on=_KEY, how="left")
        return merged["world_idx"].fillna(-1).to_numpy(dtype=np.int32)

    # -- persistence (reference :237-267) -------------------------------------------------------------
    def save(self, directory) -> None:
        """Write ``camera_array.toml``, ``image_points.csv``, ``world_points.csv`` and, when the volume has one,
        ``constraints.toml`` (atomic writes).  As in the reference, ``optimization_status`` is not persisted."""
        from pathlib import Path

        directory = Path(directory)
        directory.mkdir(parents=True, exist_ok=True)
        self.camera_array.to_toml(directory / "camera_array.toml")
        self.image_points.to_csv(directory / "image_points.csv")
        self.world_points.to_csv(directory / "world_points.csv")
        if self.constraints is not None:
            self.constraints.to_toml(directory / "constraints.toml")

    @classmethod
    def load(cls, directory) -> "CaptureVolume":
        from pathlib import Path

        directory = Path(directory)
        con_path = directory / "constraints.toml"
        return cls(camera_array=CameraArray.from_toml(directory / "camera_array.toml"),
                   image_points=ImagePoints.from_csv(directory / "image_points.csv"),
                   world_points=WorldPoints.from_csv(directory / "world_points.csv"),
                   constraints=ConstraintSet.from_toml(con_path) if con_path.exists() else None)

    # -- marshalling (reference :346-358) ------------------------------------------------------------
    def _matched_arrays(self):
        col = self.image_points.arrays()
        index_of = self.camera_array.posed_cam_id_to_index
        # the volume is immutable apart from its camera array (a camera can be posed later): the arrays are kept per posed-camera set — optimize(),
        # the reprojection report and the filters of one volume marshal the same 2M rows (cfg4: 20 ms each time)
        key = tuple(sorted(index_of.items()))
        kept = getattr(self, "_matched_cache", None)
        if kept is not None and kept[0] == key:
            return kept[1]
        cam_id = col["cam_id"]
        # cam_id -> position among the posed cameras through a dense table (ids are small non-negative integers)
        lo = min(int(cam_id.min()), min(index_of, default=0)) if cam_id.size else 0
        hi = max(int(cam_id.max()), max(index_of, default=0)) if cam_id.size else 0
        if hi - lo < (1 << 20):
            table = np.full(hi - lo + 1, -1, dtype=np.int32)
            for cid, i in index_of.items():
                table[cid - lo] = i
            cam_idx = table[cam_id - lo]
        else:  # sparse ids: binary search in the sorted posed ids
            ids = np.array(sorted(index_of), dtype=np.int64)
            pos = np.array([index_of[c] for c in ids], dtype=np.int32)
            at = np.minimum(np.searchsorted(ids, cam_id), len(ids) - 1) if len(ids) else np.zeros(cam_id.shape, dtype=np.int64)
            cam_idx = np.where(ids[at] == cam_id, pos[at], -1).astype(np.int32) if len(ids) else np.full(cam_id.shape, -1, dtype=np.int32)
        mask = (self.img_to_obj_map >= 0) & (cam_idx >= 0)
        if mask.all():  # (the usual case after triangulation + filtering: nothing to select)
            camera_indices, obj_indices = cam_idx, self.img_to_obj_map.astype(np.int32)  # (a copy: the map itself stays writable)
            image_coords = np.empty((len(cam_idx), 2))
            image_coords[:, 0] = col["img_loc_x"]; image_coords[:, 1] = col["img_loc_y"]
        else:
            camera_indices = cam_idx[mask]
            image_coords = np.stack([col["img_loc_x"][mask], col["img_loc_y"][mask]], axis=1)
            obj_indices = self.img_to_obj_map[mask].astype(np.int32)
        for a in (mask, camera_indices, image_coords, obj_indices):
            a.setflags(write=False)  # shared between calls
        out = (mask, camera_indices, image_coords, obj_indices)
        object.__setattr__(self, "_matched_cache", (key, out))
        return out

    def pixel_f_scale(self, px: float = 1.0) -> float:
        focal = [cam.matrix[0, 0] for cam in self.camera_array.posed_cameras.values() if cam.matrix is not None]
        return px / float(np.median(focal))

    # -- the hot path ----------------------------------------------------------------------------------
    def optimize(
        self,
        ftol: float = 1e-8,
        max_nfev: int | None = None,
        verbose: int = 0,
        strict: bool = True,
        use_constraints: bool = True,
        pixel_sigma: float = 1.0,
        *,
        refine_intrinsics: bool = False,
        loss: str = "linear",
        f_scale: float = 1.0,
        _engine_factory=None,
    ) -> "CaptureVolume":
        """Bundle adjustment via pixel-space residuals, on the MI355X engine."""
        _, camera_indices, image_coords, obj_indices = self._matched_arrays()
        con_args = (None, None, None, None)
        if use_constraints and self.constraints is not None:
            arrays = self._build_constraint_arrays()
            if arrays is not None:
                groups_a, groups_b, distances, sigmas = arrays
                f_median = float(np.median([cam.matrix[0, 0] for cam in self.camera_array.posed_cameras.values()]))
                # residual units are normalised image coordinates (pixels / fx): a row is 1 when the distance is off
                # by one sigma scaled like a pixel_sigma reprojection error (reference :377-383)
                con_args = (groups_a, groups_b, distances, (pixel_sigma / f_median) / sigmas)
                logger.info(f"Adding {len(distances)} constraint rows (f_median={f_median:.0f}, pixel_sigma={pixel_sigma})")
        new_cameras = deepcopy(self.camera_array)
        par = BundleParameterization.from_camera_array(
            new_cameras, n_points=len(self.world_points), refine_intrinsics=refine_intrinsics
        )
        x0 = par.pack(new_cameras, self.world_points.points)
        logger.info(f"Beginning bundle adjustment on {len(image_coords)} observations")
        result = least_squares(
            None,
            x0,
            args=(par, camera_indices, image_coords, obj_indices, *con_args),
            jac=None,
            verbose=verbose,
            x_scale="jac",
            loss=loss,
            f_scale=f_scale,
            ftol=ftol,
            max_nfev=max_nfev,
            method="trf",
            bounds=par.bounds(),
            engine_factory=_engine_factory,
        )
        reason = STATUS_REASONS.get(result.status, f"unknown_{result.status}")
        converged = result.status in (1, 2, 3, 4)
        if strict and not converged:
            raise CalibrationError(
                f"Bundle adjustment did not converge: {reason}\n"
                f"Pass strict=False to suppress this error and inspect the result."
            )
        new_points = par.unpack_into(new_cameras, result.x)
        status = OptimizationStatus(
            converged=converged,
            termination_reason=reason,
            iterations=int(result.nfev),
            final_cost=float(result.cost),
            bound_warnings=par.bound_warnings(result.x),
        )
        out = CaptureVolume(
            camera_array=new_cameras,
            image_points=self.image_points,
            world_points=self.world_points.with_points(new_points),
            constraints=self.constraints,
            _optimization_status=status,
            _known_map=self.img_to_obj_map,
        )
        kept = getattr(self, "_constraint_cache", None)
        if kept is not None:  # same world-point keys, same constraint set
            object.__setattr__(out, "_constraint_cache", kept)
        return out

    # -- constraint rows (reference :446-605) ----------------------------------------------------------------
    def _constraint_blocks(self):
        """Per constraint that fires: ``(constraint, sync indices (n,), rows_a (n, 4), rows_b (n, 4))`` — four world-point rows per endpoint (a corner
        endpoint is its row four times) at every sync index at which all endpoint keypoints have a world point.  Static-static constraints fire once
        at ``STATIC_SYNC_INDEX``, mobile-mobile ones at every shared sync index, mixed ones never (reference ``_firing_sync_indices`` :518-530).
        The blocks depend on the KEYS of the world points and on the constraint set only, so they are built once per volume and handed to the
        volumes ``optimize()`` returns (same keys, new coordinates): on the reference's 4-camera session with its board they were 5 of the call's 7 ms."""
        kept = getattr(self, "_constraint_cache", None)
        if kept is not None:
            return kept
        con = self.constraints
        blocks = []
        if con is not None and (con.distances or con.centroid_distances):
            df = self.world_points._df
            order = np.lexsort((df["sync_index"].to_numpy(), df["keypoint_id"].to_numpy(), df["object_id"].to_numpy()))
            obj, kp, sync = (df[c].to_numpy()[order] for c in ("object_id", "keypoint_id", "sync_index"))
            # one slice of (sorted sync indices, world rows) per keypoint
            start = np.flatnonzero(np.r_[True, (obj[1:] != obj[:-1]) | (kp[1:] != kp[:-1])]) if len(obj) else np.array([], dtype=np.int64)
            end = np.r_[start[1:], len(obj)]
            table = {(int(obj[s]), int(kp[s])): (sync[s:e], order[s:e]) for s, e in zip(start, end)}
            static_ids = con.static_object_ids
            # One row table for the keypoints the constraints name: world row of (keypoint, sync index) or -1, over the sorted distinct sync indices.
            # A constraint fires where every endpoint has a row; all constraints of a kind are then ONE fancy-indexed comparison (an np.intersect1d
            # per endpoint pair was 1.6 of the 3.2 ms of a first optimize() on the reference's 4-camera session with its board).
            wanted = {(dc.object_id_a, dc.keypoint_id_a) for dc in con.distances} | {(dc.object_id_b, dc.keypoint_id_b) for dc in con.distances}
            wanted |= {(o, k) for cc in con.centroid_distances for o in (cc.object_id_a, cc.object_id_b) for k in range(4)}
            named = [key for key in wanted if key in table]
            usync = np.unique(np.concatenate([table[key][0] for key in named])) if named else np.array([], dtype=np.int64)
            row_of = {key: i for i, key in enumerate(named)}
            rowmat = np.full((len(named) + 1, len(usync)), -1, dtype=np.int64)  # last row: a keypoint without any world point
            for key, i in row_of.items():
                s_k, r_k = table[key]
                rowmat[i, np.searchsorted(usync, s_k)] = r_k  # (sorted by sync; of duplicate world keys the LAST row, as the reference's dict of rows keeps)
            is_static_col = usync == STATIC_SYNC_INDEX
            missing = len(named)

            def fire_all(constraints, endpoint_keys, split):
                """Append (constraint, syncs, rows_a, rows_b) for those of `constraints` that fire (`endpoint_keys(c)`: the keys of its endpoints;
                `split`: (instances, endpoints) world rows -> the two (instances, 4) row arrays)."""
                usable = [c for c in constraints if (c.object_id_a in static_ids) == (c.object_id_b in static_ids)]  # mixed static / mobile: never
                if not usable or not len(usync):
                    return
                idx = np.array([[row_of.get(key, missing) for key in endpoint_keys(c)] for c in usable], dtype=np.int64)  # (n, endpoints)
                rows = rowmat[idx]                                                                                          # (n, endpoints, syncs)
                static_c = np.array([c.object_id_a in static_ids for c in usable], dtype=bool)
                ok = (rows >= 0).all(axis=1) & (static_c[:, None] == is_static_col[None, :])
                ci, si = np.nonzero(ok)  # constraint-major, sync indices ascending inside a constraint: the order of the rows
                if not len(ci):
                    return
                rows_a, rows_b = split(rows[ci, :, si])
                syncs = usync[si]
                cut = np.searchsorted(ci, np.arange(len(usable) + 1)).tolist()
                for i, c in enumerate(usable):
                    if cut[i + 1] > cut[i]:
                        blocks.append((c, syncs[cut[i]:cut[i + 1]], rows_a[cut[i]:cut[i + 1]], rows_b[cut[i]:cut[i + 1]]))

            fire_all(con.distances, lambda c: ((c.object_id_a, c.keypoint_id_a), (c.object_id_b, c.keypoint_id_b)),
                     lambda r: (np.repeat(r[:, :1], 4, axis=1), np.repeat(r[:, 1:2], 4, axis=1)))
            fire_all(con.centroid_distances, lambda c: [(o, k) for o in (c.object_id_a, c.object_id_b) for k in range(4)], lambda r: (r[:, :4], r[:, 4:]))
        object.__setattr__(self, "_constraint_cache", blocks)
        return blocks

    def _constraint_instances(self):
        """``(constraint, sync_index, rows_a, rows_b)`` per instance, in the order of ``_build_constraint_arrays``'s rows."""
        for c, syncs, rows_a, rows_b in self._constraint_blocks():
            for i, si in enumerate(syncs):
                yield c, int(si), rows_a[i].tolist(), rows_b[i].tolist()

    def _build_constraint_arrays(self):
        """``(groups_a (n, 4) int32, groups_b (n, 4) int32, distances (n,), sigmas (n,))`` or None."""
        blocks = self._constraint_blocks()
        if not blocks:
            return None
        n = [len(b[1]) for b in blocks]
        return (np.concatenate([b[2] for b in blocks]).astype(np.int32), np.concatenate([b[3] for b in blocks]).astype(np.int32),
                np.repeat(np.array([b[0].distance for b in blocks], dtype=np.float64), n),
                np.repeat(np.array([b[0].sigma for b in blocks], dtype=np.float64), n))

    @property
    def unique_sync_indices(self) -> np.ndarray:
        """Sorted sync indices that have a world point (reference :933-941: the slider range of its viewers; STATIC_SYNC_INDEX included when
        static points exist, as there)."""
        return np.unique(self.world_points._df["sync_index"].to_numpy())

    def rigidity_report(self) -> RigidityReport:
        """Measured distance of every constraint instance with the current world points (no optimisation)."""
        xyz = self.world_points.points
        out = []
        for c, si, rows_a, rows_b in self._constraint_instances():
            actual = float(np.linalg.norm(xyz[rows_a].mean(axis=0) - xyz[rows_b].mean(axis=0)))
            if isinstance(c, DistanceConstraint):
                out.append(ConstraintViolation(c.object_id_a, c.keypoint_id_a, c.object_id_b, c.keypoint_id_b, si, c.distance, actual))
            else:
                out.append(ConstraintViolation(c.object_id_a, -1, c.object_id_b, -1, si, c.distance, actual, kind="centroid"))
        return RigidityReport(violations=tuple(out))

    # -- metric of record --------------------------------------------------------------------------------
    def _pixel_errors(self, camera_indices, image_coords, obj_indices, _engine_factory=None) -> np.ndarray:
        """(projected - observed) in pixels with the stored intrinsics/extrinsics, evaluated on the device:
        residuals are ``/fx_initial`` of a locked parameterization, so pixels = residual * fx."""
        par = BundleParameterization.from_camera_array(
            self.camera_array, n_points=len(self.world_points), refine_intrinsics=False
        )
        x = par.pack(self.camera_array, self.world_points.points)
        problem = BAProblem(par, camera_indices, image_coords, obj_indices)
        if _engine_factory is None:
            from caliscope_amd.hip_engine import HipEngine

            eng = HipEngine(problem, evaluation_only=True)  # the report needs residuals only: no Schur plan
        else:
            eng = _engine_factory(problem)
        try:
            r, _ = eng.residuals(x)
        finally:
            close = getattr(eng, "close", None)
            if close is not None:
                close()
        fx = np.array([b.fx_initial for b in par.blocks])[camera_indices]
        return r.reshape(-1, 2) * fx[:, None]

    def compute_reprojection_report(self, _engine_factory=None) -> ReprojectionReport:
        mask, camera_indices, image_coords, obj_indices = self._matched_arrays()
        n_total, n_matched = len(mask), int(mask.sum())
        if n_matched == 0:
            raise ValueError("No matched observations for reprojection error calculation")
        err = self._pixel_errors(camera_indices, image_coords, obj_indices, _engine_factory)
        sq = np.einsum("ij,ij->i", err, err)
        all_df = self.image_points._df
        # columns as arrays: a boolean-indexed DataFrame copy of 2M rows costs more than everything else in this function
        everything = n_matched == n_total  # (the usual case after triangulation and filtering: nothing to select)
        # (everything: to_numpy() hands out VIEWS of the ImagePoints table — copied, or writing to report.raw_errors would write through into the
        # table and the cached matched arrays)
        col = {c: (all_df[c].to_numpy().copy() if everything else all_df[c].to_numpy()[mask]) for c in ("sync_index", "cam_id", "object_id", "keypoint_id")}
        raw = pd.DataFrame(
            {
                "sync_index": col["sync_index"], "cam_id": col["cam_id"], "object_id": col["object_id"], "keypoint_id": col["keypoint_id"],
                "error_x": err[:, 0], "error_y": err[:, 1], "euclidean_error": np.sqrt(sq),
            },
            copy=False,  # (the arrays are this function's own: no need to copy them into consolidated blocks)
        )
        index_of = self.camera_array.posed_cam_id_to_index
        n_cam = len(index_of)
        cam_sum = np.bincount(camera_indices, weights=sq, minlength=n_cam)
        cam_cnt = np.bincount(camera_indices, minlength=n_cam)
        by_camera = {cid: 0.0 for cid in self.camera_array.posed_cameras}
        for cid, i in index_of.items():
            by_camera[cid] = float(np.sqrt(cam_sum[i] / cam_cnt[i])) if cam_cnt[i] else 0.0
        # RMS per (object_id, keypoint_id): one integer key, unique + bincount instead of a pandas group-by
        obj_id, kp_id = col["object_id"].astype(np.int64), col["keypoint_id"].astype(np.int64)
        kp_lo = int(kp_id.min()) if kp_id.size else 0
        span = int(kp_id.max()) - kp_lo + 1 if kp_id.size else 1
        keys, inv = _group_index(obj_id * span + (kp_id - kp_lo))
        mean_sq = np.bincount(inv, weights=sq, minlength=keys.size) / np.maximum(np.bincount(inv, minlength=keys.size), 1)
        by_point = dict(zip(zip((keys // span).tolist(), (keys % span + kp_lo).tolist()), np.sqrt(mean_sq).tolist()))
        cams_all, inv_all = _group_index(all_df["cam_id"].to_numpy())
        cams_ok, inv_ok = (cams_all, inv_all) if everything else _group_index(col["cam_id"])
        n_all, n_ok = np.bincount(inv_all, minlength=cams_all.size), np.bincount(inv_ok, minlength=cams_ok.size)
        total_by_cam, matched_by_cam = dict(zip(cams_all.tolist(), n_all.tolist())), dict(zip(cams_ok.tolist(), n_ok.tolist()))
        unmatched_by_camera = {
            int(c): int(total_by_cam.get(c, 0) - matched_by_cam.get(c, 0)) for c in self.camera_array.cameras
        }
        return ReprojectionReport(
            overall_rmse=float(np.sqrt(np.mean(sq))), by_camera=by_camera, by_point=by_point,
            n_unmatched_observations=n_total - n_matched, unmatched_rate=(n_total - n_matched) / n_total,
            unmatched_by_camera=unmatched_by_camera, raw_errors=raw, n_observations_matched=n_matched,
            n_observations_total=n_total, n_cameras=len(self.camera_array.posed_cameras), n_points=len(self.world_points),
        )

    @cached_property
    def reprojection_report(self) -> ReprojectionReport:
        return self.compute_reprojection_report()

    # -- outlier filtering between solver passes (reference capture_volume.py:607-753) ------------------
    def _filter_by_reprojection_thresholds(self, thresholds: dict, min_per_camera: int, _engine_factory=None, _report=None) -> "CaptureVolume":
        """Keep observations whose pixel error is <= their camera's threshold, but never fewer than
        ``min_per_camera`` per camera (the best ones are kept); world points left without any observation
        are pruned; the optimisation status is cleared."""
        report = _report if _report is not None else (self.compute_reprojection_report(_engine_factory) if _engine_factory else self.reprojection_report)
        raw = report.raw_errors
        err = raw["euclidean_error"].to_numpy()
        cam = raw["cam_id"].to_numpy()
        cams, inv = _group_index(cam)
        keep = err <= np.array([thresholds[int(c)] for c in cams], dtype=np.float64)[inv]
        kept_per_cam = np.bincount(inv, weights=keep, minlength=len(cams)).astype(np.int64)
        rows_per_cam = np.bincount(inv, minlength=len(cams))
        for k in np.flatnonzero((kept_per_cam < min_per_camera) & (kept_per_cam < rows_per_cam)):  # safety floor: rarely any
            idx = np.flatnonzero(inv == k)
            n_needed = min(min_per_camera, idx.size) - int(kept_per_cam[k])
            dropped = np.sort(err[idx][~keep[idx]])
            if dropped.size >= n_needed:
                keep[idx] = err[idx] <= dropped[n_needed - 1]
        mask, *_ = self._matched_arrays()
        keep_rows = np.zeros(len(mask), dtype=bool)  # like the reference's inner merge: unmatched rows go too
        keep_rows[np.flatnonzero(mask)[keep]] = True
        obj = self.img_to_obj_map[keep_rows]  # all >= 0: kept rows are matched rows
        seen = np.zeros(len(self.world_points), dtype=bool)
        seen[obj] = True
        # the surviving observations keep their world point, whose row number drops by the pruned rows before it: the new
        # volume gets its observation -> point map from the old one instead of a merge over every observation
        new_row = np.cumsum(seen, dtype=np.int64) - 1
        return CaptureVolume(self.camera_array, self.image_points.take(keep_rows), self.world_points.take(seen), self.constraints,
                             _known_map=new_row[obj].astype(np.int32))

    def filter_by_percentile_error(self, percentile: float, scope: str = "per_camera", min_per_camera: int = 10,
                                   _engine_factory=None) -> "CaptureVolume":
        """Remove the worst ``percentile`` percent of observations by reprojection error
        (``scope``: "per_camera" thresholds, the reference's default, or one "overall" threshold)."""
        if not (0 < percentile <= 100):
            raise ValueError(f"percentile must be between 0 and 100, got {percentile}")
        if min_per_camera < 1:
            raise ValueError(f"min_per_camera must be >= 1, got {min_per_camera}")
        if scope not in ("per_camera", "overall"):
            raise ValueError(f"scope must be 'per_camera' or 'overall', got {scope}")
        report = self.compute_reprojection_report(_engine_factory) if _engine_factory else self.reprojection_report
        raw = report.raw_errors
        keep_percentile = 100 - percentile
        if scope == "per_camera":
            # one stable sort by camera instead of a boolean selection over every observation per camera
            err, cam = raw["euclidean_error"].to_numpy(), raw["cam_id"].to_numpy()
            cams, inv = _group_index(cam)
            # numpy's stable sort of 16-bit keys is a radix sort
            order = np.argsort(inv.astype(np.int16) if cams.size < 32768 else inv, kind="stable")
            start = np.concatenate([[0], np.cumsum(np.bincount(inv, minlength=cams.size))[:-1]]).astype(np.int64)
            stop = np.append(start[1:], len(order))
            thresholds = {cam_id: float(np.inf) for cam_id in self.camera_array.posed_cameras}
            for c, a, b in zip(cams.tolist(), start.tolist(), stop.tolist()):
                if c in thresholds:
                    thresholds[c] = float(np.percentile(err[order[a:b]], keep_percentile))
        else:
            thr = float(np.percentile(raw["euclidean_error"], keep_percentile))
            thresholds = {cam_id: thr for cam_id in self.camera_array.posed_cameras}
        return self._filter_by_reprojection_thresholds(thresholds, min_per_camera, _engine_factory, _report=report)  # one report for both steps

    def filter_by_absolute_error(self, max_pixels: float, min_per_camera: int = 10, _engine_factory=None) -> "CaptureVolume":
        """Remove observations with a reprojection error above ``max_pixels`` (reference :687-707)."""
        if max_pixels <= 0:
            raise ValueError(f"max_pixels must be positive, got {max_pixels}")
        if min_per_camera < 1:
            raise ValueError(f"min_per_camera must be >= 1, got {min_per_camera}")
        thresholds = {cam_id: float(max_pixels) for cam_id in self.camera_array.posed_cameras}
        return self._filter_by_reprojection_thresholds(thresholds, min_per_camera, _engine_factory)

    @classmethod
    def bootstrap(cls, image_points: ImagePoints, camera_array: CameraArray, constraints=None, *, _triangulate=None) -> "CaptureVolume":
        """Starting volume for ``optimize`` from 2-D observations (reference ``:268-320``): copy the cameras, triangulate
        every point seen by two or more posed cameras (on the device), keep the input untouched.

        The reference first estimates the poses from pairwise PnP / essential-matrix decompositions (OpenCV, upstream of
        the solver path and not rebuilt here): this ``bootstrap`` takes the poses the cameras already carry — a previous
        calibration, a rig description, another tool's estimate — and raises ``CalibrationError`` for cameras that have
        observations but no pose.  Validation and errors otherwise follow the reference (``:288-307``)."""
        point_cams = set(int(c) for c in image_points.df["cam_id"].unique())
        missing = point_cams - set(camera_array.cameras)
        if missing:
            raise CalibrationError(f"ImagePoints reference cameras {missing} not in the CameraArray.")
        uncalibrated = [cid for cid, cam in camera_array.cameras.items() if cam.matrix is None or cam.distortions is None]
        if uncalibrated:
            raise CalibrationError(
                f"Cannot run extrinsic calibration -- cameras {uncalibrated} have no intrinsic calibration.\n\n"
                f"Run calibrate_intrinsics() for each camera first."
            )
        unposed = sorted(cid for cid in point_cams
                         if not camera_array.cameras[cid].ignore
                         and (camera_array.cameras[cid].rotation is None or camera_array.cameras[cid].translation is None))
        if unposed:
            raise CalibrationError(
                f"Cameras {unposed} have observations but no pose estimate. This backend refines poses; the initial pose "
                f"network (PnP / essential matrix) is upstream of it: load a previous calibration or supply estimates."
            )
        cameras = deepcopy(camera_array)
        static_ids = constraints.static_object_ids if constraints else frozenset()
        triangulate = _triangulate or (lambda ip, cams, static: ip.triangulate(cams, static_object_ids=static))
        world_points = triangulate(image_points, cameras, static_ids)
        return cls(camera_array=cameras, image_points=image_points, world_points=world_points, constraints=constraints)

    @classmethod
    def from_arrays(cls, camera_array: CameraArray, camera_ids, image_coords, obj_indices, points_xyz) -> "CaptureVolume":
        """Build a volume from flat arrays (synthetic scenes): observation i is keypoint 0 of object
        ``obj_indices[i]`` at sync_index 0 — one world point per object."""
        obj_indices = np.asarray(obj_indices)
        n_pts = len(points_xyz)
        img = pd.DataFrame(
            {
                "sync_index": 0, "cam_id": np.asarray(camera_ids), "object_id": obj_indices, "keypoint_id": 0,
                "img_loc_x": np.asarray(image_coords)[:, 0], "img_loc_y": np.asarray(image_coords)[:, 1],
            }
        )
        pts = np.asarray(points_xyz, dtype=np.float64)
        world = pd.DataFrame(
            {"sync_index": 0, "object_id": np.arange(n_pts), "keypoint_id": 0, "x_coord": pts[:, 0], "y_coord": pts[:, 1], "z_coord": pts[:, 2]}
        )
        return cls(camera_array, ImagePoints(img), WorldPoints(world))
