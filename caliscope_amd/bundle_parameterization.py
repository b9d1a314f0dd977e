"""Parameter-vector layout of the bundle-adjustment problem.

Host-side mirror of the reference's ``core/bundle_parameterization.py`` — same class and
method names, same argument meaning, same error behaviour — because this layout is the
contract between ``CaptureVolume.optimize`` and the solver:

* camera ``i`` occupies ``x[off_i : off_i + n_i]`` with ``[rvec(3), tvec(3)]`` and, when its
  intrinsics are free, ``[s, k1, k2]`` (``fx = s*fx0``, ``fy = s*fy0``; ``p1, p2, k3`` stay fixed);
* fisheye cameras (4 equidistant coefficients) are always 6 wide;
* cameras are ordered by ``CameraArray.posed_index_to_cam_id``;
* world points follow at ``x[n_camera_params:]`` as row-major ``(P, 3)``.

Reference: ``bundle_parameterization.py:36-51`` (CameraBlock), ``:59-112`` (from_camera_array),
``:114-149`` (offsets/pack/unpack), ``:151-164`` (bounds), ``:166-186`` (trial inputs),
``:232-260`` (bound warnings).  New here: :meth:`BundleParameterization.device_tables`, the flat
tables the C ABI (``include/caliscope_ba.h``) takes.
"""

from __future__ import annotations

from dataclasses import dataclass
from functools import cached_property

import numpy as np

from caliscope_amd.exceptions import CalibrationError

N_EXTRINSIC_PARAMS = 6
N_FREE_INTRINSIC_PARAMS = 3

MODEL_PINHOLE_BC5 = 0
MODEL_FISHEYE4 = 1

# (name, lower, upper, proximity, relative?) for the free intrinsics at block offset 6,7,8
_FREE_SPEC = (("f", 0.5, 2.0, 0.01, True), ("k1", -1.0, 1.0, 0.01, False), ("k2", -2.0, 2.0, 0.01, False))


@dataclass(frozen=True)
class BoundWarning:
    cam_id: int
    parameter: str  # "f" | "k1" | "k2"
    bound: str  # "lower" | "upper"
    value: float


@dataclass(frozen=True)
class IntrinsicEstimate:
    cam_id: int
    f_recovered: float
    k1_recovered: float
    k2_recovered: float
    f_initial: float
    k1_initial: float
    k2_initial: float


@dataclass(frozen=True)
class CameraBlock:
    cam_id: int
    free_intrinsics: bool
    fx_initial: float
    fy_initial: float
    cx: float
    cy: float
    fisheye: bool
    dist_fixed: tuple[float, ...]
    k1_initial: float = 0.0
    k2_initial: float = 0.0

    @property
    def n_params(self) -> int:
        return N_EXTRINSIC_PARAMS + (N_FREE_INTRINSIC_PARAMS if self.free_intrinsics else 0)

    def _K(self, s: float = 1.0) -> np.ndarray:
        return np.array([[s * self.fx_initial, 0.0, self.cx], [0.0, s * self.fy_initial, self.cy], [0.0, 0.0, 1.0]])


@dataclass(frozen=True)
class BundleParameterization:
    blocks: tuple[CameraBlock, ...]
    n_points: int

    @classmethod
    def from_camera_array(cls, camera_array, n_points: int, *, refine_intrinsics: bool) -> "BundleParameterization":
        order = camera_array.posed_index_to_cam_id
        blocks = []
        for idx in sorted(order):
            cam_id = order[idx]
            cam = camera_array.cameras[cam_id]
            if cam.matrix is None or cam.distortions is None:
                raise CalibrationError(
                    f"Camera {cam_id} has no intrinsics. "
                    f"Run intrinsic calibration or synthesize defaults before optimizing."
                )
            K = np.asarray(cam.matrix, dtype=np.float64)
            coeffs = np.asarray(cam.distortions, dtype=np.float64).ravel()
            common = dict(
                cam_id=cam_id, fx_initial=float(K[0, 0]), fy_initial=float(K[1, 1]), cx=float(K[0, 2]), cy=float(K[1, 2])
            )
            if cam.fisheye:
                if coeffs.size != 4:
                    raise CalibrationError(
                        f"Fisheye camera {cam_id} requires exactly 4 distortion coefficients "
                        f"(equidistant model), got {coeffs.size}."
                    )
                blocks.append(CameraBlock(free_intrinsics=False, fisheye=True, dist_fixed=tuple(coeffs), **common))
            else:
                # Brown-Conrady [k1, k2, p1, p2, k3]: k1, k2 may be refined, the rest is carried along
                blocks.append(
                    CameraBlock(
                        free_intrinsics=bool(refine_intrinsics),
                        fisheye=False,
                        dist_fixed=tuple(coeffs[2:5]),
                        k1_initial=float(coeffs[0]),
                        k2_initial=float(coeffs[1]),
                        **common,
                    )
                )
        return cls(blocks=tuple(blocks), n_points=int(n_points))

    # -- layout --------------------------------------------------------------------------------
    @cached_property
    def camera_param_offsets(self) -> tuple[int, ...]:
        widths = [b.n_params for b in self.blocks]
        return tuple(int(v) for v in np.concatenate([[0], np.cumsum(widths)[:-1]])) if widths else ()

    @cached_property
    def n_camera_params(self) -> int:
        return int(sum(b.n_params for b in self.blocks))

    @property
    def n_params(self) -> int:
        return self.n_camera_params + 3 * self.n_points

    def pack(self, camera_array, world_points_xyz) -> np.ndarray:
        x = np.empty(self.n_params)
        for blk, off in zip(self.blocks, self.camera_param_offsets):
            cam = camera_array.cameras[blk.cam_id]
            x[off : off + 6] = cam.extrinsics_to_vector()
            if blk.free_intrinsics:
                coeffs = np.asarray(cam.distortions, dtype=np.float64).ravel()
                x[off + 6 : off + 9] = (1.0, coeffs[0], coeffs[1])
        x[self.n_camera_params :] = np.asarray(world_points_xyz, dtype=np.float64).reshape(-1)
        return x

    def unpack_into(self, camera_array, x) -> np.ndarray:
        x = np.asarray(x, dtype=np.float64)
        for blk, off in zip(self.blocks, self.camera_param_offsets):
            cam = camera_array.cameras[blk.cam_id]
            cam.extrinsics_from_vector(x[off : off + 6])
            if blk.free_intrinsics:
                s, k1, k2 = (float(v) for v in x[off + 6 : off + 9])
                cam.matrix = blk._K(s)
                cam.distortions = np.array([k1, k2, *blk.dist_fixed])
        return x[self.n_camera_params :].reshape(-1, 3)

    def bounds(self) -> tuple[np.ndarray, np.ndarray]:
        lo = np.full(self.n_params, -np.inf)
        hi = np.full(self.n_params, np.inf)
        for blk, off in zip(self.blocks, self.camera_param_offsets):
            if blk.free_intrinsics:
                for j, (_, lb, ub, _, _) in enumerate(_FREE_SPEC):
                    lo[off + 6 + j] = lb
                    hi[off + 6 + j] = ub
        return lo, hi

    @property
    def has_finite_bounds(self) -> bool:
        return any(b.free_intrinsics for b in self.blocks)

    def trial_projection_inputs(self, x, block_index: int):
        """``(rvec, tvec, K, dist)`` of one camera at trial point ``x``."""
        blk = self.blocks[block_index]
        off = self.camera_param_offsets[block_index]
        rvec, tvec = x[off : off + 3], x[off + 3 : off + 6]
        if blk.free_intrinsics:
            s, k1, k2 = x[off + 6 : off + 9]
            return rvec, tvec, blk._K(s), np.array([k1, k2, *blk.dist_fixed])
        if blk.fisheye:
            return rvec, tvec, blk._K(), np.array(blk.dist_fixed)
        return rvec, tvec, blk._K(), np.array([blk.k1_initial, blk.k2_initial, *blk.dist_fixed])

    def sparsity(self, camera_indices, obj_indices, n_constraints: int = 0, constraint_groups_a=None, constraint_groups_b=None):
        """Structure of the Jacobian as a ``scipy.sparse.lil_matrix`` of 0/1 (reference ``:188-230``): a reprojection row
        pair touches its camera's block and its point's three columns, a constraint row the coordinates of every point of
        both endpoint groups.  The solver here never needs it (the structure is implicit in the point-sorted records); it is
        kept for callers that hand ``jac_sparsity`` to scipy and as the contract the kernels' Jacobian is tested against."""
        from scipy.sparse import coo_matrix

        cam = np.asarray(camera_indices, dtype=np.int64)
        obj = np.asarray(obj_indices, dtype=np.int64)
        n_obs = cam.shape[0]
        n_rows = 2 * n_obs + int(n_constraints)
        n_cols = self.n_camera_params + 3 * self.n_points
        offsets = np.asarray(self.camera_param_offsets, dtype=np.int64)
        widths = np.array([b.n_params for b in self.blocks], dtype=np.int64)
        rows, cols = [], []
        obs = np.arange(n_obs, dtype=np.int64)
        w_obs = widths[cam] if n_obs else np.zeros(0, dtype=np.int64)
        for p in range(int(widths.max()) if len(widths) else 0):
            has = obs[w_obs > p]
            for half in (0, 1):
                rows.append(2 * has + half)
                cols.append(offsets[cam[has]] + p)
        for coord in range(3):
            for half in (0, 1):
                rows.append(2 * obs + half)
                cols.append(self.n_camera_params + 3 * obj + coord)
        if constraint_groups_a is not None and constraint_groups_b is not None and n_constraints > 0:
            c_idx = 2 * n_obs + np.arange(int(n_constraints), dtype=np.int64)
            for groups in (np.asarray(constraint_groups_a, dtype=np.int64), np.asarray(constraint_groups_b, dtype=np.int64)):
                for col in range(groups.shape[1]):
                    for coord in range(3):
                        rows.append(c_idx)
                        cols.append(self.n_camera_params + 3 * groups[:, col] + coord)
        r = np.concatenate(rows) if rows else np.zeros(0, dtype=np.int64)
        c = np.concatenate(cols) if cols else np.zeros(0, dtype=np.int64)
        pattern = coo_matrix((np.ones(r.size, dtype=np.int8), (r, c)), shape=(n_rows, n_cols)).tocsr()
        pattern.data[:] = 1  # repeated endpoint rows (corner groups) mark a column once
        return pattern.astype(int).tolil()

    def bound_warnings(self, x) -> tuple[BoundWarning, ...]:
        found = []
        for blk, off in zip(self.blocks, self.camera_param_offsets):
            if not blk.free_intrinsics:
                continue
            for j, (name, lb, ub, prox, relative) in enumerate(_FREE_SPEC):
                v = float(x[off + 6 + j])
                shown = v * blk.fx_initial if name == "f" else v
                for side, bound in (("lower", lb), ("upper", ub)):
                    tol = prox * abs(bound) if relative else prox
                    if abs(v - bound) <= tol:
                        found.append(BoundWarning(blk.cam_id, name, side, shown))
        return tuple(found)

    def intrinsic_estimates(self, camera_array) -> tuple[IntrinsicEstimate, ...]:
        out = []
        for blk in self.blocks:
            if blk.free_intrinsics:
                cam = camera_array.cameras[blk.cam_id]
                out.append(
                    IntrinsicEstimate(
                        blk.cam_id,
                        float(cam.matrix[0, 0]),
                        float(cam.distortions[0]),
                        float(cam.distortions[1]),
                        blk.fx_initial,
                        blk.k1_initial,
                        blk.k2_initial,
                    )
                )
        return tuple(out)

    # -- tables for the C ABI ------------------------------------------------------------------
    def device_tables(self) -> dict[str, np.ndarray]:
        return device_tables(self)


def n_params_of(parameterization) -> int:
    """Length of the parameter vector (works for the reference's own BundleParameterization too)."""
    return int(parameterization.n_camera_params) + 3 * int(parameterization.n_points)


def device_tables(parameterization) -> dict[str, np.ndarray]:
    """Flat per-camera tables consumed by ``cba_create`` (include/caliscope_ba.h).

    Duck-typed on ``blocks`` / ``camera_param_offsets`` so that the reference's own
    ``caliscope.core.bundle_parameterization.BundleParameterization`` can be passed straight through the
    ``least_squares`` seam.  ``cam_const[c] = [fx0, fy0, cx, cy, d0..d4, 0, 0, 0]`` where ``d`` is
    ``[k1, k2, p1, p2, k3]`` (pinhole; k1,k2 are the *initial* values, overridden by x when free) or
    ``[k1, k2, k3, k4, 0]`` (fisheye).
    """
    blocks = parameterization.blocks
    C = len(blocks)
    n_params = np.zeros(C, dtype=np.int32)
    model = np.zeros(C, dtype=np.int32)
    const = np.zeros((C, 12), dtype=np.float64)
    for i, blk in enumerate(blocks):
        n_params[i] = blk.n_params
        model[i] = MODEL_FISHEYE4 if blk.fisheye else MODEL_PINHOLE_BC5
        const[i, 0:4] = (blk.fx_initial, blk.fy_initial, blk.cx, blk.cy)
        if blk.fisheye:
            const[i, 4:8] = blk.dist_fixed
        else:
            const[i, 4:9] = (blk.k1_initial, blk.k2_initial, *blk.dist_fixed)
    return {
        "cam_n_params": n_params,
        "cam_model": model,
        "cam_const": const,
        "cam_offsets": np.asarray(parameterization.camera_param_offsets, dtype=np.int32),
    }
