"""Camera containers for the BA path.

Mirrors the part of the reference's ``cameras/camera_array.py`` that the hot path touches:
``CameraData`` fields (``camera_array.py:18-41``), ``extrinsics_to_vector`` /
``extrinsics_from_vector`` (``:115-133``), the posed / non-ignored ordering that fixes the
parameter-vector layout (``:240-272``) and the ``camera_array.toml`` reader (``:377-441``).
The reference delegates Rodrigues conversions to OpenCV; here they are a few lines of numpy
(host side only, O(cameras)) so that no OpenCV is required.

Convention: ``rotation`` / ``translation`` map world -> camera, ``X_cam = R @ X_world + t``.
"""

from __future__ import annotations

from copy import deepcopy as _generic_deepcopy
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict

import numpy as np

from caliscope_amd.persistence import PersistenceError, safe_write_toml

_TINY = np.finfo(np.float64).eps


def rvec_to_matrix(rvec) -> np.ndarray:
    """Axis-angle -> rotation matrix (what ``cv2.Rodrigues(rvec)[0]`` returns)."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    angle = float(np.sqrt(r @ r))
    if angle < _TINY:
        return np.eye(3)
    ax = r / angle
    ca, sa = np.cos(angle), np.sin(angle)
    cross = np.array([[0.0, -ax[2], ax[1]], [ax[2], 0.0, -ax[0]], [-ax[1], ax[0], 0.0]])
    return ca * np.eye(3) + (1.0 - ca) * np.outer(ax, ax) + sa * cross


def matrix_to_rvec(R) -> np.ndarray:
    """Rotation matrix -> axis-angle (what ``cv2.Rodrigues(R)[0].ravel()`` returns)."""
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    u, _, vt = np.linalg.svd(R)  # OpenCV projects onto SO(3) first
    R = u @ vt
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    half_sin = 0.5 * float(np.sqrt(w @ w))
    cos_a = min(1.0, max(-1.0, 0.5 * (float(np.trace(R)) - 1.0)))
    angle = float(np.arccos(cos_a))
    if half_sin >= 1e-5:
        return w * (angle / (2.0 * half_sin))
    if cos_a > 0.0:
        return np.zeros(3)
    # angle ~ pi: read the axis off the diagonal, signs from the off-diagonals
    ax = np.sqrt(np.maximum((np.diag(R) + 1.0) * 0.5, 0.0))
    if R[0, 1] < 0:
        ax[1] = -ax[1]
    if R[0, 2] < 0:
        ax[2] = -ax[2]
    if abs(ax[0]) < abs(ax[1]) and abs(ax[0]) < abs(ax[2]) and ((R[1, 2] > 0) != (ax[1] * ax[2] > 0)):
        ax[2] = -ax[2]
    return ax * (angle / float(np.linalg.norm(ax)))


@dataclass
class CameraData:
    cam_id: int
    size: tuple[int, int]
    rotation_count: int = 0
    error: float | None = None
    matrix: np.ndarray | None = None
    distortions: np.ndarray | None = None
    exposure: int | None = None
    grid_count: int | None = None
    ignore: bool = False
    translation: np.ndarray | None = None
    rotation: np.ndarray | None = None
    fisheye: bool = False

    def __deepcopy__(self, memo) -> "CameraData":
        """Field-wise copy (arrays copied, scalars and the size tuple shared): what ``copy.deepcopy`` produces, without its per-object
        bookkeeping — ``CaptureVolume.optimize`` copies the whole array on every call (64 cameras: 5 ms through the generic path)."""
        out = object.__new__(type(self))
        for name, value in self.__dict__.items():
            if isinstance(value, np.ndarray):
                out.__dict__[name] = value.copy()
            elif value is None or isinstance(value, (int, float, bool, str)):
                out.__dict__[name] = value
            else:  # (the size tuple, anything a caller hung on the object)
                out.__dict__[name] = _generic_deepcopy(value, memo)
        memo[id(self)] = out
        return out

    @classmethod
    def from_intrinsics(cls, cam_id: int, size: tuple[int, int], focal_length: float | None = None, *, fx: float | None = None,
                        fy: float | None = None, cx: float | None = None, cy: float | None = None, distortions=None) -> "CameraData":
        """Pinhole camera from scalar intrinsics (reference camera_array.py:51-91): ``focal_length`` for square pixels or both
        ``fx`` and ``fy``; the principal point defaults to the image centre; no pose."""
        if focal_length is not None:
            if fx is not None or fy is not None:
                raise ValueError("Pass either focal_length or fx/fy, not both.")
            fx = fy = focal_length
        elif fx is None or fy is None:
            raise ValueError("Pass either focal_length or both fx and fy.")
        w, h = size
        K = np.array([[fx, 0.0, w / 2.0 if cx is None else cx], [0.0, fy, h / 2.0 if cy is None else cy], [0.0, 0.0, 1.0]])
        return cls(cam_id=cam_id, size=size, matrix=K, distortions=np.zeros(5) if distortions is None else np.asarray(distortions, dtype=np.float64))

    @property
    def transformation(self) -> np.ndarray:
        """4 x 4 world -> camera transform ``[[R, t], [0, 1]]`` (reference :93-108)."""
        if self.rotation is None or self.translation is None:
            raise ValueError(f"Camera {self.cam_id} has no pose")
        T = np.eye(4)
        T[:3, :3] = np.asarray(self.rotation, dtype=np.float64)
        T[:3, 3] = np.asarray(self.translation, dtype=np.float64).ravel()
        return T

    @transformation.setter
    def transformation(self, T) -> None:
        T = np.asarray(T, dtype=np.float64)
        self.rotation = T[0:3, 0:3].copy()
        self.translation = T[0:3, 3].copy()

    def erase_calibration_data(self) -> None:
        """Forget intrinsics, pose and calibration statistics (reference :211-217)."""
        self.error = self.matrix = self.distortions = self.grid_count = self.translation = self.rotation = None

    def synthesize_default_intrinsics(self) -> None:
        """Blind intrinsics from the resolution: f = width / 2, principal point at the centre, no distortion (reference
        :219-236; refused for fisheye cameras and without a size)."""
        from caliscope_amd.exceptions import CalibrationError

        if self.size is None:
            raise CalibrationError(f"Camera {self.cam_id} has no resolution data. Load video metadata before synthesizing intrinsics.")
        if self.fisheye:
            raise CalibrationError(f"Camera {self.cam_id} is fisheye; blind intrinsics are not supported for the equidistant model. "
                                   f"Run intrinsic calibration for this camera.")
        w, h = self.size
        self.matrix = np.array([[w / 2.0, 0.0, w / 2.0], [0.0, w / 2.0, h / 2.0], [0.0, 0.0, 1.0]])
        self.distortions = np.zeros(5)

    def extrinsics_to_vector(self) -> np.ndarray:
        """``[rvec(3), tvec(3)]`` — the first six parameters of this camera's block."""
        if self.rotation is None or self.translation is None:
            raise ValueError(f"Camera {self.cam_id} has no pose")
        return np.concatenate([matrix_to_rvec(self.rotation), np.asarray(self.translation, dtype=np.float64).ravel()])

    def extrinsics_from_vector(self, row) -> None:
        row = np.asarray(row, dtype=np.float64)
        self.rotation = rvec_to_matrix(row[0:3])
        self.translation = row[3:6].copy()

    @property
    def position(self) -> np.ndarray:
        """Camera centre in world coordinates, ``-R^T t``."""
        return -np.asarray(self.rotation).T @ np.asarray(self.translation).ravel()

    @property
    def normalized_projection_matrix(self) -> np.ndarray:
        """``[R | t]`` (3 x 4): projection to the normalised image plane (reference camera_array.py, same name)."""
        if self.rotation is None or self.translation is None:
            raise ValueError(f"Camera {self.cam_id} has no pose")
        return np.hstack([np.asarray(self.rotation, dtype=np.float64), np.asarray(self.translation, dtype=np.float64).reshape(3, 1)])

    def undistort_points(self, points, *, output="normalized"):
        """Remove lens distortion (reference camera_array.py:135-174); runs on the device (caliscope_amd.triangulation)."""
        from caliscope_amd.triangulation import undistort_points

        return undistort_points(self, points, output=output)


@dataclass
class CameraArray:
    cameras: Dict[int, CameraData] = field(default_factory=dict)

    @property
    def posed_cameras(self) -> Dict[int, CameraData]:
        return {c: cam for c, cam in self.cameras.items() if cam.rotation is not None and cam.translation is not None}

    @property
    def unposed_cameras(self) -> Dict[int, CameraData]:
        return {c: cam for c, cam in self.cameras.items() if cam.rotation is None or cam.translation is None}

    @property
    def posed_cam_id_to_index(self) -> Dict[int, int]:
        """cam_id -> optimisation index: posed AND not ignored, ascending cam_id."""
        ids = sorted(c for c, cam in self.posed_cameras.items() if not cam.ignore)
        return {c: i for i, c in enumerate(ids)}

    @property
    def posed_index_to_cam_id(self) -> Dict[int, int]:
        return {i: c for c, i in self.posed_cam_id_to_index.items()}

    @property
    def normalized_projection_matrices(self) -> Dict[int, np.ndarray]:
        """cam_id -> ``[R | t]`` for the posed, non-ignored cameras (reference camera_array.py:368-375)."""
        return {c: self.cameras[c].normalized_projection_matrix for c in self.posed_cam_id_to_index}

    @classmethod
    def from_image_sizes(cls, sizes: Dict[int, tuple]) -> "CameraArray":
        """Uncalibrated array from ``{cam_id: (width, height)}`` (reference :335-344)."""
        return cls({int(c): CameraData(cam_id=int(c), size=(int(s[0]), int(s[1]))) for c, s in sizes.items()})

    def all_extrinsics_calibrated(self) -> bool:
        """Every camera has a pose (an empty array counts as calibrated, as in the reference :346-350)."""
        return not self.cameras or not self.unposed_cameras

    def all_intrinsics_calibrated(self) -> bool:
        """At least one camera and every camera has a matrix and distortion coefficients (reference :352-361)."""
        return bool(self.cameras) and all(c.matrix is not None and c.distortions is not None for c in self.cameras.values())

    def all_cameras_have_resolution(self) -> bool:
        active = [c for c in self.cameras.values() if not c.ignore]
        return len(active) > 0 and all(c.size is not None for c in active)

    def __getitem__(self, cam_id: int) -> CameraData:
        return self.cameras[cam_id]

    def __setitem__(self, cam_id: int, camera: CameraData) -> None:
        self.cameras[cam_id] = camera

    def to_toml(self, path: Path | str) -> None:
        """Save to ``camera_array.toml`` (reference camera_array.py:443-487): rotations as Rodrigues vectors, ``None``
        fields omitted, atomic write."""

        path = Path(path)
        try:
            path.parent.mkdir(parents=True, exist_ok=True)
            cams = {}
            for cam_id, cam in self.cameras.items():
                entry = {
                    "cam_id": cam.cam_id, "size": list(cam.size), "rotation_count": cam.rotation_count, "error": cam.error,
                    "matrix": None if cam.matrix is None else np.asarray(cam.matrix, dtype=np.float64).tolist(),
                    "distortions": None if cam.distortions is None else np.asarray(cam.distortions, dtype=np.float64).ravel().tolist(),
                    "translation": None if cam.translation is None else np.asarray(cam.translation, dtype=np.float64).ravel().tolist(),
                    "rotation": None if cam.rotation is None else matrix_to_rvec(cam.rotation).tolist(),
                    "exposure": cam.exposure, "grid_count": cam.grid_count, "fisheye": bool(cam.fisheye),
                }
                cams[str(cam_id)] = {k: v for k, v in entry.items() if v is not None}
            safe_write_toml({"cameras": cams}, path)
        except Exception as exc:
            raise PersistenceError(f"Failed to save CameraArray to {path}: {exc}") from exc

    def to_aniposelib_toml(self, path: Path | str) -> None:
        """aniposelib-compatible export of the posed cameras (reference camera_array.py:489-534)."""

        path = Path(path)
        try:
            path.parent.mkdir(parents=True, exist_ok=True)
            data: dict = {}
            for cam_id, cam in self.posed_cameras.items():
                data[f"cam_{cam_id}"] = {k: v for k, v in {
                    "name": f"cam_{cam_id}", "size": [int(cam.size[0]), int(cam.size[1])],
                    "matrix": None if cam.matrix is None else np.asarray(cam.matrix, dtype=np.float64).tolist(),
                    "distortions": None if cam.distortions is None else np.asarray(cam.distortions, dtype=np.float64).ravel().tolist(),
                    "rotation": matrix_to_rvec(cam.rotation).tolist(),
                    "translation": np.asarray(cam.translation, dtype=np.float64).ravel().tolist(),
                    "fisheye": bool(cam.fisheye),
                }.items() if v is not None}
            data["metadata"] = {"adjusted": False, "error": 0.0}
            safe_write_toml(data, path)
        except Exception as exc:
            raise PersistenceError(f"Failed to save aniposelib CameraArray to {path}: {exc}") from exc

    # -- persistence (reference camera_array.py:377-534) ------------------------------------------
    @classmethod
    def from_toml(cls, path: Path | str) -> "CameraArray":
        import tomli

        path = Path(path)
        if not path.exists():
            raise PersistenceError(f"CameraArray file not found: {path}")
        try:
            with open(path, "rb") as fh:
                doc = tomli.load(fh)
        except Exception as exc:
            raise PersistenceError(f"Failed to load CameraArray from {path}: {exc}") from exc
        if not doc or "cameras" not in doc:
            return cls({})

        def scalar(v):  # legacy files hold the string "null" where a value is missing (reference toml_helpers._clean_scalar)
            return None if v is None or v == "null" else v

        out: Dict[int, CameraData] = {}
        for key, entry in (doc.get("cameras") or {}).items():
            try:
                def arr(name):
                    v = entry.get(name)
                    return None if v is None or v == "null" else np.asarray(v, dtype=np.float64)

                rot = arr("rotation")
                if rot is not None:
                    if rot.shape == (3, 3):
                        pass  # legacy files store the matrix
                    elif rot.size == 3:
                        rot = rvec_to_matrix(rot.ravel())
                    else:
                        raise ValueError(f"invalid rotation shape {rot.shape}")
                size = entry["size"]
                out[int(key)] = CameraData(
                    cam_id=int(key),
                    size=(int(size[0]), int(size[1])),
                    rotation_count=int(entry.get("rotation_count", 0)),
                    error=scalar(entry.get("error")),
                    matrix=arr("matrix"),
                    distortions=arr("distortions"),
                    exposure=scalar(entry.get("exposure")),
                    grid_count=scalar(entry.get("grid_count")),
                    ignore=bool(entry.get("ignore", False)),
                    translation=arr("translation"),
                    rotation=rot,
                    fisheye=bool(entry.get("fisheye", False)),
                )
            except Exception as exc:
                raise PersistenceError(f"Failed to parse camera {key}: {exc}") from exc
        return cls(out)
