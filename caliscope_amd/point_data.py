"""2-D / 3-D point tables on either side of the BA path.

Mirrors the containers the reference's ``CaptureVolume.optimize`` reads
(``core/point_data.py:256-276`` column schemas, ``:323-373`` ImagePoints, WorldPoints):
validated, copy-on-read pandas DataFrames with the same column names, plus the same CSV
round-trip (``from_csv`` / ``to_csv``).  ``ImagePoints.triangulate`` (``:416-559``) is provided through
``caliscope_amd.triangulation`` (device kernel); gap filling and smoothing are upstream of the path and out of scope.
"""

from __future__ import annotations

from pathlib import Path

import numpy as np
import pandas as pd

STATIC_SYNC_INDEX = -1  # reference core/point_data.py (static objects share one world point)

IMAGE_POINT_COLUMNS = {
    "sync_index": "int",
    "cam_id": "int",
    "object_id": "int",
    "keypoint_id": "int",
    "img_loc_x": "float",
    "img_loc_y": "float",
}
IMAGE_POINT_OPTIONAL = ("obj_loc_x", "obj_loc_y", "obj_loc_z", "frame_time")
WORLD_POINT_COLUMNS = {
    "sync_index": "int",
    "object_id": "int",
    "keypoint_id": "int",
    "x_coord": "float",
    "y_coord": "float",
    "z_coord": "float",
}
WORLD_POINT_OPTIONAL = ("frame_time",)


def _validated(df: pd.DataFrame, required: dict, optional: tuple, what: str) -> pd.DataFrame:
    df = df.copy()
    missing = [c for c in required if c not in df.columns]
    if missing:
        raise ValueError(f"{what} validation failed: column(s) {missing} not in dataframe. Columns found: {list(df.columns)}")
    for col in optional:
        if col not in df.columns:
            df[col] = np.nan
    for col, kind in required.items():
        num = pd.to_numeric(df[col], errors="coerce")
        if num.isna().any():
            raise ValueError(f"{what} validation failed: non-nullable column '{col}' contains {int(num.isna().sum())} null value(s)")
        df[col] = num.astype("int64") if kind == "int" else num.astype("float64")
    return df


class ImagePoints:
    """Validated, immutable table of 2-D observations (one row per camera x keypoint x frame)."""

    def __init__(self, df: pd.DataFrame):
        self._df = _validated(df, IMAGE_POINT_COLUMNS, IMAGE_POINT_OPTIONAL, "ImagePoints")

    @property
    def df(self) -> pd.DataFrame:
        return self._df.copy()

    def __len__(self) -> int:
        return len(self._df)

    @classmethod
    def from_csv(cls, path: str | Path) -> "ImagePoints":
        return cls(pd.read_csv(path))

    def to_csv(self, path: str | Path) -> None:
        from caliscope_amd.persistence import CSV_FLOAT_PRECISION, safe_write_csv

        Path(path).parent.mkdir(parents=True, exist_ok=True)
        safe_write_csv(self._df, Path(path), index=False, float_format=CSV_FLOAT_PRECISION)

    def triangulate(self, camera_array, static_object_ids=frozenset()) -> "WorldPoints":
        """Undistort + DLT-triangulate every (sync_index, object_id, keypoint_id) seen by two or more posed cameras
        (reference ``core/point_data.py:416-559``), on the MI355X."""
        from caliscope_amd.triangulation import triangulate

        return triangulate(self, camera_array, static_object_ids)


class WorldPoints:
    """Validated, immutable table of 3-D points keyed by (sync_index, object_id, keypoint_id)."""

    def __init__(self, df: pd.DataFrame):
        self._df = _validated(df, WORLD_POINT_COLUMNS, WORLD_POINT_OPTIONAL, "WorldPoints")

    @property
    def df(self) -> pd.DataFrame:
        return self._df.copy()

    @property
    def points(self) -> np.ndarray:
        return self._df[["x_coord", "y_coord", "z_coord"]].to_numpy(dtype=np.float64)

    def __len__(self) -> int:
        return len(self._df)

    @classmethod
    def from_csv(cls, path: str | Path) -> "WorldPoints":
        return cls(pd.read_csv(path))

    def to_csv(self, path: str | Path) -> None:
        from caliscope_amd.persistence import CSV_FLOAT_PRECISION, safe_write_csv

        Path(path).parent.mkdir(parents=True, exist_ok=True)
        safe_write_csv(self._df, Path(path), index=False, float_format=CSV_FLOAT_PRECISION)
