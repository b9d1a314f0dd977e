"""2-D / 3-D point tables on either side of the BA path.

Mirrors the containers the reference's ``CaptureVolume.optimize`` reads
(``core/point_data.py:256-276`` column schemas, ``:323-373`` ImagePoints, WorldPoints):
validated, copy-on-read pandas DataFrames with the same column names, plus the same CSV
round-trip (``from_csv`` / ``to_csv``).  ``ImagePoints.triangulate`` (``:416-559``) is provided through
``caliscope_amd.triangulation`` (device kernel).  ``fill_gaps`` / ``filter_to_objects`` / ``smooth`` (``:375-414``, ``:606-651``) are
host-side table utilities around the path, vectorised here (the reference loops over every track with a merge).
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


def _fill_track_gaps(df: pd.DataFrame, track_keys: list, value_cols: list, max_gap_size: int) -> pd.DataFrame:
    """Rows for the missing sync indices of every track (``track_keys``), as the reference fills them: inside a hole of g
    frames the first k = min(g, max_gap_size) get a row, with values on a straight line through the k + 1 equal steps
    between the valid frames on either side (pandas ``interpolate(method="linear")`` after the rows beyond the limit were
    dropped: evenly spaced in ROWS, not in sync index); all other columns of the new rows are NaN."""
    if len(df) == 0 or max_gap_size <= 0:
        return df
    df = df.sort_values(track_keys + ["sync_index"], kind="stable").reset_index(drop=True)
    keys = df[track_keys].to_numpy()
    sync = df["sync_index"].to_numpy()
    same = np.all(keys[1:] == keys[:-1], axis=1)
    gap = np.where(same, sync[1:] - sync[:-1] - 1, 0)
    left = np.flatnonzero(gap > 0)                  # row before a hole
    if left.size == 0:
        return df
    k = np.minimum(gap[left], max_gap_size)           # rows to insert per hole
    src = np.repeat(left, k)
    j = np.arange(k.sum()) - np.repeat(np.cumsum(k) - k, k) + 1   # 1..k inside each hole
    new = pd.DataFrame({c: np.repeat(df[c].to_numpy()[left], k) for c in track_keys})
    new["sync_index"] = sync[src] + j
    frac = j / (np.repeat(k, k) + 1.0)
    for c in value_cols:
        if c in df.columns:
            v = df[c].to_numpy(dtype=np.float64)
            new[c] = v[src] + (v[src + 1] - v[src]) * frac
    out = pd.concat([df, new], ignore_index=True)     # columns absent from `new` become NaN
    return out.sort_values(track_keys + ["sync_index"], kind="stable").reset_index(drop=True)


def _take_rows(df: pd.DataFrame, rows) -> pd.DataFrame:
    """Row subset (boolean mask or index array) re-indexed from 0, column by column: numpy selects 2M rows of one column in a few milliseconds, while
    ``df[mask]`` on the consolidated frame takes every block apart and puts it together again (0.12 s of the 0.17 s of a filter pass on 2M observations)."""
    rows = np.asarray(rows)
    return pd.DataFrame({c: df[c].to_numpy()[rows] for c in df.columns}, copy=False)


def _validated(df: pd.DataFrame, required: dict, optional: tuple, what: str) -> pd.DataFrame:
    df = df.copy()
    missing = [c for c in required if c not in df.columns]
    if missing:
        raise ValueError(f"{what} validation failed: column(s) {missing} not in dataframe. Columns found: {list(df.columns)}")
    for col in optional:
        if col not in df.columns:
            df[col] = np.nan
    for col, kind in required.items():
        have = df[col].dtype
        if (kind == "int" and have == np.int64) or (kind == "float" and have == np.float64 and not np.isnan(df[col].to_numpy()).any()):
            continue  # already what validation would make of it (an int64 column has no nulls): no conversion pass over millions of rows
        num = pd.to_numeric(df[col], errors="coerce")
        if num.isna().any():
            raise ValueError(f"{what} validation failed: non-nullable column '{col}' contains {int(num.isna().sum())} null value(s)")
        df[col] = num.astype("int64") if kind == "int" else num.astype("float64")
    return df


class ImagePoints:
    """Validated, immutable table of 2-D observations (one row per camera x keypoint x frame)."""

    def __init__(self, df: pd.DataFrame):
        self._df = _validated(df, IMAGE_POINT_COLUMNS, IMAGE_POINT_OPTIONAL, "ImagePoints")
        self._arrays = None

    def take(self, rows) -> "ImagePoints":
        """The rows selected by a boolean mask or an index array, re-indexed from 0.  A subset of a validated table is
        valid, so nothing is re-checked (the filters between solver passes use this)."""
        out = object.__new__(ImagePoints)
        out._df = _take_rows(self._df, rows)
        out._arrays = None
        return out

    def arrays(self) -> dict:
        """The required columns as numpy arrays (read-only, built once: the table is immutable) — the marshalling of
        ``CaptureVolume.optimize`` and the report work on these instead of going through pandas on every call."""
        if self._arrays is None:
            cols = {c: self._df[c].to_numpy() for c in IMAGE_POINT_COLUMNS}
            for a in cols.values():
                a.setflags(write=False)
            self._arrays = cols
        return self._arrays

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

    def fill_gaps(self, max_gap_size: int = 3) -> "ImagePoints":
        """Fill holes of up to ``max_gap_size`` frames in every (cam_id, object_id, keypoint_id) track by linear interpolation of
        the pixel position and the frame time (reference ``:375-402``; a longer hole gets its first ``max_gap_size`` frames)."""
        return ImagePoints(_fill_track_gaps(self._df, ["cam_id", "object_id", "keypoint_id"], ["img_loc_x", "img_loc_y", "frame_time"], max_gap_size))

    def filter_to_objects(self, object_ids) -> "ImagePoints":
        """Only the rows whose object_id is in ``object_ids`` (reference ``:404-414``)."""
        return ImagePoints(self._df[self._df["object_id"].isin(set(int(o) for o in object_ids))].copy())

    def triangulate(self, camera_array, static_object_ids=frozenset()) -> "WorldPoints":
        """Undistort + DLT-triangulate every (sync_index, object_id, keypoint_id) seen by two or more posed cameras
        (reference ``core/point_data.py:416-559``), on the MI355X."""
        from caliscope_amd.triangulation import triangulate

        return triangulate(self, camera_array, static_object_ids)


class WorldPoints:
    """Validated, immutable table of 3-D points keyed by (sync_index, object_id, keypoint_id)."""

    def __init__(self, df: pd.DataFrame):
        self._df = _validated(df, WORLD_POINT_COLUMNS, WORLD_POINT_OPTIONAL, "WorldPoints")
        # first / last sync index of the moving points; static points sit at STATIC_SYNC_INDEX (reference :586-595)
        sync = self._df["sync_index"].to_numpy()
        moving = sync[sync != STATIC_SYNC_INDEX]
        self.min_index = int(moving.min()) if moving.size else 0
        self.max_index = int(moving.max()) if moving.size else 0

    @property
    def df(self) -> pd.DataFrame:
        return self._df.copy()

    @property
    def points(self) -> np.ndarray:
        xyz = getattr(self, "_xyz", None)
        if xyz is None:
            xyz = self._xyz = self._df[["x_coord", "y_coord", "z_coord"]].to_numpy(dtype=np.float64)
        return xyz.copy()

    def take(self, rows) -> "WorldPoints":
        """Row subset (boolean mask or index array) of a validated table, re-indexed from 0, without re-validation."""
        out = object.__new__(WorldPoints)
        out._df = _take_rows(self._df, rows)
        sync = out._df["sync_index"].to_numpy()
        moving = sync[sync != STATIC_SYNC_INDEX]
        out.min_index = int(moving.min()) if moving.size else 0
        out.max_index = int(moving.max()) if moving.size else 0
        return out

    def with_points(self, xyz) -> "WorldPoints":
        """The same keys with new coordinates — what ``CaptureVolume.optimize`` returns its points in.  The table was
        validated when this object was built, so only the new block is checked (finite ``(P, 3)`` floats) instead of
        re-validating every column, which costs more than a small solve."""
        xyz = np.asarray(xyz, dtype=np.float64)
        if xyz.shape != (len(self._df), 3):
            raise ValueError(f"expected {(len(self._df), 3)} coordinates, got {xyz.shape}")
        if not np.isfinite(xyz).all():
            raise ValueError("WorldPoints validation failed: non-finite coordinates")
        out = object.__new__(WorldPoints)
        own = xyz.copy()  # the new table's coordinates; its key columns are the (immutable) ones of this table, not copies: a fresh frame over
        axis = {"x_coord": 0, "y_coord": 1, "z_coord": 2}  # existing arrays is 0.1 ms at 200 000 points, copy + column assignment 3.4
        cols = {c: (own[:, axis[c]] if c in axis else self._df[c].to_numpy()) for c in self._df.columns}
        out._df = pd.DataFrame(cols, index=self._df.index, copy=False)
        out.min_index, out.max_index, out._xyz = self.min_index, self.max_index, own
        return out

    def __len__(self) -> int:
        return len(self._df)

    def fill_gaps(self, max_gap_size: int = 3) -> "WorldPoints":
        """Fill holes of up to ``max_gap_size`` frames in every (object_id, keypoint_id) trajectory (reference ``:606-634``)."""
        return WorldPoints(_fill_track_gaps(self._df, ["object_id", "keypoint_id"], ["x_coord", "y_coord", "z_coord", "frame_time"], max_gap_size))

    def smooth(self, fps: float, cutoff_freq: float, order: int = 2) -> "WorldPoints":
        """Zero-phase Butterworth low-pass of every trajectory with more than ``3 * order`` samples (reference ``:636-651``;
        samples are taken in table order, as there)."""
        from scipy.signal import butter, filtfilt

        b, a = butter(order, cutoff_freq, btype="low", fs=fps, output="ba")
        df = self._df.copy()
        track = df.groupby(["object_id", "keypoint_id"], sort=False).ngroup().to_numpy()
        order_idx = np.argsort(track, kind="stable")
        bounds = np.flatnonzero(np.r_[True, track[order_idx][1:] != track[order_idx][:-1], True])
        xyz = df[["x_coord", "y_coord", "z_coord"]].to_numpy(dtype=np.float64)
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            if hi - lo > 3 * order:
                rows = order_idx[lo:hi]
                xyz[rows] = filtfilt(b, a, xyz[rows], axis=0)
        df[["x_coord", "y_coord", "z_coord"]] = xyz
        return WorldPoints(df)

    @classmethod
    def from_csv(cls, path: str | Path) -> "WorldPoints":
        return cls(pd.read_csv(path))

    def to_csv(self, path: str | Path) -> None:
        from caliscope_amd.persistence import CSV_FLOAT_PRECISION, safe_write_csv

        Path(path).parent.mkdir(parents=True, exist_ok=True)
        safe_write_csv(self._df, Path(path), index=False, float_format=CSV_FLOAT_PRECISION)
