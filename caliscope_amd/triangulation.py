"""Undistortion + batched DLT triangulation on the MI355X — the step that produces the world-point x0 of the BA path.

Host-side mirror of the reference's ``core/point_data.py``:

* :func:`triangulate_image_points` — same signature and result as ``point_data.py:121-229`` (inputs are undistorted
  normalised coordinates; returns ``(sync_indices, object_ids, keypoint_ids, xyz)`` of the points seen by at least two
  cameras);
* :func:`triangulate` — what ``ImagePoints.triangulate`` (``:416-559``) does: undistort every observation with its
  camera's intrinsics (``CameraData.undistort_points(..., output="normalized")``, ``camera_array.py:135-174``), then
  triangulate per ``(sync_index, object_id, keypoint_id)``; observations of ``static_object_ids`` are pooled over all
  frames under :data:`STATIC_SYNC_INDEX`.  Undistortion and the DLT run in one kernel (``cba_triangulate``).

The reference groups points by camera set to batch ``np.linalg.svd``; on the device every point is one thread, so no
grouping is needed and the rows come back sorted by ``(sync_index, object_id, keypoint_id)`` (callers use the result as a
keyed table).  There is no CPU fallback: without the library or a GPU this raises ``BackendError``.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import pandas as pd

from caliscope_amd import _lib
from caliscope_amd.exceptions import BackendError
from caliscope_amd.point_data import STATIC_SYNC_INDEX, WORLD_POINT_COLUMNS, ImagePoints, WorldPoints


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def _group(sync_indices, camera_ids, object_ids, keypoint_ids):
    """Sort observations by (sync, object, keypoint, camera); returns order, starts of the point groups."""
    order = np.lexsort((camera_ids, keypoint_ids, object_ids, sync_indices))
    s, o, k = sync_indices[order], object_ids[order], keypoint_ids[order]
    brk = np.flatnonzero((np.diff(s) != 0) | (np.diff(o) != 0) | (np.diff(k) != 0)) + 1
    starts = np.concatenate([[0], brk, [len(order)]]).astype(np.int64)
    return order, starts


def _run(cam_P, starts, cam_index, xy, *, cam_model=None, cam_intr=None, float32_io=True, device_id=0, want_undistorted=False):
    lib = _lib.load()
    n_points = len(starts) - 1
    cam_P = np.ascontiguousarray(cam_P, dtype=np.float64).reshape(-1, 12)
    cam_index = np.ascontiguousarray(cam_index, dtype=np.int32)
    xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
    starts = np.ascontiguousarray(starts, dtype=np.int64)
    xyz = np.empty((n_points, 3))
    und = np.empty_like(xy) if want_undistorted else None
    desc = _lib.TriangulateDesc(
        n_cams=cam_P.shape[0],
        cam_model=_ptr(cam_model, C.c_int32) if cam_intr is not None else None,
        cam_intr=_ptr(cam_intr, C.c_double) if cam_intr is not None else None,
        cam_P=_ptr(cam_P, C.c_double), n_points=n_points, pt_start=_ptr(starts, C.c_int64),
        obs_cam=_ptr(cam_index, C.c_int32), obs_xy=_ptr(xy, C.c_double), float32_io=1 if float32_io else 0,
    )
    rc = lib.cba_triangulate(C.byref(desc), device_id, _ptr(xyz, C.c_double), _ptr(und, C.c_double) if und is not None else None)
    if rc != 0:
        raise BackendError(f"cba_triangulate failed ({rc}): {lib.cba_last_error().decode()}")
    return xyz, und


def _empty():
    e = np.array([], dtype=np.int64)
    return e, e.copy(), e.copy(), np.zeros((0, 3))


def triangulate_image_points(projection_matrices, sync_indices, camera_ids, object_ids, keypoint_ids, img_xy, *, device_id=0):
    """Drop-in for ``caliscope.core.point_data.triangulate_image_points`` (undistorted normalised ``img_xy``)."""
    sync_indices, camera_ids = np.asarray(sync_indices, dtype=np.int64), np.asarray(camera_ids, dtype=np.int64)
    object_ids, keypoint_ids = np.asarray(object_ids, dtype=np.int64), np.asarray(keypoint_ids, dtype=np.int64)
    img_xy = np.asarray(img_xy, dtype=np.float64).reshape(-1, 2)
    if len(keypoint_ids) < 2:
        return _empty()
    cam_ids = sorted(projection_matrices)
    index_of = {c: i for i, c in enumerate(cam_ids)}
    cam_P = np.stack([np.asarray(projection_matrices[c], dtype=np.float64).reshape(12) for c in cam_ids])
    order, starts = _group(sync_indices, camera_ids, object_ids, keypoint_ids)
    cam_index = np.array([index_of[int(c)] for c in camera_ids[order]], dtype=np.int32)
    xyz, _ = _run(cam_P, starts, cam_index, img_xy[order], device_id=device_id)
    keep = np.diff(starts) >= 2
    first = order[starts[:-1]]
    if not keep.any():
        return _empty()
    return sync_indices[first][keep], object_ids[first][keep], keypoint_ids[first][keep], xyz[keep]


def camera_tables(camera_array, cam_ids):
    """Flat per-camera tables of ``cba_triangulate``: model, [fx fy cx cy d0..d4], normalised [R | t]."""
    model = np.zeros(len(cam_ids), dtype=np.int32)
    intr = np.zeros((len(cam_ids), 9))
    P = np.zeros((len(cam_ids), 12))
    for i, c in enumerate(cam_ids):
        cam = camera_array.cameras[c]
        if cam.matrix is None or cam.distortions is None:
            raise ValueError(f"Camera {c} lacks intrinsic calibration; cannot undistort points.")
        K = np.asarray(cam.matrix, dtype=np.float64)
        d = np.asarray(cam.distortions, dtype=np.float64).ravel()
        model[i] = 1 if cam.fisheye else 0
        intr[i, :4] = (K[0, 0], K[1, 1], K[0, 2], K[1, 2])
        intr[i, 4 : 4 + min(len(d), 5)] = d[:5]
        P[i] = np.hstack([np.asarray(cam.rotation, dtype=np.float64), np.asarray(cam.translation, dtype=np.float64).reshape(3, 1)]).reshape(12)
    return model, intr, P


def triangulate(image_points: ImagePoints, camera_array, static_object_ids=frozenset(), *, float32_io=True, device_id=0) -> WorldPoints:
    """``ImagePoints.triangulate(camera_array, static_object_ids)`` of the reference, on the device."""
    df = image_points.df
    cols = list(WORLD_POINT_COLUMNS) + ["frame_time"]
    if df.empty:
        return WorldPoints(pd.DataFrame(columns=cols))
    posed = camera_array.posed_cam_id_to_index
    cam_ids = sorted(c for c in df["cam_id"].unique() if c in posed)
    if not cam_ids:
        return WorldPoints(pd.DataFrame(columns=cols))
    data = df[df["cam_id"].isin(cam_ids)]
    frame_times = df.groupby("sync_index")["frame_time"].mean()
    sync = data["sync_index"].to_numpy(dtype=np.int64).copy()
    obj = data["object_id"].to_numpy(dtype=np.int64)
    if static_object_ids:
        sync[np.isin(obj, list(static_object_ids))] = STATIC_SYNC_INDEX
    kp = data["keypoint_id"].to_numpy(dtype=np.int64)
    cam = data["cam_id"].to_numpy(dtype=np.int64)
    xy = np.column_stack([data["img_loc_x"].to_numpy(), data["img_loc_y"].to_numpy()])
    model, intr, P = camera_tables(camera_array, cam_ids)
    index_of = {c: i for i, c in enumerate(cam_ids)}
    order, starts = _group(sync, cam, obj, kp)
    cam_index = np.array([index_of[int(c)] for c in cam[order]], dtype=np.int32)
    xyz, _ = _run(P, starts, cam_index, xy[order], cam_model=model, cam_intr=intr, float32_io=float32_io, device_id=device_id)
    keep = np.diff(starts) >= 2
    first = order[starts[:-1]][keep]
    out_sync = sync[first]
    ft = frame_times.reindex(out_sync).to_numpy()
    ft = np.where(out_sync == STATIC_SYNC_INDEX, np.nan, ft)
    out = pd.DataFrame({
        "sync_index": out_sync, "object_id": obj[first], "keypoint_id": kp[first],
        "x_coord": xyz[keep, 0], "y_coord": xyz[keep, 1], "z_coord": xyz[keep, 2], "frame_time": ft,
    })
    return WorldPoints(out)


def undistort_points(camera, points, *, output="normalized", float32_io=True, device_id=0) -> np.ndarray:
    """``CameraData.undistort_points`` on the device (``output="pixels"`` re-applies the camera matrix, as cv2's P=K)."""
    if camera.matrix is None or camera.distortions is None:
        raise ValueError(f"Camera {camera.cam_id} lacks intrinsic calibration; cannot undistort points.")
    pts = np.asarray(points, dtype=np.float64).reshape(-1, 2)
    if len(pts) == 0:
        return pts.copy()
    K = np.asarray(camera.matrix, dtype=np.float64)
    d = np.asarray(camera.distortions, dtype=np.float64).ravel()
    model = np.array([1 if camera.fisheye else 0], dtype=np.int32)
    intr = np.zeros((1, 9))
    intr[0, :4] = (K[0, 0], K[1, 1], K[0, 2], K[1, 2])
    intr[0, 4 : 4 + min(len(d), 5)] = d[:5]
    P = np.hstack([np.eye(3), np.zeros((3, 1))]).reshape(1, 12)
    # every observation is a one-view "point": its DLT result is NaN and ignored, only the undistortion is used
    n = len(pts)
    _, und = _run(P, np.arange(n + 1, dtype=np.int64), np.zeros(n, dtype=np.int32), pts, cam_model=model, cam_intr=intr,
                  float32_io=float32_io, device_id=device_id, want_undistorted=True)
    if output == "normalized":
        return und
    return np.column_stack([K[0, 0] * und[:, 0] + K[0, 2], K[1, 1] * und[:, 1] + K[1, 2]])
