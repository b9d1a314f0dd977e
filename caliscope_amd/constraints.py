"""Rigid-geometry constraints of the calibration target, the input of the constraint rows of the BA.

Same public surface as the reference's ``core/constraints.py`` (records ``:20-66``, ``ConstraintSet`` ``:68-518``,
``ConstraintViolation`` / ``RigidityReport`` ``:521-565``) so a ``constraints.toml`` written by either side loads in
the other and ``CaptureVolume(constraints=...)`` means the same thing:

* a ``DistanceConstraint`` pins the distance of two keypoints (object_id, keypoint_id) seen at the same sync index;
* a ``CentroidDistanceConstraint`` pins the distance of the corner centroids (keypoints 0..3) of two markers;
* ``static_object_ids`` name objects that do not move: their keypoints are ONE world point for the whole recording
  (stored at ``STATIC_SYNC_INDEX``), constraints among them fire once;
* ``PointRemap`` folds the observations of a marker printed on the back of a thin board onto the front marker.

The compilers take the target descriptions by duck type (OpenCV is not a dependency of this package):
``from_grid`` is the geometry core; ``from_charuco`` / ``from_chessboard`` / ``from_marker_set`` read the same
attributes the reference's ``Charuco`` / ``Chessboard`` / ``ArucoMarkerSet`` expose.

What the rows do inside the solver is in ``csrc/cba_kernels.h`` ("Rigid-distance constraint rows").
"""

from __future__ import annotations

from dataclasses import dataclass
from functools import cached_property
from pathlib import Path
from typing import Literal

import numpy as np


@dataclass(frozen=True)
class DistanceConstraint:
    object_id_a: int
    keypoint_id_a: int
    object_id_b: int
    keypoint_id_b: int
    distance: float  # metres
    sigma: float  # metres, 1-sigma of the measured distance


@dataclass(frozen=True)
class CentroidDistanceConstraint:
    """Distance between the corner centroids (mean of keypoints 0..3) of two markers."""

    object_id_a: int
    object_id_b: int
    distance: float
    sigma: float


@dataclass(frozen=True)
class PointRemap:
    """Observation identity rewrite: (object_id_from, keypoint_id_from) becomes (object_id_to, keypoint_id_to) and
    takes that keypoint's board coordinates."""

    object_id_from: int
    keypoint_id_from: int
    object_id_to: int
    keypoint_id_to: int
    obj_loc_x: float
    obj_loc_y: float
    obj_loc_z: float


def _grid_edges(corners: np.ndarray, spacing: float) -> np.ndarray:
    """Keypoint pairs that make a planar grid of corners rigid: neighbour edges along both axes, both diagonals of
    every complete cell, and the six distances among the four extreme corners (which cross every fold line the
    local truss leaves free).  The grid position of a corner is recovered from its coordinates (rounded to the
    pitch), not from its id.  Returns an (n_edges, 2) int array."""
    corners = np.asarray(corners, dtype=np.float64)
    ij = np.rint(corners[:, :2] / spacing).astype(np.int64)
    ij -= ij.min(axis=0)
    nx, ny = ij.max(axis=0) + 1
    occ = np.full((nx + 1, ny + 1), -1, dtype=np.int64)  # one spare row/column: neighbour reads never wrap
    occ[ij[:, 0], ij[:, 1]] = np.arange(len(corners))
    here = occ[:-1, :-1]
    right, up, diag = occ[1:, :-1], occ[:-1, 1:], occ[1:, 1:]
    pairs = []
    # along x: a missing corner breaks the chain, the next present one in the row is the neighbour (sorted chain)
    for axis in (0, 1):
        lines = occ[:nx, :ny] if axis == 0 else occ[:nx, :ny].T
        for k in range(lines.shape[1]):
            ids = lines[:, k][lines[:, k] >= 0]
            if ids.size > 1:
                pairs.append(np.stack([ids[:-1], ids[1:]], axis=1))
    cell = (here >= 0) & (right >= 0) & (up >= 0) & (diag >= 0)
    if cell.any():
        pairs.append(np.stack([here[cell], diag[cell]], axis=1))
        pairs.append(np.stack([right[cell], up[cell]], axis=1))
    ext = [occ[0, 0], occ[0, ny - 1], occ[nx - 1, 0], occ[nx - 1, ny - 1]]
    if min(ext) < 0:
        raise KeyError("the grid has no corner at one of its four extreme positions")
    pairs.append(np.array([(ext[a], ext[b]) for a in range(4) for b in range(a + 1, 4)], dtype=np.int64))
    return np.concatenate(pairs, axis=0)


@dataclass(frozen=True)
class ConstraintSet:
    distances: tuple[DistanceConstraint, ...]
    static_object_ids: frozenset[int]
    centroid_distances: tuple[CentroidDistanceConstraint, ...] = ()
    point_remaps: tuple[PointRemap, ...] = ()
    # charuco only: substrate thickness in metres (0.0 = thin board); not None declares that the extraction holds
    # exactly object ids {0} (thin) or {0, 1} (two-sided)
    back_face_thickness_m: float | None = None

    # ---- compilers -----------------------------------------------------------------------------------------
    @staticmethod
    def _truss_distance_constraints(corners, spacing: float, sigma_m: float, object_id: int = 0) -> tuple[DistanceConstraint, ...]:
        """Distance rows of one grid face (reference ``constraints.py:217-308``)."""
        given = np.asarray(corners)
        edges = _grid_edges(np.asarray(given, dtype=np.float64), spacing)
        # lengths in the corners' OWN precision, pair by pair, as the reference forms them (:299): OpenCV boards and Chessboard.get_object_points hand
        # over float32 corners, and a float64 norm of the same numbers differs in the eighth digit (a nanometre — but the compiled set is compared, and
        # persisted, as numbers: tests/golden/reference_host/compilers_*.npz)
        return tuple(DistanceConstraint(object_id, int(a), object_id, int(b), float(np.linalg.norm(given[a] - given[b])), sigma_m) for a, b in edges)

    @staticmethod
    def _cross_face_constraints(corners, spacing: float, thickness_m: float, sigma_m: float) -> tuple[DistanceConstraint, ...]:
        """Rows tying the back face (object 1) of a thick board to the front face (object 0): per corner a tie of
        length t to the same corner and braces of length hypot(pitch, t) to the back-face +x and +y neighbours
        (reference ``constraints.py:310-357``; the braces remove the lateral shear the ties alone allow)."""
        corners = np.asarray(corners, dtype=np.float64)
        ij = np.rint(corners[:, :2] / spacing).astype(np.int64)
        where = {(int(i), int(j)): k for k, (i, j) in enumerate(ij)}
        brace = float(np.hypot(spacing, thickness_m))
        rows = []
        for k, (i, j) in enumerate(ij):
            rows.append(DistanceConstraint(0, k, 1, k, thickness_m, sigma_m))
            for nb in (where.get((int(i) + 1, int(j))), where.get((int(i), int(j) + 1))):
                if nb is not None:
                    rows.append(DistanceConstraint(0, k, 1, nb, brace, sigma_m))
        return tuple(rows)

    @classmethod
    def from_grid(cls, corners, spacing: float, sigma_m: float = 0.002, *, thickness_m: float | None = None,
                  thickness_sigma_m: float = 0.0005) -> "ConstraintSet":
        """Constraints of a planar corner grid (N x 3, metres).  ``thickness_m`` > 0 adds the back face (object 1)
        and the cross-face rows; ``thickness_m`` not None records the closed identity universe of a charuco."""
        rows = cls._truss_distance_constraints(corners, spacing, sigma_m)
        if thickness_m is not None and thickness_m > 0:
            rows = rows + cls._truss_distance_constraints(corners, spacing, sigma_m, object_id=1)
            rows = rows + cls._cross_face_constraints(corners, spacing, thickness_m, thickness_sigma_m)
        return cls(distances=rows, static_object_ids=frozenset(), back_face_thickness_m=thickness_m)

    @classmethod
    def from_charuco(cls, charuco, sigma_m: float = 0.002, thickness_sigma_m: float = 0.0005) -> "ConstraintSet":
        """Reference ``constraints.py:359-395``: ``charuco.board.getChessboardCorners()`` /
        ``.getSquareLength()`` / ``charuco.thickness_m``."""
        corners = np.asarray(charuco.board.getChessboardCorners())
        return cls.from_grid(corners, float(charuco.board.getSquareLength()), sigma_m,
                             thickness_m=float(charuco.thickness_m), thickness_sigma_m=thickness_sigma_m)

    @classmethod
    def from_chessboard(cls, chessboard, sigma_m: float = 0.002) -> "ConstraintSet":
        """Reference ``constraints.py:397-418``; refuses a board without a metric square size."""
        if chessboard.square_size_cm is None:
            raise ValueError("from_chessboard requires square_size_cm to be set; a unit-spacing constraint set would "
                             "silently pin the wrong scale.")
        return cls.from_grid(chessboard.get_object_points(), chessboard.square_size_cm / 100, sigma_m)

    @classmethod
    def from_marker_set(cls, marker_set, sigma_m: float = 0.002, center_sigma_m: float = 0.005) -> "ConstraintSet":
        """Reference ``constraints.py:84-190``.  ``marker_set.markers`` {id: marker with ``.corners`` (4 x 3) and
        ``.static``}, ``.links`` (``marker_a``, ``marker_b``, ``is_center``, ``corner_a``, ``corner_b``, ``distance_m``,
        ``sigma_m``), ``.mirror_pairs`` (``marker_a``, ``marker_b``, ``is_zero_thickness``, ``thickness_m``, ``sigma_m``,
        ``corner_mapping``)."""
        folded = {pair.marker_b for pair in marker_set.mirror_pairs if pair.is_zero_thickness}
        rows: list[DistanceConstraint] = []
        for mid, marker in marker_set.markers.items():
            if mid in folded:
                continue  # its observations carry the front marker's identity
            c = np.asarray(marker.corners, dtype=np.float64)
            rows += [DistanceConstraint(mid, i, mid, j, float(np.linalg.norm(c[i] - c[j])), sigma_m)
                     for i in range(4) for j in range(i + 1, 4)]
        centroids: list[CentroidDistanceConstraint] = []
        for link in marker_set.links:
            if link.is_center:
                centroids.append(CentroidDistanceConstraint(link.marker_a, link.marker_b, link.distance_m,
                                                            center_sigma_m if link.sigma_m is None else link.sigma_m))
            else:
                rows.append(DistanceConstraint(link.marker_a, link.corner_a, link.marker_b, link.corner_b, link.distance_m,
                                               sigma_m if link.sigma_m is None else link.sigma_m))
        remaps: list[PointRemap] = []
        for pair in marker_set.mirror_pairs:
            for ca, cb in pair.corner_mapping:
                if pair.is_zero_thickness:
                    loc = np.asarray(marker_set.markers[pair.marker_a].corners[ca], dtype=np.float64)
                    remaps.append(PointRemap(pair.marker_b, cb, pair.marker_a, ca, float(loc[0]), float(loc[1]), float(loc[2])))
                else:
                    rows.append(DistanceConstraint(pair.marker_a, ca, pair.marker_b, cb, pair.thickness_m,
                                                   sigma_m if pair.sigma_m is None else pair.sigma_m))
        static = frozenset(mid for mid, m in marker_set.markers.items() if m.static and mid not in folded)
        return cls(distances=tuple(rows), static_object_ids=static, centroid_distances=tuple(centroids), point_remaps=tuple(remaps))

    # ---- observation rewrite ---------------------------------------------------------------------------------
    def remap_image_points(self, image_points):
        """Apply the zero-thickness remaps (reference ``constraints.py:192-215``); the input is returned as is when
        there are none."""
        if not self.point_remaps:
            return image_points
        from caliscope_amd.point_data import ImagePoints

        df = image_points.df
        obj, kp = df["object_id"].to_numpy().copy(), df["keypoint_id"].to_numpy().copy()
        loc = {c: df[c].to_numpy(dtype=np.float64).copy() for c in ("obj_loc_x", "obj_loc_y", "obj_loc_z") if c in df.columns}
        for r in self.point_remaps:
            # one after the other on the identities AS THEY ARE BY NOW, as the reference's loop over the same frame does: an observation whose new
            # identity is the source of a later remap moves again (tests/golden/reference_host/remap_*.npz)
            hit = (obj == r.object_id_from) & (kp == r.keypoint_id_from)
            obj[hit], kp[hit] = r.object_id_to, r.keypoint_id_to
            for c, v in zip(("obj_loc_x", "obj_loc_y", "obj_loc_z"), (r.obj_loc_x, r.obj_loc_y, r.obj_loc_z)):
                if c in loc:
                    loc[c][hit] = v
        df["object_id"], df["keypoint_id"] = obj, kp
        for c, v in loc.items():
            df[c] = v
        return ImagePoints(df)

    # ---- persistence (reference constraints.py:420-518) ---------------------------------------------------------
    def to_toml(self, path) -> None:
        from dataclasses import asdict

        from caliscope_amd.persistence import safe_write_toml

        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        doc: dict = {"static_object_ids": sorted(self.static_object_ids), "distances": [asdict(d) for d in self.distances]}
        if self.centroid_distances:
            doc["centroid_distances"] = [asdict(c) for c in self.centroid_distances]
        if self.point_remaps:
            doc["point_remaps"] = [asdict(r) for r in self.point_remaps]
        if self.back_face_thickness_m is not None:
            doc["back_face_thickness_m"] = self.back_face_thickness_m
        safe_write_toml(doc, path)

    @classmethod
    def from_toml(cls, path) -> "ConstraintSet":
        import tomli

        from caliscope_amd.persistence import PersistenceError

        path = Path(path)
        if not path.exists():
            raise PersistenceError(f"ConstraintSet file not found: {path}")
        try:
            with open(path, "rb") as fh:
                doc = tomli.load(fh)
            return cls(
                distances=tuple(DistanceConstraint(**d) for d in doc.get("distances", [])),
                static_object_ids=frozenset(doc.get("static_object_ids", [])),
                centroid_distances=tuple(CentroidDistanceConstraint(**c) for c in doc.get("centroid_distances", [])),
                point_remaps=tuple(PointRemap(**r) for r in doc.get("point_remaps", [])),
                back_face_thickness_m=doc.get("back_face_thickness_m"),
            )
        except Exception as exc:
            raise PersistenceError(f"Failed to load ConstraintSet from {path}: {exc}") from exc


@dataclass(frozen=True)
class ConstraintViolation:
    object_id_a: int
    keypoint_id_a: int  # -1 for a centroid endpoint
    object_id_b: int
    keypoint_id_b: int
    sync_index: int
    expected: float
    actual: float
    kind: Literal["corner", "centroid"] = "corner"


@dataclass(frozen=True)
class RigidityReport:
    """Measured vs. target distances of every constraint instance (reference ``constraints.py:535-565``)."""

    violations: tuple[ConstraintViolation, ...]

    @cached_property
    def _err(self) -> np.ndarray:
        return np.array([v.actual - v.expected for v in self.violations], dtype=np.float64)

    @cached_property
    def _rel(self) -> np.ndarray:
        return np.array([(v.actual - v.expected) / v.expected if v.expected != 0 else 0.0 for v in self.violations], dtype=np.float64)

    @cached_property
    def rmse_mm(self) -> float:
        return float(np.sqrt(np.mean(self._err**2)) * 1000.0) if self.violations else 0.0

    @cached_property
    def relative_rmse_pct(self) -> float:
        if not self.violations:
            return 0.0
        rel = np.array([(v.actual - v.expected) / v.expected for v in self.violations])
        return float(np.sqrt(np.mean(rel**2)) * 100.0)

    @cached_property
    def max_violation_mm(self) -> float:
        return float(np.abs(self._err).max() * 1000.0) if self.violations else 0.0

    def _per_object(self, values: np.ndarray, factor: float) -> dict[int, float]:
        acc: dict[int, list[float]] = {}
        for v, e in zip(self.violations, values):
            for oid in {v.object_id_a, v.object_id_b}:
                acc.setdefault(oid, []).append(float(e))
        return {oid: float(np.sqrt(np.mean(np.square(es))) * factor) for oid, es in acc.items()}

    @cached_property
    def per_object_rmse_mm(self) -> dict[int, float]:
        return self._per_object(self._err, 1000.0)

    @cached_property
    def per_object_relative_rmse_pct(self) -> dict[int, float]:
        return self._per_object(self._rel, 100.0)
