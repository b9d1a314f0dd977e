"""ConstraintSet compilers, persistence and the CaptureVolume marshalling of constraint rows.

Expectations follow the reference's tests/test_constraints.py (counts, distances, firing rules); the target
descriptions are small duck-typed stand-ins since OpenCV is not a dependency here."""

from dataclasses import dataclass, field
from types import SimpleNamespace

import numpy as np
import pandas as pd
import pytest

from caliscope_amd.cameras import CameraArray, CameraData
from caliscope_amd.capture_volume import CaptureVolume
from caliscope_amd.constraints import (CentroidDistanceConstraint, ConstraintSet, DistanceConstraint, PointRemap)
from caliscope_amd.persistence import PersistenceError
from caliscope_amd.point_data import STATIC_SYNC_INDEX, ImagePoints, WorldPoints


# ---- stand-ins for Charuco / Chessboard / ArucoMarkerSet -----------------------------------------------------
def grid_corners(cols, rows, pitch, shuffle=False):
    c = np.array([[i * pitch, j * pitch, 0.0] for j in range(rows) for i in range(cols)], dtype=np.float32)
    return c[np.random.default_rng(0).permutation(len(c))] if shuffle else c


def charuco(cols=4, rows=6, pitch=0.03, thickness=0.0, shuffle=False):
    corners = grid_corners(cols, rows, pitch, shuffle)
    board = SimpleNamespace(getChessboardCorners=lambda: corners, getSquareLength=lambda: pitch)
    return SimpleNamespace(board=board, thickness_m=thickness)


@dataclass
class Marker:
    marker_id: int
    size_m: float
    static: bool = False

    @property
    def corners(self):
        h = self.size_m / 2
        return np.array([[-h, h, 0], [h, h, 0], [h, -h, 0], [-h, -h, 0]], dtype=np.float64)


@dataclass
class Link:
    marker_a: int
    marker_b: int
    distance_m: float
    corner_a: int | None = None
    corner_b: int | None = None
    sigma_m: float | None = None

    @property
    def is_center(self):
        return self.corner_a is None


@dataclass
class Mirror:
    marker_a: int
    marker_b: int
    thickness_m: float = 0.0
    sigma_m: float | None = None
    corner_mapping: tuple = ((0, 1), (1, 0), (2, 3), (3, 2))

    @property
    def is_zero_thickness(self):
        return self.thickness_m == 0.0


@dataclass
class MarkerSet:
    markers: dict
    links: list = field(default_factory=list)
    mirror_pairs: list = field(default_factory=list)


def one_camera():
    cam = CameraData(cam_id=0, size=(400, 400), matrix=np.array([[200.0, 0, 200], [0, 200, 200], [0, 0, 1]]), distortions=np.zeros(5),
                     rotation=np.eye(3), translation=np.array([0.0, 0.0, 5.0]))
    return CameraArray({0: cam})


# ---- compilers ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shuffle", [False, True])
def test_grid_truss_count_and_lengths(shuffle):
    ch = charuco(shuffle=shuffle)
    cs = ConstraintSet.from_charuco(ch)
    R, C = 6, 4
    assert len(cs.distances) == R * (C - 1) + (R - 1) * C + 2 * (R - 1) * (C - 1) + 6  # reference test :1134-1148
    assert all(d.distance > 0 and d.object_id_a == 0 and d.object_id_b == 0 for d in cs.distances)
    assert cs.static_object_ids == frozenset() and cs.centroid_distances == () and cs.back_face_thickness_m == 0.0
    given = ch.board.getChessboardCorners()  # float32, as OpenCV's boards hand them over
    for d in cs.distances:  # lengths in the corners' own precision, as the reference forms them (constraints.py:299; tests/golden/reference_host/compilers_*)
        assert d.distance == float(np.linalg.norm(given[d.keypoint_id_a] - given[d.keypoint_id_b]))
    corners = given.astype(np.float64)
    lengths = np.array([d.distance for d in cs.distances])
    assert np.isclose(lengths, 0.03, atol=1e-7).sum() == R * (C - 1) + (R - 1) * C
    assert np.isclose(lengths, 0.03 * np.sqrt(2), atol=1e-7).sum() == 2 * (R - 1) * (C - 1)
    xs, ys = corners[:, 0], corners[:, 1]
    ext = {int(np.argmin((xs - x) ** 2 + (ys - y) ** 2)) for x in (xs.min(), xs.max()) for y in (ys.min(), ys.max())}
    braces = {frozenset((d.keypoint_id_a, d.keypoint_id_b)) for d in cs.distances if d.keypoint_id_a in ext and d.keypoint_id_b in ext}
    assert len(ext) == 4 and len(braces) == 6


def test_thick_board_adds_back_face_ties_and_braces():
    thin, thick = ConstraintSet.from_charuco(charuco()), ConstraintSet.from_charuco(charuco(thickness=0.005))
    n_truss, R, C = len(thin.distances), 6, 4
    front = [d for d in thick.distances if d.object_id_a == 0 and d.object_id_b == 0]
    back = [d for d in thick.distances if d.object_id_a == 1 and d.object_id_b == 1]
    cross = [d for d in thick.distances if d.object_id_a == 0 and d.object_id_b == 1]
    assert len(front) == len(back) == n_truss and len(front) + len(back) + len(cross) == len(thick.distances)
    ties = [d for d in cross if d.keypoint_id_a == d.keypoint_id_b]
    braces = [d for d in cross if d.keypoint_id_a != d.keypoint_id_b]
    assert len(ties) == R * C and all(d.distance == 0.005 and d.sigma == 0.0005 for d in ties)
    assert len(braces) == R * (C - 1) + (R - 1) * C
    assert all(d.distance == pytest.approx(np.hypot(0.03, 0.005)) for d in braces)
    assert thick.back_face_thickness_m == 0.005


def test_chessboard_needs_a_metric_square():
    board = SimpleNamespace(square_size_cm=None, get_object_points=lambda: grid_corners(3, 3, 1.0))
    with pytest.raises(ValueError, match="square_size_cm"):
        ConstraintSet.from_chessboard(board)
    board = SimpleNamespace(square_size_cm=2.5, get_object_points=lambda: grid_corners(3, 3, 0.025))
    cs = ConstraintSet.from_chessboard(board)
    assert len(cs.distances) == 3 * 2 + 2 * 3 + 2 * 4 + 6 and cs.back_face_thickness_m is None


def test_marker_set_compiler():
    ms = MarkerSet({i: Marker(i, 0.1) for i in range(8)}, links=[Link(0, 1, 0.5, corner_a=0, corner_b=2)])
    cs = ConstraintSet.from_marker_set(ms)
    assert len(cs.distances) == 8 * 6 + 1 and cs.distances[-1] == DistanceConstraint(0, 0, 1, 2, 0.5, 0.002)
    single = ConstraintSet.from_marker_set(MarkerSet({3: Marker(3, 0.1)}))
    assert sorted(round(d.distance, 12) for d in single.distances) == sorted([0.1] * 4 + [round(0.1 * np.sqrt(2), 12)] * 2)
    # centre link -> centroid row; sigma defaults: corner 0.002, centre 0.005, a link's own sigma wins
    ms = MarkerSet({0: Marker(0, 0.1), 1: Marker(1, 0.1)}, links=[Link(0, 1, 1.0), Link(0, 1, 1.1, sigma_m=0.01), Link(0, 1, 0.9, 1, 1, sigma_m=0.004)])
    cs = ConstraintSet.from_marker_set(ms)
    assert cs.centroid_distances == (CentroidDistanceConstraint(0, 1, 1.0, 0.005), CentroidDistanceConstraint(0, 1, 1.1, 0.01))
    assert cs.distances[-1].sigma == 0.004 and len(cs.distances) == 13
    # thick mirror pair -> four ties; thin one -> four remaps, no rows for marker b, b not static
    ms = MarkerSet({0: Marker(0, 0.1, static=True), 1: Marker(1, 0.1, static=True)}, mirror_pairs=[Mirror(0, 1, 0.004, sigma_m=0.001)])
    cs = ConstraintSet.from_marker_set(ms)
    ties = [d for d in cs.distances if d.object_id_a != d.object_id_b]
    assert len(ties) == 4 and all(d.distance == 0.004 and d.sigma == 0.001 for d in ties) and cs.static_object_ids == {0, 1}
    ms = MarkerSet({0: Marker(0, 0.1, static=True), 1: Marker(1, 0.1, static=True)}, mirror_pairs=[Mirror(0, 1)])
    cs = ConstraintSet.from_marker_set(ms)
    assert len(cs.distances) == 6 and len(cs.point_remaps) == 4 and cs.static_object_ids == {0}
    assert cs.point_remaps[0] == PointRemap(1, 1, 0, 0, -0.05, 0.05, 0.0)


def test_remap_image_points():
    df = pd.DataFrame({"sync_index": [0, 0, 0], "cam_id": [0, 0, 0], "object_id": [0, 1, 1], "keypoint_id": [0, 1, 2],
                       "img_loc_x": [1.0, 2.0, 3.0], "img_loc_y": [1.0, 2.0, 3.0], "obj_loc_x": [9.0, 9.0, 9.0],
                       "obj_loc_y": [9.0, 9.0, 9.0], "obj_loc_z": [9.0, 9.0, 9.0]})
    ip = ImagePoints(df)
    assert ConstraintSet((), frozenset()).remap_image_points(ip) is ip
    cs = ConstraintSet((), frozenset(), point_remaps=(PointRemap(1, 1, 0, 0, -0.05, 0.05, 0.0),))
    out = cs.remap_image_points(ip).df
    assert list(out["object_id"]) == [0, 0, 1] and list(out["keypoint_id"]) == [0, 0, 2]
    assert list(out["obj_loc_x"]) == [9.0, -0.05, 9.0] and list(out["img_loc_x"]) == [1.0, 2.0, 3.0]


def test_toml_round_trip(tmp_path):
    cs = ConstraintSet(
        distances=(DistanceConstraint(0, 1, 2, 3, 0.25, 0.002), DistanceConstraint(4, 0, 4, 1, 0.1, 0.001)),
        static_object_ids=frozenset({4, 2}),
        centroid_distances=(CentroidDistanceConstraint(0, 2, 1.5, 0.005),),
        point_remaps=(PointRemap(1, 1, 0, 0, -0.05, 0.05, 0.0),),
        back_face_thickness_m=0.003,
    )
    cs.to_toml(tmp_path / "sub" / "constraints.toml")
    assert ConstraintSet.from_toml(tmp_path / "sub" / "constraints.toml") == cs
    plain = ConstraintSet(distances=cs.distances, static_object_ids=frozenset())
    plain.to_toml(tmp_path / "plain.toml")
    text = (tmp_path / "plain.toml").read_text()
    assert "centroid_distances" not in text and "point_remaps" not in text and "back_face_thickness_m" not in text
    assert ConstraintSet.from_toml(tmp_path / "plain.toml") == plain
    with pytest.raises(PersistenceError):
        ConstraintSet.from_toml(tmp_path / "missing.toml")
    (tmp_path / "broken.toml").write_text("distances = [{object_id_a = 1}]\n")
    with pytest.raises(PersistenceError):
        ConstraintSet.from_toml(tmp_path / "broken.toml")


# ---- CaptureVolume marshalling ---------------------------------------------------------------------------------------
def _volume(world_rows, img_rows, cs):
    return CaptureVolume(one_camera(), ImagePoints(pd.DataFrame(img_rows)), WorldPoints(pd.DataFrame(world_rows)), cs)


def _rows(frames, objects, static=False, present=lambda si, o, k: True):
    offs = [(-0.5, 0.5), (0.5, 0.5), (0.5, -0.5), (-0.5, -0.5)]
    world, img = [], []
    for si in frames:
        for o, cx in objects:
            for k, (dx, dy) in enumerate(offs):
                if not present(si, o, k):
                    continue
                img.append(dict(sync_index=si, cam_id=0, object_id=o, keypoint_id=k, img_loc_x=200.0 + 40 * (cx + dx), img_loc_y=200.0 + 40 * dy))
                if not static:
                    world.append(dict(sync_index=si, object_id=o, keypoint_id=k, x_coord=cx + dx, y_coord=dy, z_coord=0.0, frame_time=si * 0.1))
    if static:
        for o, cx in objects:
            for k, (dx, dy) in enumerate(offs):
                world.append(dict(sync_index=STATIC_SYNC_INDEX, object_id=o, keypoint_id=k, x_coord=cx + dx, y_coord=dy, z_coord=0.0, frame_time=np.nan))
    return world, img


def test_mobile_marker_fires_every_frame():
    cs = ConstraintSet.from_marker_set(MarkerSet({0: Marker(0, 1.0)}))
    vol = _volume(*_rows([0, 1, 2], [(0, 0.0)]), cs)
    ga, gb, dist, sig = vol._build_constraint_arrays()
    assert ga.shape == gb.shape == (18, 4) and ga.dtype == np.int32 and len(dist) == len(sig) == 18  # reference test :531-596
    assert np.all(ga[:, :1] == ga) and np.all(gb[:, :1] == gb)
    df = vol.world_points.df
    assert np.all(df["sync_index"].to_numpy()[ga[:, 0]] == df["sync_index"].to_numpy()[gb[:, 0]])
    rep = vol.rigidity_report()
    assert len(rep.violations) == 18 and rep.rmse_mm == pytest.approx(0.0, abs=1e-9) and rep.max_violation_mm == pytest.approx(0.0, abs=1e-9)
    assert ConstraintSet((), frozenset()) and _volume(*_rows([0], [(0, 0.0)]), ConstraintSet((), frozenset()))._build_constraint_arrays() is None


def test_constraint_rows_are_built_once_per_volume_and_follow_optimize():
    """The firing table depends on the world points' KEYS and the constraint set only: built once, kept, handed to the volume ``optimize()``
    returns (same keys, new coordinates) — and equal to what a volume built from scratch over the same tables computes."""
    from caliscope_amd.engine import TrfResult

    cs = ConstraintSet.from_marker_set(MarkerSet({0: Marker(0, 1.0), 1: Marker(1, 0.5)}, links=[Link(0, 1, 2.0)]))
    vol = _volume(*_rows([0, 1, 2, 5], [(0, 0.0), (1, 2.0)], present=lambda si, o, k: not (si == 2 and o == 1 and k == 1)), cs)
    first = vol._build_constraint_arrays()
    assert vol._constraint_blocks() is vol._constraint_blocks()
    inst = list(vol._constraint_instances())
    assert len(inst) == len(first[2]) and [i[2] for i in inst] == first[0].tolist() and [i[3] for i in inst] == first[1].tolist()
    assert [i[0].distance for i in inst] == first[2].tolist() and all(isinstance(i[1], int) for i in inst)

    class Shift:  # an engine that moves every point by a millimetre
        def __init__(self, problem):
            self.ncp = problem.parameterization.n_camera_params

        def solve(self, x0, **kw):
            x = x0.copy()
            x[self.ncp:] += 1e-3
            return TrfResult(x=x, cost=0.0, optimality=0.0, nfev=2, njev=2, status=2)

    out = vol.optimize(_engine_factory=Shift)
    assert out._constraint_blocks() is vol._constraint_blocks()
    fresh = CaptureVolume(out.camera_array, out.image_points, WorldPoints(out.world_points.df), cs)
    for a, b in zip(out._build_constraint_arrays(), fresh._build_constraint_arrays()):
        assert a.dtype == b.dtype and np.array_equal(a, b)
    assert out.rigidity_report().rmse_mm == pytest.approx(fresh.rigidity_report().rmse_mm)
    assert len(out.rigidity_report().violations) == len(first[2])


def test_centroid_needs_all_eight_corners_and_static_fires_once():
    cs = ConstraintSet((), frozenset(), centroid_distances=(CentroidDistanceConstraint(0, 1, 2.0, 0.005),))
    vol = _volume(*_rows([0, 1, 2], [(0, 0.0), (1, 2.0)], present=lambda si, o, k: not (si == 1 and o == 1 and k == 3)), cs)
    ga, gb, dist, _ = vol._build_constraint_arrays()
    assert ga.shape == (2, 4) and len(set(ga[0])) == 4 and len(set(gb[0])) == 4
    df = vol.world_points.df
    assert set(df["sync_index"].to_numpy()[ga[:, 0]]) == {0, 2}
    assert set(df["object_id"].to_numpy()[ga.ravel()]) == {0} and set(df["object_id"].to_numpy()[gb.ravel()]) == {1}
    rep = vol.rigidity_report()
    assert [v.kind for v in rep.violations] == ["centroid"] * 2 and rep.violations[0].keypoint_id_a == -1
    assert rep.violations[0].actual == pytest.approx(2.0)
    # static-static: once, at STATIC_SYNC_INDEX (reference test :957-1011); static-mobile never
    cs = ConstraintSet((), frozenset({0, 1}), centroid_distances=(CentroidDistanceConstraint(0, 1, 2.0, 0.005),))
    vol = _volume(*_rows([0, 1], [(0, 0.0), (1, 2.0)], static=True), cs)
    ga, gb, dist, _ = vol._build_constraint_arrays()
    assert ga.shape == (1, 4) and np.all(vol.img_to_obj_map >= 0)
    assert np.all(vol.world_points.df["sync_index"].to_numpy()[ga.ravel()] == STATIC_SYNC_INDEX)
    mixed = ConstraintSet((DistanceConstraint(0, 0, 1, 0, 2.0, 0.002),), frozenset({0}))
    world_s, img_s = _rows([0, 1], [(0, 0.0)], static=True)
    world_m, img_m = _rows([0, 1], [(1, 2.0)])
    assert _volume(world_s + world_m, img_s + img_m, mixed)._build_constraint_arrays() is None


def test_rigidity_report_of_a_deformed_square():
    cs = ConstraintSet.from_marker_set(MarkerSet({0: Marker(0, 1.0)}))
    world, img = _rows([0], [(0, 0.0)])
    world[1]["x_coord"] += 0.1  # corner 1 pulled 0.1 m along x
    rep = _volume(world, img, cs).rigidity_report()
    err = {(v.keypoint_id_a, v.keypoint_id_b): v.actual - v.expected for v in rep.violations}
    assert err[(0, 1)] == pytest.approx(0.1) and err[(2, 3)] == pytest.approx(0.0, abs=1e-12)
    assert err[(1, 2)] == pytest.approx(np.hypot(0.1, 1.0) - 1.0)
    assert rep.max_violation_mm == pytest.approx(100.0)
    assert rep.rmse_mm == pytest.approx(np.sqrt(np.mean(np.square(list(err.values())))) * 1000)
    assert rep.relative_rmse_pct == pytest.approx(np.sqrt(np.mean([(e / v.expected) ** 2 for e, v in zip(err.values(), rep.violations)])) * 100)
    assert set(rep.per_object_rmse_mm) == {0} and rep.per_object_rmse_mm[0] == pytest.approx(rep.rmse_mm)


def test_save_and_load_keep_the_constraints(tmp_path):
    cs = ConstraintSet.from_marker_set(MarkerSet({0: Marker(0, 1.0, static=True), 1: Marker(1, 1.0, static=True)}, links=[Link(0, 1, 2.0)]))
    vol = _volume(*_rows([0, 1], [(0, 0.0), (1, 2.0)], static=True), cs)
    vol.save(tmp_path / "vol")
    back = CaptureVolume.load(tmp_path / "vol")
    assert back.constraints == cs and np.array_equal(back.img_to_obj_map, vol.img_to_obj_map)
    ga, _, dist, sig = back._build_constraint_arrays()
    assert ga.shape == (13, 4) and sorted(set(sig)) == [0.002, 0.005]


def test_optimize_with_constraints_matches_scipy_on_the_oracle():
    """CaptureVolume.optimize(use_constraints=True) through the numpy engine vs scipy on the same rows."""
    from oracle.engine import OracleEngine
    from oracle.solver import optimize_scipy
    from tests.constrained_scene import board_volume
    from tests.helpers import aligned_difference

    vol, sc = board_volume()
    factory = lambda prob: OracleEngine(prob.parameterization, prob.camera_indices, prob.image_coords, prob.obj_indices,
                                        loss=prob.loss, f_scale=prob.f_scale, constraints=prob.constraint_args())
    out = vol.optimize(_engine_factory=factory)
    assert out.optimization_status.converged
    ga, gb, dist, sig = vol._build_constraint_arrays()
    w = (1.0 / 1394.6) / sig
    _, cam, uv, obj = vol._matched_arrays()
    ref = optimize_scipy(sc["par"], cam, uv, obj, sc["x0"], constraints=(ga, gb, dist, w))
    assert abs(out.optimization_status.final_cost - ref.cost) <= 1e-8 * ref.cost
    x = sc["par"].pack(out.camera_array, out.world_points.points)
    pos, ang, scale = aligned_difference(sc["par"], x, ref.x)
    assert pos < 1e-6 and ang < 1e-6 and abs(scale - 1) < 1e-6
    free = vol.optimize(use_constraints=False, _engine_factory=factory)
    assert out.rigidity_report().rmse_mm < free.rigidity_report().rmse_mm


def _naive_constraint_rows(world_df, cs):
    """The reference's rule restated with dicts and sets (core/capture_volume.py:446-531): a dict of rows per keypoint (a later row of the same key
    overwrites an earlier one), a constraint fires at the sync indices all of its endpoint keypoints share — static-static ones at
    STATIC_SYNC_INDEX only, mobile-mobile ones everywhere else, mixed ones never.  Returns {(kind, constraint index, sync): (rows_a, rows_b)}."""
    lookup = {}
    for row, (si, oid, kid) in enumerate(zip(world_df["sync_index"], world_df["object_id"], world_df["keypoint_id"])):
        lookup.setdefault((int(oid), int(kid)), {})[int(si)] = row
    static_ids, out = cs.static_object_ids, {}

    def firing(is_static, tables):
        if is_static:
            return [STATIC_SYNC_INDEX] if all(STATIC_SYNC_INDEX in t for t in tables) else []
        return [si for si in set.intersection(*(set(t) for t in tables)) if si != STATIC_SYNC_INDEX]

    for n, dc in enumerate(cs.distances):
        a_static, b_static = dc.object_id_a in static_ids, dc.object_id_b in static_ids
        if a_static != b_static:
            continue
        ta, tb = lookup.get((dc.object_id_a, dc.keypoint_id_a), {}), lookup.get((dc.object_id_b, dc.keypoint_id_b), {})
        for si in firing(a_static, (ta, tb)):
            out[("d", n, si)] = ([ta[si]] * 4, [tb[si]] * 4)
    for n, cc in enumerate(cs.centroid_distances):
        a_static, b_static = cc.object_id_a in static_ids, cc.object_id_b in static_ids
        if a_static != b_static:
            continue
        ca = [lookup.get((cc.object_id_a, k), {}) for k in range(4)]
        cb = [lookup.get((cc.object_id_b, k), {}) for k in range(4)]
        for si in firing(a_static, (*ca, *cb)):
            out[("c", n, si)] = ([ca[k][si] for k in range(4)], [cb[k][si] for k in range(4)])
    return out


def _random_constraint_case(seed):
    """Random objects (some static), frames with holes, shuffled rows, a few DUPLICATE world keys; distance and centroid constraints between random
    keypoints, some of them mixing a static with a moving object, some naming keypoints that never appear."""
    rng = np.random.default_rng(seed)
    n_obj = int(rng.integers(2, 6))
    static = {o for o in range(n_obj) if rng.random() < 0.3}
    frames = sorted(rng.choice(40, size=int(rng.integers(3, 12)), replace=False).tolist())
    world, img = [], []
    for o in range(n_obj):
        for k in range(4):
            for si in ([STATIC_SYNC_INDEX] if o in static else frames):
                if rng.random() < 0.2:
                    continue  # a hole: the keypoint was not reconstructed here
                for _ in range(2 if rng.random() < 0.05 else 1):  # now and then the same key twice
                    world.append(dict(sync_index=si, object_id=o, keypoint_id=k, x_coord=rng.normal(), y_coord=rng.normal(), z_coord=rng.normal(),
                                      frame_time=np.nan if o in static else si * 0.1))
        for si in frames:  # every object is observed at every frame (static ones included: they look their point up at STATIC_SYNC_INDEX)
            for k in range(4):
                img.append(dict(sync_index=si, cam_id=0, object_id=o, keypoint_id=k, img_loc_x=200.0 + rng.normal(), img_loc_y=200.0 + rng.normal()))
    order = rng.permutation(len(world))
    world = [world[i] for i in order]
    dist = tuple(DistanceConstraint(int(rng.integers(0, n_obj)), int(rng.integers(0, 5)), int(rng.integers(0, n_obj)), int(rng.integers(0, 5)),
                                    float(rng.uniform(0.1, 2.0)), float(rng.uniform(0.001, 0.01))) for _ in range(int(rng.integers(1, 12))))
    cent = tuple(CentroidDistanceConstraint(int(a), int(b), float(rng.uniform(0.1, 2.0)), 0.005)
                 for a, b in rng.integers(0, n_obj, size=(int(rng.integers(0, 4)), 2)) if a != b)
    cs = ConstraintSet(dist, frozenset(static), centroid_distances=cent)
    with _no_warning_filter():
        return _volume(world, img, cs), cs


@pytest.mark.parametrize("seed", range(12))
def test_firing_table_against_the_reference_rule_on_random_tables(seed):
    vol, cs = _random_constraint_case(seed)
    expected = _naive_constraint_rows(vol.world_points.df, cs)
    got = {}
    index_of = {id(c): ("d", n) for n, c in enumerate(cs.distances)} | {id(c): ("c", n) for n, c in enumerate(cs.centroid_distances)}
    for c, si, rows_a, rows_b in vol._constraint_instances():
        kind, n = index_of[id(c)]
        assert (kind, n, si) not in got
        got[(kind, n, si)] = (rows_a, rows_b)
    assert got == expected
    arrays = vol._build_constraint_arrays()
    if not expected:
        assert arrays is None
        return
    ga, gb, d, sig = arrays
    assert ga.shape == gb.shape == (len(expected), 4) and ga.dtype == gb.dtype == np.int32
    assert [r[2] for r in vol._constraint_instances()] == ga.tolist() and [r[3] for r in vol._constraint_instances()] == gb.tolist()
    assert d.tolist() == [r[0].distance for r in vol._constraint_instances()] and sig.tolist() == [r[0].sigma for r in vol._constraint_instances()]


def test_the_random_tables_exercise_something():
    """Between them the twelve cases fire constraints, leave some silent and contain duplicate world keys."""
    fired, dupes = [], 0
    for seed in range(12):
        vol, cs = _random_constraint_case(seed)
        fired.append(len(_naive_constraint_rows(vol.world_points.df, cs)))
        dupes += int(vol.world_points.df.duplicated(subset=["sync_index", "object_id", "keypoint_id"]).sum())
    assert sum(fired) > 50 and dupes > 5, (fired, dupes)


def _no_warning_filter():
    import contextlib
    import warnings

    @contextlib.contextmanager
    def ctx():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # (duplicate keys are warned about by the tables, as in the reference)
            yield

    return ctx()
