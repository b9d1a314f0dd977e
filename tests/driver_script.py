"""Scripted stand-ins for the three heavy calls of ``calibrate_extrinsics`` — ``CaptureVolume.bootstrap``, ``optimize``,
``filter_by_percentile_error`` — applied to a ``CaptureVolume`` class given by the caller: the reference's (in
tests/golden/make_reference_host_fixtures.py, build container) or this package's (tests/test_reference_host_fixtures.py).  Everything else of the
driver runs for real on both sides: blind intrinsics, anchors, observation remaps, the static-marker guard on the rigidity report, the depth-ratio
gate, the stage sequence with its arguments, the final ``CalibrationRun``.  ``trace`` receives what the driver did, in order."""
from contextlib import contextmanager
from copy import deepcopy

import numpy as np


@contextmanager
def scripted(volume_cls, world_points_cls, world_table, trace):
    """``world_table``: DataFrame of "triangulated" world points (all objects); bootstrap hands back the rows of the objects the image points
    still contain, so a re-bootstrap after dropping a marker sees a table without it."""
    saved = {name: volume_cls.__dict__[name] for name in ("bootstrap", "optimize", "filter_by_percentile_error")}

    def bootstrap(cls, image_points, camera_array, constraints=None, **kwargs):
        present = set(int(o) for o in image_points.df["object_id"].unique())
        rows = world_table[world_table["object_id"].isin(present)].reset_index(drop=True)
        trace.append(("bootstrap", sorted(int(c) for c in camera_array.cameras), int(len(image_points.df)), int(len(rows)),
                      None if constraints is None else (len(constraints.distances), len(constraints.centroid_distances), sorted(constraints.static_object_ids))))
        return cls(camera_array=deepcopy(camera_array), image_points=image_points, world_points=world_points_cls(rows), constraints=constraints)

    def optimize(self, *args, **kwargs):
        assert not args
        trace.append(("optimize", sorted((k, (round(v, 15) if isinstance(v, float) else v)) for k, v in kwargs.items() if not k.startswith("_"))))
        cams = deepcopy(self.camera_array)
        if kwargs.get("refine_intrinsics"):  # a "refined" camera: something for the intrinsic estimates to report
            for cid in cams.posed_cam_id_to_index:
                cam = cams.cameras[cid]
                cam.matrix = cam.matrix.copy()
                cam.matrix[0, 0] *= 1.0 + 0.001 * (cid + 1)
                cam.matrix[1, 1] *= 1.0 + 0.001 * (cid + 1)
                d = np.array(cam.distortions, dtype=np.float64).ravel()
                d[0] += 0.01
                d[1] -= 0.02
                cam.distortions = d
        return type(self)(camera_array=cams, image_points=self.image_points, world_points=self.world_points, constraints=self.constraints)

    def filter_by_percentile_error(self, percentile, *args, **kwargs):
        trace.append(("filter", float(percentile), sorted(k for k in kwargs if not k.startswith("_")), len(args)))
        return self

    volume_cls.bootstrap = classmethod(bootstrap)
    volume_cls.optimize = optimize
    volume_cls.filter_by_percentile_error = filter_by_percentile_error
    try:
        yield
    finally:
        for name, value in saved.items():
            setattr(volume_cls, name, value)
