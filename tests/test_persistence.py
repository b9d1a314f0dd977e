"""On-disk formats of the path: camera_array.toml, the two CSV tables, capture-volume save/load."""

import numpy as np
import pandas as pd
import pytest
import tomli

from caliscope_amd.cameras import CameraArray, CameraData, rvec_to_matrix
from caliscope_amd.capture_volume import CaptureVolume
from caliscope_amd.persistence import dumps_toml
from caliscope_amd.point_data import ImagePoints, WorldPoints
from caliscope_amd.synthetic import ring_camera_array

GOLDEN = "tests/golden/post_optimization"


def test_toml_writer_round_trips_through_tomli():
    data = {"cameras": {"0": {"cam_id": 0, "size": [1920, 1080], "error": 0.25, "fisheye": False, "matrix": [[1.5, 0.0, 2.0], [0.0, 1.5, 3.0], [0.0, 0.0, 1.0]],
                              "name": 'a "quoted" name', "big": 1e-12, "neg": -3}},
            "metadata": {"adjusted": False, "error": 0.0}}
    assert tomli.loads(dumps_toml(data)) == data


def test_camera_array_round_trip(tmp_path):
    cams = ring_camera_array(5)
    cams.cameras[2].rotation = np.eye(3)          # identity rotation = all-zero Rodrigues vector must survive
    cams.cameras[3].rotation = None               # unposed camera: pose keys omitted
    cams.cameras[3].translation = None
    cams.cameras[4].fisheye = True
    cams.cameras[4].distortions = np.array([0.05, -0.01, 0.003, -0.001])
    cams.cameras[1].error = 0.31
    path = tmp_path / "sub" / "camera_array.toml"
    cams.to_toml(path)
    assert not path.with_suffix(".toml.tmp").exists()
    back = CameraArray.from_toml(path)
    assert sorted(back.cameras) == sorted(cams.cameras)
    for c, cam in cams.cameras.items():
        b = back.cameras[c]
        assert b.fisheye == cam.fisheye and tuple(b.size) == tuple(cam.size) and b.error == cam.error
        assert np.allclose(b.matrix, cam.matrix, rtol=0, atol=0) and np.allclose(b.distortions, np.ravel(cam.distortions), rtol=0, atol=0)
        if cam.rotation is None:
            assert b.rotation is None and b.translation is None
        else:
            assert np.abs(b.rotation - cam.rotation).max() < 1e-14 and np.abs(np.ravel(b.translation) - np.ravel(cam.translation)).max() == 0
    assert 3 not in back.posed_cameras and len(back.posed_cameras) == 4


def test_reference_file_survives_a_rewrite(tmp_path):
    """The reference's own camera_array.toml (golden fixture): load -> save -> load is the identity on every field."""
    a = CameraArray.from_toml(f"{GOLDEN}/camera_array.toml")
    a.to_toml(tmp_path / "camera_array.toml")
    b = CameraArray.from_toml(tmp_path / "camera_array.toml")
    for c in a.cameras:
        assert np.abs(a.cameras[c].rotation - b.cameras[c].rotation).max() < 1e-13
        assert np.abs(np.ravel(a.cameras[c].translation) - np.ravel(b.cameras[c].translation)).max() == 0
        assert np.array_equal(a.cameras[c].matrix, b.cameras[c].matrix)


def test_aniposelib_export(tmp_path):
    cams = ring_camera_array(3)
    cams.to_aniposelib_toml(tmp_path / "anipose.toml")
    doc = tomli.loads((tmp_path / "anipose.toml").read_text())
    assert doc["metadata"] == {"adjusted": False, "error": 0.0}
    assert set(doc) == {"cam_0", "cam_1", "cam_2", "metadata"} and doc["cam_1"]["name"] == "cam_1"
    assert np.abs(rvec_to_matrix(np.array(doc["cam_2"]["rotation"])) - cams.cameras[2].rotation).max() < 1e-14


def test_capture_volume_save_load(tmp_path):
    cams = CameraArray.from_toml(f"{GOLDEN}/camera_array.toml")
    ip = ImagePoints.from_csv(f"{GOLDEN}/image_points.csv") if (__import__("pathlib").Path(GOLDEN) / "image_points.csv").exists() else None
    if ip is None:
        ip = ImagePoints.from_csv(f"{GOLDEN}/xy_CHARUCO.csv")
        wp = WorldPoints.from_csv(f"{GOLDEN}/xyz_CHARUCO.csv")
    else:
        wp = WorldPoints.from_csv(f"{GOLDEN}/world_points.csv")
    vol = CaptureVolume(camera_array=cams, image_points=ip, world_points=wp)
    vol.save(tmp_path / "vol")
    assert sorted(p.name for p in (tmp_path / "vol").iterdir()) == ["camera_array.toml", "image_points.csv", "world_points.csv"]
    back = CaptureVolume.load(tmp_path / "vol")
    assert len(back.image_points) == len(ip) and len(back.world_points) == len(wp)
    assert np.array_equal(back.img_to_obj_map, vol.img_to_obj_map)
    a, b = wp.df[["x_coord", "y_coord", "z_coord"]].to_numpy(), back.world_points.df[["x_coord", "y_coord", "z_coord"]].to_numpy()
    assert np.abs(a - b).max() <= 5e-7  # %.6f, as the reference writes


def test_camera_helpers_without_opencv():
    from caliscope_amd.cameras import CameraArray, CameraData, rvec_to_matrix
    from caliscope_amd.exceptions import CalibrationError

    cam = CameraData.from_intrinsics(3, (1920, 1080), 1400.0)
    assert cam.matrix[0, 0] == cam.matrix[1, 1] == 1400.0 and (cam.matrix[0, 2], cam.matrix[1, 2]) == (960.0, 540.0) and np.all(cam.distortions == 0)
    cam2 = CameraData.from_intrinsics(3, (640, 480), fx=500.0, fy=510.0, cx=300.0, distortions=[0.1, 0, 0, 0, 0])
    assert (cam2.matrix[0, 0], cam2.matrix[1, 1], cam2.matrix[0, 2], cam2.matrix[1, 2]) == (500.0, 510.0, 300.0, 240.0) and cam2.distortions[0] == 0.1
    for bad in (dict(focal_length=1.0, fx=1.0), dict(fx=1.0), {}):
        with pytest.raises(ValueError):
            CameraData.from_intrinsics(0, (10, 10), **bad)
    with pytest.raises(ValueError):
        _ = cam.transformation
    T = np.eye(4); T[:3, :3] = rvec_to_matrix([0.1, -0.2, 0.3]); T[:3, 3] = [1.0, 2.0, 3.0]
    cam.transformation = T
    assert np.allclose(cam.transformation, T) and np.allclose(cam.normalized_projection_matrix, T[:3])
    cam.erase_calibration_data()
    assert cam.matrix is None and cam.rotation is None and cam.translation is None
    cam.synthesize_default_intrinsics()
    assert cam.matrix[0, 0] == 960.0 and (cam.matrix[0, 2], cam.matrix[1, 2]) == (960.0, 540.0)
    with pytest.raises(CalibrationError):
        CameraData(cam_id=1, size=(10, 10), fisheye=True).synthesize_default_intrinsics()
    arr = CameraArray.from_image_sizes({2: (1280, 720), 0: (640, 480)})
    assert arr.all_cameras_have_resolution() and not arr.all_intrinsics_calibrated() and not arr.all_extrinsics_calibrated()
    assert CameraArray().all_extrinsics_calibrated() and not CameraArray().all_intrinsics_calibrated() and not CameraArray().all_cameras_have_resolution()
    for c in arr.cameras.values():
        c.synthesize_default_intrinsics(); c.rotation = np.eye(3); c.translation = np.zeros(3)
    assert arr.all_intrinsics_calibrated() and arr.all_extrinsics_calibrated()


def test_from_toml_failures_raise_persistence_error(tmp_path):
    """The reference's repository layer catches PersistenceError (cameras/camera_array.py:377-470): a missing file, invalid TOML and a
    malformed camera entry must all surface as that type, never as an unrelated exception."""
    from caliscope_amd.persistence import PersistenceError

    with pytest.raises(PersistenceError, match="not found"):
        CameraArray.from_toml(tmp_path / "missing.toml")
    bad = tmp_path / "bad.toml"
    bad.write_text("[cameras\nthis is = not toml")
    with pytest.raises(PersistenceError, match="Failed to load"):
        CameraArray.from_toml(bad)
    shape = tmp_path / "shape.toml"
    shape.write_text('[cameras.0]\nsize = [640, 480]\nrotation = [1.0, 2.0]\n')
    with pytest.raises(PersistenceError, match="Failed to parse camera 0"):
        CameraArray.from_toml(shape)
    nosize = tmp_path / "nosize.toml"
    nosize.write_text('[cameras.3]\nrotation_count = 0\n')
    with pytest.raises(PersistenceError, match="Failed to parse camera 3"):
        CameraArray.from_toml(nosize)
