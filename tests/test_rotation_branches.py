"""`extrinsics_to_vector` at the edge of `cv2.Rodrigues(R)`'s domain.  The fixtures generated from the reference's own code
(tests/golden/reference_host/) convert rotations with scipy's `Rotation` in OpenCV's place, so the branch OpenCV takes near theta = pi is pinned HERE,
against the formula of SURVEY.md Appendix A.1 restated independently: `c = (tr R - 1) / 2`, `theta = acos c`, `v = (R32 - R23, R13 - R31, R21 - R12)`,
`s = |v| / 2`; `s < 1e-5` and `c <= 0`: `r_i = theta * sqrt(max((R_ii + 1) / 2, 0))`, signs from the off-diagonals (r_x >= 0; r_y takes the sign of
R12, r_z of R13; when r_x is the smallest entry, r_z's sign follows R23's sign relative to r_y's), then scaled to length theta; `c > 0`: r = 0; otherwise
`r = v * theta / (2 s)`."""
import numpy as np
import pytest

from caliscope_amd.cameras import matrix_to_rvec, rvec_to_matrix


def _appendix_a1(R):
    u, _, vt = np.linalg.svd(R)  # (OpenCV re-orthonormalises first; the appendix calls it irrelevant for valid rotations, but the half-turn branch takes
    R = u @ vt                   # square roots of (R_ii + 1) / 2 ~ 0: a rounding of 1e-16 in R is 1e-8 in r there)
    c = min(1.0, max(-1.0, 0.5 * (np.trace(R) - 1.0)))
    theta = np.arccos(c)
    v = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = 0.5 * np.linalg.norm(v)
    if s >= 1e-5:
        return v * theta / (2.0 * s)
    if c > 0:
        return np.zeros(3)
    r = np.sqrt(np.maximum((np.diag(R) + 1.0) / 2.0, 0.0))
    if R[0, 1] < 0:
        r[1] = -r[1]
    if R[0, 2] < 0:
        r[2] = -r[2]
    if abs(r[0]) < abs(r[1]) and abs(r[0]) < abs(r[2]) and ((R[1, 2] > 0) != (r[1] * r[2] > 0)):
        r[2] = -r[2]
    return r * theta / np.linalg.norm(r)


def _exact(axis, theta):
    """Rodrigues' formula in extended precision, rounded once: the matrix a camera file would hold."""
    k = np.asarray(axis, dtype=np.longdouble)
    k = k / np.sqrt(k @ k)
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]], dtype=np.longdouble)
    th = np.longdouble(theta)
    return (np.eye(3, dtype=np.longdouble) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)).astype(np.float64)


AXES = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (0.6, -0.8, 0.0), (1, 2, -3), (-1, 2, 3), (0.01, 1, 1), (0.01, 1, -1), (0.01, -1, 1), (1, 1e-3, -1), (-2, -1, -0.5)]


@pytest.mark.parametrize("axis", AXES)
@pytest.mark.parametrize("gap", [0.0, 1e-9, 5e-6, 2e-5, 1e-3])
def test_rotation_vector_near_pi_follows_appendix_a1(axis, gap):
    R = _exact(axis, np.pi - gap)
    got, want = matrix_to_rvec(R), _appendix_a1(R)
    assert np.allclose(got, want, rtol=0, atol=1e-10), (axis, gap, got, want)
    # ... and is a rotation vector OF that matrix — to O(gap) inside the half-turn branch, which reads the axis off the diagonal and drops what the
    # antisymmetric part (|v| / 2 = sin(gap) < 1e-5) says: OpenCV's own behaviour, the reference inherits it
    assert np.allclose(rvec_to_matrix(got), R, atol=max(1e-7, 4 * gap if gap < 1.5e-5 else 1e-7)), (axis, gap)
    assert abs(np.linalg.norm(got) - (np.pi - gap)) < 1e-6


def test_small_angles_and_identity():
    assert np.array_equal(matrix_to_rvec(np.eye(3)), np.zeros(3))
    for axis in AXES:
        for theta in (1e-9, 1e-6, 9e-6, 1.1e-5, 1e-3):
            R = _exact(axis, theta)
            got, want = matrix_to_rvec(R), _appendix_a1(R)
            assert np.allclose(got, want, rtol=0, atol=1e-15)
            k = np.asarray(axis, float) / np.linalg.norm(axis)
            if theta > 2e-5:  # (below OpenCV's threshold s < 1e-5 the answer is the zero vector: the reference's files never hold such a camera)
                assert np.allclose(got, k * theta, rtol=1e-6, atol=1e-12)
            elif theta < 9.5e-6:
                assert np.array_equal(got, np.zeros(3))
