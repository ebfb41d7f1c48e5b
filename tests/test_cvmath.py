"""Self-consistency of the OpenCV restatements in pose2sim_amd/cvmath.py (nothing under the reference pins
their outputs, so these are round trips and limiting cases, not parity with OpenCV -- DESIGN.md section 2)."""
import numpy as np

from pose2sim_amd import cvmath, synth


def test_rodrigues_round_trips():
    rng = np.random.default_rng(0)
    for _ in range(200):
        r = rng.normal(0, 1, 3) * rng.choice([1e-9, 1e-3, 1.0, 3.0])
        R = cvmath.rodrigues(r)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and abs(np.linalg.det(R) - 1.0) < 1e-13
        th = np.linalg.norm(r)
        if 1e-4 < th < np.pi - 1e-3:                           # the vector is unique below a half turn
            assert np.allclose(cvmath.rodrigues_from_matrix(R), r, atol=1e-11)
            assert np.allclose(cvmath.rodrigues_inv(R), r, atol=1e-9)
        elif th <= 1e-4:                                       # OpenCV returns 0 below sin(theta) = 1e-5
            assert np.linalg.norm(cvmath.rodrigues_from_matrix(R) - r) < 2e-5
        assert np.allclose(cvmath.rodrigues(cvmath.rodrigues_from_matrix(R)), R, atol=1e-8)
    # half turns: the diagonal branch of the matrix -> vector direction
    for axis in np.eye(3):
        R = cvmath.rodrigues(np.pi * axis)
        assert np.allclose(cvmath.rodrigues(cvmath.rodrigues_from_matrix(R)), R, atol=1e-9)
    assert np.allclose(cvmath.rodrigues_from_matrix(np.eye(3)), 0.0)


def test_project_then_undistort_is_the_pinhole_projection():
    """projectPoints with distortion followed by undistortPoints(..., newK) lands on the distortion-free pixel of
    the same ray under newK, to the float32 output rounding (~1e-4 px at 2 000 px) and the 5 fixed-point passes."""
    cams = synth.make_cameras(6, seed=4, distort=True)
    rng = np.random.default_rng(1)
    Q = rng.uniform(-1.0, 1.0, (300, 3)) + np.array([0.0, 0.0, 1.0])
    worst = 0.0
    for c in range(6):
        K, dist, R, T = cams['K'][c], cams['dist'][c], cams['R_mat'][c], cams['T'][c]
        newK = cams['optim_K'][c]
        uv = cvmath.project_points(Q, R, T, K, dist)
        und = cvmath.undistort_points(uv, K, dist, newK)
        Xc = Q @ np.asarray(R).T + np.asarray(T).reshape(1, 3)
        ideal = (Xc[:, :2] / Xc[:, 2:3]) @ np.diag([newK[0, 0], newK[1, 1]]) + np.array([newK[0, 2], newK[1, 2]])
        front = Xc[:, 2] > 0.5
        worst = max(worst, float(np.abs(und - ideal)[front].max()))
    assert worst < 5e-3, worst
    # without distortion both are exact inverses up to the float32 rounding of undistortPoints
    K = cams['K'][0]
    uv = cvmath.project_points(Q, cams['R_mat'][0], cams['T'][0], K, np.zeros(4))
    assert np.abs(cvmath.undistort_points(uv, K, np.zeros(4), K) - uv.astype(np.float32)).max() < 5e-4


def test_optimal_new_camera_matrix_limits():
    K = np.array([[1400.0, 0.0, 960.0], [0.0, 1390.0, 540.0], [0.0, 0.0, 1.0]])
    same = cvmath.get_optimal_new_camera_matrix(K, np.zeros(4), (1920, 1080), 1.0, (1920, 1080))
    assert np.allclose(same, K, atol=1e-6)                     # no distortion: nothing to adapt
    d = np.array([-0.08, 0.03, 5e-4, -3e-4])
    newK = cvmath.get_optimal_new_camera_matrix(K, d, (1920, 1080), 1.0, (1920, 1080))
    assert newK.shape == (3, 3) and newK[2, 2] == 1.0 and newK[0, 1] == 0.0
    assert 0.5 * K[0, 0] < newK[0, 0] < 1.5 * K[0, 0] and 0.5 * K[1, 1] < newK[1, 1] < 1.5 * K[1, 1]
