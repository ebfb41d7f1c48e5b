"""Camera-model arithmetic the reference takes from OpenCV, restated in NumPy.

The reference calls ``cv2.Rodrigues`` (common.py:284,318), ``cv2.getOptimalNewCameraMatrix``
(common.py:281,314), ``cv2.undistortPoints`` (triangulation.py:811) and ``cv2.projectPoints``
(triangulation.py:473,535).  OpenCV is not part of this image, and nothing under the reference
pins any of their outputs, so these are restatements of OpenCV's published algorithms
(calib3d: Brown-Conrady model k1,k2,p1,p2[,k3[,k4,k5,k6]], 5 fixed-point iterations for the
inverse).  PARITY UNPINNED against OpenCV itself; they are self-checked by round trip in
tests/test_cvmath.py.  These run once per calibration (host side, C cameras), never per unit;
the per-unit versions of undistort / project live in csrc/p2s_tri.hip (and, for the tests, in oracle/).
"""
import numpy as np


def rodrigues(rvec):
    """Rotation vector (3,) -> rotation matrix (3,3).  cv2.Rodrigues forward direction."""
    r = np.asarray(rvec, dtype=np.float64).reshape(3)
    theta = np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2])
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    c, s = np.cos(theta), np.sin(theta)
    c1 = 1.0 - c
    x, y, z = r / theta
    rrt = np.array([[x * x, x * y, x * z], [x * y, y * y, y * z], [x * z, y * z, z * z]])
    r_x = np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])
    return c * np.eye(3) + c1 * rrt + s * r_x


def rodrigues_inv(R):
    """Rotation matrix -> rotation vector (used by the synthetic generator only)."""
    R = np.asarray(R, dtype=np.float64)
    cos_t = np.clip((np.trace(R) - 1.0) / 2.0, -1.0, 1.0)
    theta = np.arccos(cos_t)
    if theta < 1e-12:
        return np.zeros(3)
    w = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    if np.pi - theta < 1e-6:
        # near pi: take the axis from the symmetric part
        A = (R + np.eye(3)) / 2.0
        axis = np.sqrt(np.clip(np.diag(A), 0.0, None))
        i = int(np.argmax(axis))
        axis = A[i] / axis[i]
        return theta * axis / np.linalg.norm(axis)
    return theta * w / (2.0 * np.sin(theta))


def rodrigues_from_matrix(R):
    """Rotation matrix -> rotation vector, cv2.Rodrigues' matrix branch restated from OpenCV's published
    algorithm: re-orthonormalise through the SVD, theta = acos((tr - 1) / 2), axis from the antisymmetric part
    (from the diagonal when sin(theta) < 1e-5).  Returns shape (3,)."""
    R = np.asarray(R, dtype=np.float64).reshape(3, 3)
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    rx, ry, rz = R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]
    s = np.sqrt((rx * rx + ry * ry + rz * rz) * 0.25)
    c = min(max((R[0, 0] + R[1, 1] + R[2, 2] - 1.0) * 0.5, -1.0), 1.0)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        rx = np.sqrt(max((R[0, 0] + 1.0) * 0.5, 0.0))
        ry = np.sqrt(max((R[1, 1] + 1.0) * 0.5, 0.0)) * (-1.0 if R[0, 1] < 0 else 1.0)
        rz = np.sqrt(max((R[2, 2] + 1.0) * 0.5, 0.0)) * (-1.0 if R[0, 2] < 0 else 1.0)
        if abs(rx) < abs(ry) and abs(rx) < abs(rz) and ((R[1, 2] > 0) != (ry * rz > 0)):
            rz = -rz
        theta /= np.sqrt(rx * rx + ry * ry + rz * rz)
        return np.array([rx, ry, rz]) * theta
    vth = theta / (2.0 * s)
    return np.array([rx, ry, rz]) * vth


def _dist12(dist):
    """Pad an OpenCV distortion vector (4, 5, 8, 12 or 14 terms) to the 12 terms used below."""
    d = np.zeros(12, dtype=np.float64)
    dist = np.asarray(dist, dtype=np.float64).ravel()
    n = min(len(dist), 12)
    d[:n] = dist[:n]
    return d


def undistort_normalized(u, v, K, dist, iters=5):
    """Pixel -> ideal normalised coordinates (cv2.undistortPoints core, MAX_ITER=5, no EPS test).

    k = (k1,k2,p1,p2,k3,k4,k5,k6,s1,s2,s3,s4); all arithmetic in float64.
    """
    K = np.asarray(K, dtype=np.float64)
    k = _dist12(dist)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    ifx, ify = 1.0 / fx, 1.0 / fy          # OpenCV multiplies by precomputed reciprocals
    u = np.asarray(u, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    x0 = (u - cx) * ifx
    y0 = (v - cy) * ify
    x, y = x0.copy(), y0.copy()
    done = np.zeros(x.shape, dtype=bool)
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
        neg = (icdist < 0) & ~done
        dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2
        dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2
        xn = (x0 - dx) * icdist
        yn = (y0 - dy) * icdist
        # icdist < 0: OpenCV falls back to the undistorted guess and stops iterating that point
        xn = np.where(neg, x0, xn)
        yn = np.where(neg, y0, yn)
        x = np.where(done, x, xn)
        y = np.where(done, y, yn)
        done = done | neg
    return x, y


def undistort_points(pts, K, dist, newK):
    """cv2.undistortPoints(pts.astype(float32), K, dist, None, newK) as used at triangulation.py:810-813.

    pts: (..., 2).  Input is rounded to float32, computed in float64, output rounded to float32
    (OpenCV returns CV_32FC2 for CV_32FC2 input) and handed back as float64 values.
    """
    pts = np.asarray(pts, dtype=np.float32).astype(np.float64)
    newK = np.asarray(newK, dtype=np.float64)
    x, y = undistort_normalized(pts[..., 0], pts[..., 1], K, dist)
    xx = newK[0, 0] * x + newK[0, 1] * y + newK[0, 2]
    yy = newK[1, 0] * x + newK[1, 1] * y + newK[1, 2]
    ww = 1.0 / (newK[2, 0] * x + newK[2, 1] * y + newK[2, 2])
    out = np.stack([xx * ww, yy * ww], axis=-1)
    return out.astype(np.float32).astype(np.float64)


def project_points(Q, rvec_or_R, T, K, dist):
    """cv2.projectPoints for one or more 3D points; returns (..., 2) float64.

    OpenCV ignores the skew term K[0,1] here (u = fx*xd + cx), reproduced.
    """
    Q = np.asarray(Q, dtype=np.float64)
    R = np.asarray(rvec_or_R, dtype=np.float64)
    if R.size == 3:
        R = rodrigues(R)
    T = np.asarray(T, dtype=np.float64).reshape(3)
    K = np.asarray(K, dtype=np.float64)
    k = _dist12(dist)
    X = Q @ R.T + T
    z = X[..., 2]
    z = np.where(z == 0, 1.0, z)   # OpenCV: z = z ? 1./z : 1
    x = X[..., 0] / z
    y = X[..., 1] / z
    r2 = x * x + y * y
    r4 = r2 * r2
    r6 = r4 * r2
    a1 = 2 * x * y
    a2 = r2 + 2 * x * x
    a3 = r2 + 2 * y * y
    cdist = 1 + k[0] * r2 + k[1] * r4 + k[4] * r6
    icdist2 = 1.0 / (1 + k[5] * r2 + k[6] * r4 + k[7] * r6)
    xd = x * cdist * icdist2 + k[2] * a1 + k[3] * a2 + k[8] * r2 + k[9] * r4
    yd = y * cdist * icdist2 + k[2] * a3 + k[3] * a1 + k[10] * r2 + k[11] * r4
    return np.stack([xd * K[0, 0] + K[0, 2], yd * K[1, 1] + K[1, 2]], axis=-1)


def get_optimal_new_camera_matrix(K, dist, size, alpha=1.0, new_size=None):
    """cv2.getOptimalNewCameraMatrix(K, dist, size, alpha, new_size)[0] (centerPrincipalPoint=False).

    OpenCV samples a 9x9 grid over the image, undistorts it to normalised coordinates, and
    interpolates between the inscribed (alpha=0) and circumscribed (alpha=1) rectangles.
    The grid spans [0, w-1] x [0, h-1] (OpenCV >= 4.5.4; earlier versions spanned [0, w]).
    """
    w, h = int(size[0]), int(size[1])
    nw, nh = (w, h) if new_size is None else (int(new_size[0]), int(new_size[1]))
    N = 9
    gx = (np.arange(N, dtype=np.float32) * np.float32(w - 1) / np.float32(N - 1)).astype(np.float32)
    gy = (np.arange(N, dtype=np.float32) * np.float32(h - 1) / np.float32(N - 1)).astype(np.float32)
    px, py = np.meshgrid(gx, gy)          # [y][x]
    ux, uy = undistort_normalized(px.astype(np.float64), py.astype(np.float64), K, dist)
    ux = ux.astype(np.float32).astype(np.float64)
    uy = uy.astype(np.float32).astype(np.float64)
    # inscribed rectangle
    iX0 = np.max(ux[:, 0]); iX1 = np.min(ux[:, N - 1])
    iY0 = np.max(uy[0, :]); iY1 = np.min(uy[N - 1, :])
    # circumscribed rectangle
    oX0 = np.min(ux); oX1 = np.max(ux); oY0 = np.min(uy); oY1 = np.max(uy)
    inner = (iX0, iY0, iX1 - iX0, iY1 - iY0)
    outer = (oX0, oY0, oX1 - oX0, oY1 - oY0)
    fx0 = (nw - 1) / inner[2]; fy0 = (nh - 1) / inner[3]
    cx0 = -fx0 * inner[0];     cy0 = -fy0 * inner[1]
    fx1 = (nw - 1) / outer[2]; fy1 = (nh - 1) / outer[3]
    cx1 = -fx1 * outer[0];     cy1 = -fy1 * outer[1]
    M = np.eye(3)
    M[0, 0] = fx0 * (1 - alpha) + fx1 * alpha
    M[1, 1] = fy0 * (1 - alpha) + fy1 * alpha
    M[0, 2] = cx0 * (1 - alpha) + cx1 * alpha
    M[1, 2] = cy0 * (1 - alpha) + cy1 * alpha
    return M
