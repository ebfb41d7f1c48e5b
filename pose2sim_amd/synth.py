"""Synthetic multi-camera 2D-keypoint generator shared by tests and bench.py.

Produces the packed observation tensor the engine consumes, ``xyl`` float32 [F][Pn][C][K][3]
(x px, y px, likelihood; NaN = missing), plus a calibration in the same form
``calib.load_calibration`` returns.  3D -> 2D projection follows the inverse of the path
(reference model: Utilities/reproj_from_trc_calib.py:446-475).  NumPy only, no GPU.
"""
import numpy as np

from . import cvmath


def make_cameras(C, seed=0, distort=False, width=1920, height=1080):
    """Ring of C cameras, radius 4-6 m, height 1-2.5 m, looking at the origin +- 0.3 m."""
    rng = np.random.default_rng(seed)
    cams = {'S': [], 'K': [], 'dist': [], 'R': [], 'R_mat': [], 'T': [], 'optim_K': [], 'inv_K': [],
            'names': []}
    for c in range(C):
        ang = 2 * np.pi * (c + rng.uniform(-0.2, 0.2)) / C
        rad = rng.uniform(4.0, 6.0)
        pos = np.array([rad * np.cos(ang), rad * np.sin(ang), rng.uniform(1.0, 2.5)])
        target = np.array([0.0, 0.0, 1.0]) + rng.uniform(-0.3, 0.3, 3)
        zc = target - pos
        zc /= np.linalg.norm(zc)
        xc = np.cross(zc, np.array([0.0, 0.0, 1.0]))
        xc /= np.linalg.norm(xc)
        yc = np.cross(zc, xc)
        R = np.stack([xc, yc, zc])            # world -> camera
        T = -R @ pos
        f = 1400.0 + rng.uniform(-100, 100)
        K = np.array([[f, 0.0, width / 2 + rng.uniform(-20, 20)],
                      [0.0, f + rng.uniform(-5, 5), height / 2 + rng.uniform(-20, 20)],
                      [0.0, 0.0, 1.0]])
        if distort:
            d = np.array([rng.uniform(-0.1, 0.1), rng.uniform(-0.05, 0.05),
                          rng.uniform(-1e-3, 1e-3), rng.uniform(-1e-3, 1e-3)])
        else:
            d = np.zeros(4)
        rvec = cvmath.rodrigues_inv(R)
        cams['S'].append(np.array([float(width), float(height)]))
        cams['K'].append(K)
        cams['dist'].append(d)
        cams['R'].append(rvec)
        cams['R_mat'].append(cvmath.rodrigues(rvec))
        cams['T'].append(T)
        cams['optim_K'].append(cvmath.get_optimal_new_camera_matrix(K, d, (width, height), 1.0))
        cams['inv_K'].append(np.linalg.inv(K))
        cams['names'].append(f'cam{c + 1:02d}')
    return cams


def projection_matrices(cams, undistort=False):
    """common.py:291-324: P = [K|0] . [[R,T],[0,1]], with optim_K when undistorting."""
    P = []
    for c in range(len(cams['K'])):
        K = cams['optim_K'][c] if undistort else cams['K'][c]
        Kh = np.hstack([K, np.zeros((3, 1))])
        H = np.vstack([np.hstack([cams['R_mat'][c], cams['T'][c].reshape(3, 1)]), [0, 0, 0, 1.0]])
        P.append(Kh @ H)
    return P


def _skeleton_offsets(K, rng):
    """K body-shaped offsets (m) around the root: a vertical spread 0..1.8 m, +-0.3 m lateral."""
    off = np.empty((K, 3))
    off[:, 0] = rng.uniform(-0.3, 0.3, K)
    off[:, 1] = rng.uniform(-0.3, 0.3, K)
    off[:, 2] = rng.uniform(0.0, 1.8, K)
    return off


def make_points3d(F, Pn, K, seed=0):
    """[F][Pn][K][3] world coordinates: root random walk (sigma 2 cm/frame) folded into a 3x3 m
    area + per-joint sinusoidal motion."""
    rng = np.random.default_rng(seed + 1000)
    Q = np.empty((F, Pn, K, 3))
    t = np.arange(F)[:, None]
    for n in range(Pn):
        walk = np.cumsum(rng.normal(0, 0.02, (F, 2)), axis=0) + rng.uniform(-1.0, 1.0, 2)
        walk = np.abs((walk + 1.5) % 6.0 - 3.0) - 1.5          # reflect into [-1.5, 1.5]
        off = _skeleton_offsets(K, rng)
        amp = rng.uniform(0.0, 0.15, (K, 3))
        freq = rng.uniform(0.01, 0.1, (K, 3))
        ph = rng.uniform(0, 2 * np.pi, (K, 3))
        Q[:, n, :, 0] = walk[:, 0:1] + off[None, :, 0] + amp[None, :, 0] * np.sin(freq[None, :, 0] * t + ph[None, :, 0])
        Q[:, n, :, 1] = walk[:, 1:2] + off[None, :, 1] + amp[None, :, 1] * np.sin(freq[None, :, 1] * t + ph[None, :, 1])
        Q[:, n, :, 2] = off[None, :, 2] + amp[None, :, 2] * np.sin(freq[None, :, 2] * t + ph[None, :, 2])
    return Q


def make_observations(Q3d, cams, seed=0, noise_px=1.5, p_lowlik=0.05, p_outlier=0.03,
                      p_missing_cam=0.01, p_lr_swap=0.0, swap_idx=None, distort=False):
    """Project [F][Pn][K][3] points into every camera -> xyl float32 [F][Pn][C][K][3]."""
    rng = np.random.default_rng(seed + 2000)
    F, Pn, K, _ = Q3d.shape
    C = len(cams['K'])
    xyl = np.empty((F, Pn, C, K, 3), dtype=np.float32)
    flat = Q3d.reshape(-1, 3)
    for c in range(C):
        d = cams['dist'][c] if distort else np.zeros(4)
        uv = cvmath.project_points(flat, cams['R_mat'][c], cams['T'][c], cams['K'][c], d)
        uv = uv.reshape(F, Pn, K, 2)
        uv = uv + rng.normal(0, noise_px, uv.shape)
        out = rng.random((F, Pn, K)) < p_outlier
        uv = uv + out[..., None] * rng.normal(0, 60.0, uv.shape)
        lik = rng.uniform(0.3, 1.0, (F, Pn, K))
        low = rng.random((F, Pn, K)) < p_lowlik
        lik = np.where(low, rng.uniform(0.0, 0.3, (F, Pn, K)), lik)
        if p_lr_swap > 0 and swap_idx is not None:
            sw = rng.random((F, Pn)) < p_lr_swap
            uv_sw = uv[:, :, swap_idx, :]
            lik_sw = lik[:, :, swap_idx]
            uv = np.where(sw[..., None, None], uv_sw, uv)
            lik = np.where(sw[..., None], lik_sw, lik)
        xyl[:, :, c, :, 0] = uv[..., 0]
        xyl[:, :, c, :, 1] = uv[..., 1]
        xyl[:, :, c, :, 2] = lik
        miss = rng.random((F, Pn)) < p_missing_cam
        xyl[:, :, c][miss] = np.nan
    return xyl


def make_config(F, C, K, Pn=1, seed=0, undistort=False, lr_swap=False, swap_idx=None, **kw):
    """One synthetic workload: returns dict(xyl, cams, P, Q3d)."""
    cams = make_cameras(C, seed=seed, distort=undistort)
    Q3d = make_points3d(F, Pn, K, seed=seed)
    xyl = make_observations(Q3d, cams, seed=seed, distort=undistort,
                            p_lr_swap=0.02 if lr_swap else 0.0, swap_idx=swap_idx, **kw)
    return {'xyl': xyl, 'cams': cams, 'P': projection_matrices(cams, undistort), 'Q3d': Q3d}
