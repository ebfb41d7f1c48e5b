"""Calibration TOML -> camera parameters and projection matrices.

Restates retrieve_calib_params (common.py:254-288) and computeP (common.py:291-324) on top of
``tomli`` (the reference uses ``toml``) and the camera model in cvmath.py.  Schema written by
calibration.py:1504-1533: one table per camera with name, size, matrix, distortions, rotation
(Rodrigues vector), translation, fisheye; tables named metadata / capture_volume / charuco /
checkerboard are skipped.
"""
import glob
import os

import numpy as np
import tomli

from . import cvmath

_SKIP = ('metadata', 'capture_volume', 'charuco', 'checkerboard')


def load_toml(path):
    with open(path, 'rb') as f:
        return tomli.load(f)


def camera_keys(calib):
    return [c for c in calib.keys() if c not in _SKIP and isinstance(calib[c], dict)]


def retrieve_calib_params(calib_file):
    """common.py:254-288 (+ camera names, used by the recap at triangulation.py:284)."""
    calib = load_toml(calib_file)
    out = {'S': [], 'K': [], 'dist': [], 'inv_K': [], 'optim_K': [], 'R': [], 'R_mat': [], 'T': [], 'names': []}
    for cam in camera_keys(calib):
        S = np.array(calib[cam]['size'], dtype=np.float64)
        K = np.array(calib[cam]['matrix'], dtype=np.float64)
        dist = np.array(calib[cam]['distortions'], dtype=np.float64)
        size = [int(s) for s in S]
        out['S'].append(S)
        out['K'].append(K)
        out['dist'].append(dist)
        out['optim_K'].append(cvmath.get_optimal_new_camera_matrix(K, dist, size, 1, size))
        out['inv_K'].append(np.linalg.inv(K))
        R = np.array(calib[cam]['rotation'], dtype=np.float64)
        out['R'].append(R)
        out['R_mat'].append(cvmath.rodrigues(R))
        out['T'].append(np.array(calib[cam]['translation'], dtype=np.float64))
        out['names'].append(calib[cam].get('name') if calib[cam].get('name') else cam)
    return out


def computeP(calib_file, undistort=False):
    """common.py:291-324: P = [K | 0] . [[R, T], [0, 1]] per camera (optim_K when undistorting)."""
    calib = load_toml(calib_file)
    P = []
    for cam in camera_keys(calib):
        K = np.array(calib[cam]['matrix'], dtype=np.float64)
        if undistort:
            S = np.array(calib[cam]['size'])
            dist = np.array(calib[cam]['distortions'], dtype=np.float64)
            size = [int(s) for s in S]
            K = cvmath.get_optimal_new_camera_matrix(K, dist, size, 1, size)
        Kh = np.block([K, np.zeros(3).reshape(3, 1)])
        R = cvmath.rodrigues(np.array(calib[cam]['rotation'], dtype=np.float64))
        T = np.array(calib[cam]['translation'], dtype=np.float64)
        H = np.block([[R, T.reshape(3, 1)], [np.zeros(3), 1]])
        P.append(Kh @ H)
    return P


def find_calibration_file(session_dir):
    """triangulation.py:698-706 / personAssociation.py:677-685: newest *.toml of the first
    directory whose name contains 'calib'; same exception types and messages."""
    try:
        calib_dir = [os.path.join(session_dir, c) for c in os.listdir(session_dir)
                     if os.path.isdir(os.path.join(session_dir, c)) and 'calib' in c.lower()][0]
    except Exception:
        raise Exception('No .toml calibration direcctory found.')
    try:
        calib_files = glob.glob(os.path.join(calib_dir, '*.toml'))
        calib_file = max(calib_files, key=os.path.getctime)
    except Exception:
        raise Exception(f'No .toml calibration file found in the {calib_dir}.')
    return calib_file


def write_calibration_toml(path, cams):
    """Write a calibration in the schema of calibration.py:1504-1533 (used by tests / demos)."""
    with open(path, 'w') as f:
        for c in range(len(cams['K'])):
            name = cams['names'][c] if 'names' in cams else f'cam_{c + 1:02d}'
            K = np.asarray(cams['K'][c])
            f.write(f'[{name}]\n')
            f.write(f'name = "{name}"\n')
            f.write(f'size = [ {float(cams["S"][c][0])!r}, {float(cams["S"][c][1])!r}]\n')
            rows = ', '.join('[ ' + ', '.join(repr(float(v)) for v in row) + ']' for row in K)
            f.write(f'matrix = [ {rows}]\n')
            f.write('distortions = [ ' + ', '.join(repr(float(v)) for v in np.asarray(cams['dist'][c]).ravel()) + ']\n')
            f.write('rotation = [ ' + ', '.join(repr(float(v)) for v in np.asarray(cams['R'][c]).ravel()) + ']\n')
            f.write('translation = [ ' + ', '.join(repr(float(v)) for v in np.asarray(cams['T'][c]).ravel()) + ']\n')
            f.write('fisheye = false\n\n')
        f.write('[metadata]\nadjusted = false\nerror = 0.0\n')
