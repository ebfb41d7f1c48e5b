"""Qualisys .qca.txt -> calibration TOML: the converter that feeds the path from the shipped demo assets without lxml or
OpenCV.  Same output text as Pose2Sim/Utilities/calib_qca_to_toml.py:59-237 (the
4-coefficient converter; the in-tree calibration.py one builds 4 distortion terms and then indexes a fifth, :149 vs :1527).

A .qca.txt file lists every camera of the Qualisys system as one <camera> element with its field of view, intrinsics
(in 1/64 pixel) and its pose as seen from the object (mm).  Here each video camera becomes one record; the pose is turned
into OpenCV's camera-centred convention (transpose, then half a turn about the camera's x axis because Qualisys looks
down -z) and the rotation into a Rodrigues vector.

Kept quirk: the reference converts the translation from mm to m when reading (:121-123) and divides by 1000 again when
writing (:233), so the translations in the file are in km; the text is kept identical to the reference's.
"""
import math
import re
import xml.etree.ElementTree as ET

import numpy as np

from . import cvmath

VIDEO_MODELS = ('Miqus Video', 'Miqus Video UnderWater', 'none')
SUBPIXEL = 64                                    # Qualisys stores intrinsics in 1/64 pixel

# half a turn about x as the reference composes it (:152-156): cos(pi) = -1 and sin(pi) = 1.2246e-16, whose digits reach
# the rotation vectors of the file
_HALF_TURN_X = np.array([[1.0, 0.0, 0.0], [0.0, math.cos(math.pi), -math.sin(math.pi)], [0.0, math.sin(math.pi), math.cos(math.pi)]])

_CAMERA_TEXT = ('[cam_{n}]\n'
                'name = "{serial}"\n'
                'size = [ {w}, {h},]\n'
                'matrix = [ [ {fu}, 0.0, {cu},], [ 0.0, {fv}, {cv},], [ 0.0, 0.0, 1.0,],]\n'
                'distortions = [ {k1}, {k2}, {p1}, {p2},]\n'
                'rotation = [ {r0}, {r1}, {r2},]\n'
                'translation = [ {t0}, {t1}, {t2},]\n'
                'fisheye = false\n\n')
_TRAILER = '[metadata]\nadjusted = false\nerror = 0.0\n'


def natural_key(text):
    """'cam10' after 'cam9': digit runs compare as numbers, the rest case-blind."""
    return [int(part) if part.isdigit() else part.lower() for part in re.split(r'(\d+)', text)]


def _attr(element, name, scale=1.0):
    return float(element.get(name)) / scale


def video_cameras(qca_path, binning_factor=1):
    """One record per video camera, in the natural order of the serial numbers: serial, size [w, h], K 3x3, dist [4],
    R 3x3 (object-centred, as stored: the file's r_ij is column i, row j) and t [m]."""
    cameras = []
    for cam in ET.parse(qca_path).getroot().findall('cameras/camera'):
        if cam.get('model') not in VIDEO_MODELS:
            continue
        fov, intr, pose = cam.find('fov_video'), cam.find('intrinsic'), cam.find('transform')
        left, top = float(fov.get('left')), float(fov.get('top'))
        px = lambda name: _attr(intr, name, SUBPIXEL) / binning_factor        # noqa: E731
        K = np.array([[px('focalLengthU'), 0.0, px('centerPointU') - left],
                      [0.0, px('focalLengthV'), px('centerPointV') - top],
                      [0.0, 0.0, 1.0]])
        cameras.append({
            'serial': cam.get('serial'),
            'size': [(float(fov.get('right')) - left) / binning_factor, (float(fov.get('bottom')) - top) / binning_factor],
            'K': K,
            'dist': np.array([px('radialDistortion1'), px('radialDistortion2'), px('tangentalDistortion1'), px('tangentalDistortion2')]),
            'R': np.array([[_attr(pose, f'r{col}{row}') for col in (1, 2, 3)] for row in (1, 2, 3)]),
            't': np.array([_attr(pose, axis, 1000) for axis in 'xyz']),
        })
    return sorted(cameras, key=lambda c: natural_key(c['serial']))


def opencv_pose(R_obj, t_obj):
    """Object-centred (Q = R Qc + t) -> camera-centred (Qc = R' Q + t') with the camera turned to look down +z."""
    R_cam = R_obj.T
    t_cam = -R_cam @ t_obj
    return _HALF_TURN_X @ R_cam, _HALF_TURN_X @ t_cam


def toml_text(cameras):
    parts = []
    for n, cam in enumerate(cameras, start=1):
        R, t = opencv_pose(cam['R'], cam['t'])
        rvec = np.array(cvmath.rodrigues_from_matrix(R)).flatten()
        t = t / 1000                                       # the second division (reference :233)
        K, d = cam['K'], cam['dist']
        parts.append(_CAMERA_TEXT.format(n=n, serial=cam['serial'], w=cam['size'][0], h=cam['size'][1], fu=K[0, 0], cu=K[0, 2],
                                         fv=K[1, 1], cv=K[1, 2], k1=d[0], k2=d[1], p1=d[2], p2=d[3], r0=rvec[0], r1=rvec[1], r2=rvec[2],
                                         t0=t[0], t1=t[1], t2=t[2]))
    return ''.join(parts) + _TRAILER


def calib_qca_to_toml(qca_path, binning_factor=1, toml_path=None):
    """<name>.qca.txt -> <name>.toml (or toml_path).  Returns the path written."""
    toml_path = toml_path or qca_path.replace('.qca.txt', '.toml')
    with open(toml_path, 'w+') as fh:
        fh.write(toml_text(video_cameras(qca_path, int(binning_factor))))
    return toml_path
