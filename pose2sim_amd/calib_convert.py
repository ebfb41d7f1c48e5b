"""Qualisys .qca.txt -> calibration TOML, the converter that feeds the path from the shipped demo assets
(SURVEY.md section 8f rank 3).  Mirrors Pose2Sim/Utilities/calib_qca_to_toml.py:59-232 (the 4-coefficient
converter; the in-tree calibration.py one builds 4 distortion terms and then indexes a fifth, :149 vs :1527):
video cameras only, natural order of their serials, intrinsics / 64 / binning, object-centred -> camera-centred
pose, a rotation of pi about the camera x axis, Rodrigues vector, and the same TOML text layout.

Reproduced quirk: read_qca already converts the translation from mm to m (:121-123) and
calib_qca_to_toml_func divides by 1000 again (:233), so the translations in the file are in km; the output
is kept identical to the reference's rather than corrected.  No lxml / OpenCV needed.
"""
import re
import xml.etree.ElementTree as ET

import numpy as np

from . import cvmath


def natural_sort_key(s):
    return [int(c) if c.isdigit() else c.lower() for c in re.split(r'(\d+)', s)]


def read_qca(qca_path, binning_factor=1):
    """-> (C names, S sizes, D distortions[4], K 3x3, R 3x3 (by line), T [m]) of the video cameras."""
    root = ET.parse(qca_path).getroot()
    C, S, D, K, R, T, vid_id = [], [], [], [], [], [], []
    for i, tag in enumerate(root.findall('cameras/camera')):
        C.append(tag.attrib.get('serial'))
        if tag.attrib.get('model') in ('Miqus Video', 'Miqus Video UnderWater', 'none'):
            vid_id.append(i)
    fov = root.findall('cameras/camera/fov_video')
    for tag in fov:
        w = (float(tag.attrib.get('right')) - float(tag.attrib.get('left'))) / binning_factor
        h = (float(tag.attrib.get('bottom')) - float(tag.attrib.get('top'))) / binning_factor
        S.append([w, h])
    for i, tag in enumerate(root.findall('cameras/camera/intrinsic')):
        k1 = float(tag.get('radialDistortion1')) / 64 / binning_factor
        k2 = float(tag.get('radialDistortion2')) / 64 / binning_factor
        p1 = float(tag.get('tangentalDistortion1')) / 64 / binning_factor
        p2 = float(tag.get('tangentalDistortion2')) / 64 / binning_factor
        D.append(np.array([k1, k2, p1, p2]))
        fu = float(tag.get('focalLengthU')) / 64 / binning_factor
        fv = float(tag.get('focalLengthV')) / 64 / binning_factor
        cu = float(tag.get('centerPointU')) / 64 / binning_factor - float(fov[i].attrib.get('left'))
        cv = float(tag.get('centerPointV')) / 64 / binning_factor - float(fov[i].attrib.get('top'))
        K.append(np.array([fu, 0., cu, 0., fv, cv, 0., 0., 1.]).reshape(3, 3))
    for tag in root.findall('cameras/camera/transform'):
        t = [float(tag.get(k)) / 1000 for k in ('x', 'y', 'z')]
        r = {k: float(tag.get(k)) for k in ('r11', 'r12', 'r13', 'r21', 'r22', 'r23', 'r31', 'r32', 'r33')}
        R.append(np.array([r['r11'], r['r21'], r['r31'], r['r12'], r['r22'], r['r32'], r['r13'], r['r23'], r['r33']]).reshape(3, 3))
        T.append(np.array(t))
    C_vid = [C[v] for v in vid_id]
    order = [vid_id[C_vid.index(c)] for c in sorted(C_vid, key=natural_sort_key)]
    pick = lambda L: [L[c] for c in order]          # noqa: E731
    return pick(C), pick(S), pick(D), pick(K), pick(R), pick(T)


def world_to_camera_persp(r, t):
    """Qc = R Q + T  <->  Q = R^-1 Qc - R^-1 T."""
    r = r.T
    return r, -r @ t


def rotate_cam(r, t, ang_x=np.pi, ang_y=0, ang_z=0):
    rt_h = np.block([[r, t.reshape(3, 1)], [np.zeros(3), 1]])
    r_ax_x = np.array([1, 0, 0, 0, np.cos(ang_x), -np.sin(ang_x), 0, np.sin(ang_x), np.cos(ang_x)]).reshape(3, 3)
    r_ax_y = np.array([np.cos(ang_y), 0, np.sin(ang_y), 0, 1, 0, -np.sin(ang_y), 0, np.cos(ang_y)]).reshape(3, 3)
    r_ax_z = np.array([np.cos(ang_z), -np.sin(ang_z), 0, np.sin(ang_z), np.cos(ang_z), 0, 0, 0, 1]).reshape(3, 3)
    r_ax_h = np.block([[r_ax_z @ r_ax_y @ r_ax_x, np.zeros(3).reshape(3, 1)], [np.zeros(3), 1]])
    m = r_ax_h @ rt_h
    return m[:3, :3], m[:3, 3]


def toml_text(C, S, D, K, R, T):
    """The text Utilities/calib_qca_to_toml.py:174-192 writes."""
    out = []
    for c in range(len(C)):
        out.append(f'[cam_{c+1}]\n')
        out.append(f'name = "{C[c]}"\n')
        out.append(f'size = [ {S[c][0]}, {S[c][1]},]\n')
        out.append(f'matrix = [ [ {K[c][0,0]}, 0.0, {K[c][0,2]},], [ 0.0, {K[c][1,1]}, {K[c][1,2]},], [ 0.0, 0.0, 1.0,],]\n')
        out.append(f'distortions = [ {D[c][0]}, {D[c][1]}, {D[c][2]}, {D[c][3]},]\n')
        out.append(f'rotation = [ {R[c][0]}, {R[c][1]}, {R[c][2]},]\n')
        out.append(f'translation = [ {T[c][0]}, {T[c][1]}, {T[c][2]},]\n')
        out.append('fisheye = false\n\n')
    out.append('[metadata]\nadjusted = false\nerror = 0.0\n')
    return ''.join(out)


def calib_qca_to_toml(qca_path, binning_factor=1, toml_path=None):
    """calib_qca_to_toml_func (:195-237).  Returns the path of the written TOML."""
    toml_path = toml_path or qca_path.replace('.qca.txt', '.toml')
    C, S, D, K, R, T = read_qca(qca_path, int(binning_factor))
    RT = [world_to_camera_persp(r, t) for r, t in zip(R, T)]
    RT = [rotate_cam(r, t, ang_x=np.pi, ang_y=0, ang_z=0) for r, t in RT]
    R = [np.array(cvmath.rodrigues_from_matrix(rt[0])).flatten() for rt in RT]
    T = np.array([rt[1] for rt in RT]) / 1000          # second division, as in the reference (:233)
    with open(toml_path, 'w+') as fh:
        fh.write(toml_text(C, S, D, K, R, T))
    return toml_path
