"""OpenPose-format JSON directories <-> packed observation tensors.

Input contract (poseEstimation.py:239-279 writer, triangulation.py:607-653 reader):
``<pose_dir>/<cam>_json/<name>_%06d.json`` holding
``{"version": 1.3, "people": [{"person_id": [-1], "pose_keypoints_2d": [x0, y0, c0, ...]}, ...]}``.
The frame id of a file is the LAST integer in its name (triangulation.py:799), cameras are the
sub-directories whose name contains 'json', ordered by the last number in the directory name
(common.py:568-584).  Anything missing (file, person, keypoint) reads as NaN
(triangulation.py:629-644).

The reference re-scans every file name with a regex for every frame (O(F^2 C)) and re-opens each
JSON once per person; here the frame -> file maps are built once and each file is parsed once.
"""
import fnmatch
import json
import os
import re

import numpy as np


def sort_stringlist_by_last_number(string_list):
    """common.py:568-584: strings with a number first (by their last number), others after."""
    def key(s):
        numbers = re.findall(r'\d+', s)
        return (False, int(numbers[-1])) if numbers else (True, s)
    return sorted(string_list, key=key)


def frame_of(filename):
    """triangulation.py:799: int(re.split(r'(\\d+)', name)[-2]) -- the last run of digits."""
    return int(re.split(r'(\d+)', filename)[-2])


def list_json_dirs(pose_dir):
    """Camera directories of pose_dir: (all sub-directories sorted, those containing 'json')."""
    names = next(os.walk(pose_dir))[1]
    names = sort_stringlist_by_last_number(names)
    return names, [k for k in names if 'json' in k]


def list_json_files(root, json_dirs):
    """[camera][file name], each list sorted by last number (triangulation.py:761-772).
    Raises (like os.listdir) when a camera directory is missing."""
    files = [fnmatch.filter(os.listdir(os.path.join(root, d)), '*.json') for d in json_dirs]
    return [sort_stringlist_by_last_number(f) for f in files]


def frame_file_map(json_files_names):
    """Per camera: frame id -> file name.  The reference keeps every file whose last number equals
    f; with well-formed directories there is exactly one, and the first one is used here."""
    maps = []
    for names in json_files_names:
        m = {}
        for j in names:
            try:
                f = frame_of(j)
            except (ValueError, IndexError):
                continue
            m.setdefault(f, j)
        maps.append(m)
    return maps


def _load(path):
    with open(path, 'r') as fh:
        return json.load(fh)


def count_persons(path):
    """triangulation.py:77-90."""
    return len(_load(path).get('people', []))


def load_observations(root, json_dirs, maps, f_range, keypoints_ids, nb_persons):
    """extract_files_frame_f (triangulation.py:607-653) for every frame of f_range at once.

    Returns float64 [F][nb_persons][C][K][3] with NaN where the reference would append NaN:
    missing / unreadable file, person index beyond the list, keypoint index beyond the list.
    """
    f0, f1 = f_range
    F = max(0, f1 - f0)
    C = len(json_dirs)
    K = len(keypoints_ids)
    ids = np.asarray(keypoints_ids, dtype=np.int64)
    out = np.full((F, nb_persons, C, K, 3), np.nan, dtype=np.float64)
    for c in range(C):
        m = maps[c]
        for fi, f in enumerate(range(f0, f1)):
            name = m.get(f)
            if name is None:
                continue
            try:
                people = _load(os.path.join(root, json_dirs[c], name))['people']
            except Exception:
                continue
            for n in range(min(nb_persons, len(people))):
                try:
                    kp = people[n]['pose_keypoints_2d']
                except Exception:
                    continue
                try:
                    arr = np.asarray(kp, dtype=np.float64)
                except Exception:
                    continue
                L = arr.shape[0] if arr.ndim == 1 else 0
                # keypoint id*3+2 must exist for the whole triplet to be read (per-keypoint try/except)
                okk = ids * 3 + 2 < L
                if okk.all():
                    out[fi, n, c, :, 0] = arr[ids * 3]
                    out[fi, n, c, :, 1] = arr[ids * 3 + 1]
                    out[fi, n, c, :, 2] = arr[ids * 3 + 2]
                else:
                    sel = np.flatnonzero(okk)
                    out[fi, n, c, sel, 0] = arr[ids[sel] * 3]
                    out[fi, n, c, sel, 1] = arr[ids[sel] * 3 + 1]
                    out[fi, n, c, sel, 2] = arr[ids[sel] * 3 + 2]
    return out


def read_people(path):
    """read_json (personAssociation.py:260-274): people with >= 3 values, JSON order; [] on error."""
    try:
        js = _load(path)
        return [p['pose_keypoints_2d'] for p in js['people'] if len(p['pose_keypoints_2d']) >= 3]
    except Exception:
        return []


def write_openpose_json(path, people_kpts):
    """Writer in the layout of poseEstimation.py:239-279 (tests / demos)."""
    people = []
    for kp in people_kpts:
        people.append({'person_id': [-1], 'pose_keypoints_2d': [float(v) for v in np.asarray(kp).ravel()],
                       'face_keypoints_2d': [], 'hand_left_keypoints_2d': [], 'hand_right_keypoints_2d': [],
                       'pose_keypoints_3d': [], 'face_keypoints_3d': [], 'hand_left_keypoints_3d': [],
                       'hand_right_keypoints_3d': []})
    with open(path, 'w') as fh:
        json.dump({'version': 1.3, 'people': people}, fh)
