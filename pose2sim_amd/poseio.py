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


def _trial_paths(root, json_dirs, json_files_names):
    paths, index = [], {}
    for c, d in enumerate(json_dirs):
        for name in json_files_names[c]:
            index[(c, name)] = len(paths)
            paths.append(os.path.join(root, d, name))
    return paths, index


def max_persons_in_trial(batch, paths):
    """max over all files of count_persons_in_json (triangulation.py:77-90, :784); a file json.load cannot
    read, or whose 'people' is not a list, goes through the Python function so that the same exception
    (or len() of whatever 'people' is) comes out."""
    best = 0
    for i in np.flatnonzero(batch.counts < 0):
        best = max(best, count_persons(paths[i]))
    return max(best, int(batch.counts.max(initial=0)))


def max_persons_sharded(root, json_dirs, json_files_names, rank, world):
    """max_persons_in_trial with the files dealt out over the ranks (every world-th file each) and one
    all-reduce(max); an unreadable file raises on every rank."""
    from . import parallel
    from .ingest import JsonBatch
    paths, _ = _trial_paths(root, json_dirs, json_files_names)
    mine = paths[rank::world]
    best, error = 0, None
    try:
        with JsonBatch(mine) as batch:
            best = max_persons_in_trial(batch, mine)
    except Exception as e:                                   # noqa: BLE001 -- re-raised on every rank below
        error = e
    return parallel.agree_max(best, error)


def load_observations(root, json_dirs, maps, f_range, keypoints_ids, nb_persons, json_files_names=None,
                      count_all_persons=False):
    """extract_files_frame_f (triangulation.py:607-653) for every frame of f_range at once, through the native
    parser (csrc/p2s_ingest.cpp): every file is read and parsed once, on host threads.

    Returns [F][nb_persons][C][K][3] -- float32 when every value is float32-representable (RTMLib output is,
    poseEstimation.py:259), else float64 -- with NaN where the reference would append NaN: missing /
    unreadable file, person index beyond the list, keypoint triplet beyond the list.
    count_all_persons: nb_persons is ignored and taken as the maximum people count over ALL files of
    json_files_names (multi-person mode, :784); returned second.
    """
    from .ingest import JsonBatch
    f0, f1 = f_range
    F = max(0, f1 - f0)
    C = len(json_dirs)
    K = len(keypoints_ids)
    if count_all_persons:
        paths, index = _trial_paths(root, json_dirs, json_files_names)
    else:
        paths, index = [], {}
    slot_of = np.full(F * C, -1, dtype=np.int64)            # (frame, camera) -> parsed file
    for c in range(C):
        m = maps[c]
        for fi, f in enumerate(range(f0, f1)):
            name = m.get(f)
            if name is None:
                continue
            i = index.get((c, name))
            if i is None:
                i = len(paths)
                index[(c, name)] = i
                paths.append(os.path.join(root, json_dirs[c], name))
            slot_of[fi * C + c] = i
    with JsonBatch(paths) as batch:
        if count_all_persons:
            nb_persons = max_persons_in_trial(batch, paths)
        file_offsets = np.full(len(paths), -1, dtype=np.int64)
        used = np.flatnonzero(slot_of >= 0)
        fi, c = used // C, used % C
        file_offsets[slot_of[used]] = (fi * nb_persons * C + c) * (K * 3)
        out = np.full((F, nb_persons, C, K, 3), np.nan, dtype=np.float32)
        if out.size and batch.gather_keypoints(keypoints_ids, nb_persons, file_offsets, C * K * 3, out):
            out = np.full((F, nb_persons, C, K, 3), np.nan, dtype=np.float64)
            batch.gather_keypoints(keypoints_ids, nb_persons, file_offsets, C * K * 3, out)
    return (out, nb_persons) if count_all_persons else out


def read_people_batch(paths):
    """read_json (personAssociation.py:260-274) for many files at once.
    -> (n_people [n_files], rows float [sum][Kj3], Kj3): people with >= 3 values in JSON order; a file that
    cannot be read, has no 'people' list, or holds a person without a 'pose_keypoints_2d' list gives none."""
    from .ingest import P2S_JSON_PERSON_NO_LIST, P2S_JSON_PERSON_NOT_NUMERIC, JsonBatch
    with JsonBatch(paths) as batch:
        lengths = batch.person_lengths
        n_files = len(paths)
        file_of = np.repeat(np.arange(n_files, dtype=np.int64), np.maximum(batch.counts, 0))
        person_of = (np.arange(len(file_of), dtype=np.int64) - batch.person_base[file_of]).astype(np.int32)
        broken = np.zeros(n_files, dtype=bool)
        broken[file_of[lengths == P2S_JSON_PERSON_NO_LIST]] = True                 # js[...]['pose_keypoints_2d'] raises
        if (lengths == P2S_JSON_PERSON_NOT_NUMERIC).any():
            bad = paths[int(file_of[np.flatnonzero(lengths == P2S_JSON_PERSON_NOT_NUMERIC)[0]])]
            raise ValueError(f'{bad}: pose_keypoints_2d must hold numbers only')
        keep = (lengths >= 3) & ~broken[file_of]
        n_people = np.bincount(file_of[keep], minlength=n_files).astype(np.int32)
        kept_len = lengths[keep]
        Kj3 = int(kept_len[0]) if kept_len.size else 0
        if kept_len.size and ((kept_len != Kj3).any() or Kj3 % 3):
            i = int(np.flatnonzero(keep)[np.flatnonzero((kept_len != Kj3) | bool(Kj3 % 3))[0]])
            raise ValueError(f'{paths[int(file_of[i])]}: every person must carry the same number of keypoint triplets')
        rows, _ = batch.gather_people(file_of[keep], person_of[keep], Kj3, np.float64)
    return n_people, rows, Kj3


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
