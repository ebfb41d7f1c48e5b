"""Shared by make_golden_e2e.py (build container) and tests/test_e2e*.py: materialise a small
synthetic trial on disk (calibration TOML + OpenPose JSON folders) from fixture arrays."""
import os

import numpy as np

from pose2sim_amd import calib as calib_mod
from pose2sim_amd import poseio, synth


def base_config(project_dir, multi_person, **tri):
    cfg = {
        'project': {'project_dir': project_dir, 'multi_person': multi_person, 'frame_rate': 60,
                    'frame_range': 'auto', 'exclude_from_batch': []},
        'pose': {'pose_model': 'Body_with_feet', 'vid_img_extension': 'mp4'},
        'personAssociation': {'likelihood_threshold_association': 0.3,
                              'single_person': {'likelihood_threshold_association': 0.3,
                                                'reproj_error_threshold_association': 20, 'tracked_keypoint': 'Neck'},
                              'multi_person': {'reconstruction_error_threshold': 0.1, 'min_affinity': 0.2}},
        'triangulation': {'reproj_error_threshold_triangulation': 15, 'likelihood_threshold_triangulation': 0.3,
                          'min_cameras_for_triangulation': 2, 'max_distance_m': 1.0,
                          'interp_if_gap_smaller_than': 20, 'interpolation': 'linear',
                          'remove_incomplete_frames': False, 'sections_to_keep': 'all', 'min_chunk_size': 10,
                          'fill_large_gaps_with': 'last_value', 'show_interp_indices': True, 'make_c3d': False,
                          'undistort_points': False, 'handle_LR_swap': False},
        'logging': {'use_custom_logging': True},
    }
    cfg['triangulation'].update(tri)
    return cfg


def cams_from_arrays(z, prefix=''):
    C = len(z[prefix + 'K'])
    return {'S': [z[prefix + 'S'][c] for c in range(C)], 'K': [z[prefix + 'K'][c] for c in range(C)],
            'dist': [z[prefix + 'dist'][c] for c in range(C)], 'R': [z[prefix + 'R'][c] for c in range(C)],
            'T': [z[prefix + 'T'][c] for c in range(C)], 'names': [f'cam_{c + 1:02d}' for c in range(C)]}


def write_trial(root, trial_name, cams, people_per_frame_cam, json_subdir='pose', n_json_kpts=26, model_ids=None,
                calib_text=None):
    """root/Config.toml (session marker), root/calibration/Calib.toml,
    root/<trial>/<json_subdir>/cam_XX_json/cam_XX_%06d.json.

    people_per_frame_cam[f][c] = list of [K_json*3] arrays (possibly empty) or None for a missing file.
    Also creates root/<trial>/pose/cam_XX_json (the reference lists camera folders there).
    """
    os.makedirs(root, exist_ok=True)
    open(os.path.join(root, 'Config.toml'), 'a').close()
    os.makedirs(os.path.join(root, 'calibration'), exist_ok=True)
    if calib_text is not None:                       # a calibration file given as text (cams is ignored)
        with open(os.path.join(root, 'calibration', 'Calib.toml'), 'w') as fh:
            fh.write(calib_text)
        C = len(people_per_frame_cam[0])
    else:
        calib_mod.write_calibration_toml(os.path.join(root, 'calibration', 'Calib.toml'), cams)
        C = len(cams['K'])
    trial = os.path.join(root, trial_name)
    for sub in {'pose', json_subdir}:
        for c in range(C):
            os.makedirs(os.path.join(trial, sub, f'cam_{c + 1:02d}_json'), exist_ok=True)
    for f, per_cam in enumerate(people_per_frame_cam):
        for c in range(C):
            people = per_cam[c]
            if people is None:
                continue
            poseio.write_openpose_json(os.path.join(trial, json_subdir, f'cam_{c + 1:02d}_json', f'cam_{c + 1:02d}_{f:06d}.json'), people)
    # the reference needs at least one file in pose/<some cam> (triangulation.py:753-754, os.walk order)
    for c in range(C):
        d = os.path.join(trial, 'pose', f'cam_{c + 1:02d}_json')
        if not os.listdir(d):
            poseio.write_openpose_json(os.path.join(d, f'cam_{c + 1:02d}_000000.json'), [])
    return trial


def people_from_xyl(xyl, ids, n_json_kpts, drop=None):
    """xyl [F][P][C][K][3] (skeleton order) -> people_per_frame_cam with JSON-order keypoints.
    NaN observations become zeros with zero confidence (what a pose estimator writes); a whole
    (frame, cam) of NaN becomes a missing file."""
    F, P, C, K, _ = xyl.shape
    out = []
    for f in range(F):
        per_cam = []
        for c in range(C):
            if np.isnan(xyl[f, :, c]).all():
                per_cam.append(None)
                continue
            people = []
            for n in range(P):
                kp = np.zeros((n_json_kpts, 3), dtype=np.float64)
                v = xyl[f, n, c].astype(np.float64)
                v = np.where(np.isnan(v), 0.0, v)
                kp[np.asarray(ids)] = v
                people.append(kp.ravel())
            per_cam.append(people)
        out.append(per_cam)
    return out


def make_single_scene(F, C, Kj, seed, n_distract=2):
    """One person of interest seen by every camera plus random distractors (inconsistent across views),
    persons in random order per camera; some low-confidence detections, some empty cameras."""
    rng = np.random.default_rng(seed)
    cams = synth.make_cameras(C, seed=seed)
    Q3d = synth.make_points3d(F, 1 + n_distract * C, Kj, seed=seed)
    xyl = synth.make_observations(Q3d, cams, seed=seed, noise_px=2.0, p_lowlik=0.0, p_outlier=0.0, p_missing_cam=0.0)
    frames = []
    for f in range(F):
        per_cam = []
        for c in range(C):
            if rng.random() < 0.04:
                per_cam.append([])
                continue
            people = [xyl[f, 0, c].astype(np.float64).copy()]
            for d in range(rng.integers(0, n_distract + 1)):
                people.append(xyl[f, 1 + c * n_distract + d, c].astype(np.float64).copy())   # seen by this camera only
            if rng.random() < 0.10:                     # the real person is poorly detected in this view
                people[0][:, 2] = rng.uniform(0.05, 0.25)
            if rng.random() < 0.05:                     # or badly localised
                people[0][:, :2] += rng.normal(0, 80, 2)
            order = rng.permutation(len(people))
            per_cam.append([people[i].ravel() for i in order])
        frames.append(per_cam)
    return cams, frames
