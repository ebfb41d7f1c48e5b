"""Robust triangulation stage: drop-in for the reference's ``triangulate_all(config_dict)``
(triangulation.py:656-959), with the per-(frame x person x keypoint) work on the MI355X.

The frame / person / keypoint loops (triangulation.py:796-845) become ONE call into the HIP
engine over the packed tensor of every frame; what stays on the host is what carries state
from frame to frame: person tracking (:847-865), interpolation (:889-894), valid-section
trimming (:897-919), gap filling (:922-926), the .trc writer (:929) and the recap log (:959).
There is no CPU fallback for the kernel work.
"""
import logging
import os

import numpy as np
import pandas as pd

from . import calib as calib_mod
from . import poseio, postproc, skeletons, trc
from . import parallel


def _make_engine():
    """The compute backend: one HIP engine on this rank's GPU (LOCAL_RANK, default 0)."""
    from .engine import Engine
    return Engine(int(os.environ.get('LOCAL_RANK', '0')))


def _session_dir(project_dir):
    """triangulation.py:680-682."""
    session_dir = os.path.realpath(os.path.join(project_dir, '..'))
    return session_dir if 'Config.toml' in os.listdir(session_dir) else os.getcwd()


def select_pose_inputs(project_dir):
    """triangulation.py:751-772: camera folders come from <project>/pose, the files from
    pose-associated, else pose-sync, else pose.  Returns (pose_dir_used, json_dirs, json_files)."""
    pose_dir = os.path.join(project_dir, 'pose')
    poseSync_dir = os.path.join(project_dir, 'pose-sync')
    poseTracked_dir = os.path.join(project_dir, 'pose-associated')
    try:
        pose_listdirs_names = next(os.walk(pose_dir))[1]
        os.listdir(os.path.join(pose_dir, pose_listdirs_names[0]))[0]
    except Exception:
        raise ValueError(f'No json files found in {pose_dir} subdirectories. Make sure you run Pose2Sim.poseEstimation() first.')
    pose_listdirs_names = poseio.sort_stringlist_by_last_number(pose_listdirs_names)
    json_dirs_names = [k for k in pose_listdirs_names if 'json' in k]
    try:
        json_files_names = poseio.list_json_files(poseTracked_dir, json_dirs_names)
        used = poseTracked_dir
    except Exception:
        try:
            json_files_names = poseio.list_json_files(poseSync_dir, json_dirs_names)
            used = poseSync_dir
        except Exception:
            try:
                json_files_names = poseio.list_json_files(pose_dir, json_dirs_names)
                used = pose_dir
            except Exception:
                raise Exception(f'No json files found in {pose_dir}, {poseSync_dir}, nor {poseTracked_dir} subdirectories. Make sure you run Pose2Sim.poseEstimation() first.')
    return used, json_dirs_names, json_files_names


def track_persons(Q, err, nex, ids_mask, f_range, multi_person, max_distance_m, n_cams):
    """triangulation.py:823-874 on the kernel's outputs, frame by frame.

    Q [F][P][K][3], err [F][P][K], nex [F][P][K], ids_mask [F][P][K] (bit c = camera c excluded).
    Returns the per-frame, per-person rows the reference appends to Q_tot / error_tot /
    nb_cams_excluded_tot / id_excluded_cams_tot (persons 0..P-1 only, :870-874).
    """
    F, P, K, _ = Q.shape
    if not multi_person:
        return Q, err.astype(np.float64), nex.astype(np.float64), ids_mask
    allmask = np.uint32((1 << n_cams) - 1) if n_cams < 32 else np.uint32(0xFFFFFFFF)
    Q_out = np.empty((F, P, K, 3))
    e_out = np.empty((F, P, K))
    n_out = np.empty((F, P, K))
    m_out = np.empty((F, P, K), dtype=np.uint32)
    Q_cur = np.full((P, K, 3), np.nan)
    Q_old = np.full((P, K, 3), np.nan)
    for fi, f in enumerate(range(*f_range)):
        nan_mask = np.isnan(Q_cur)
        Q_old = np.where(nan_mask, Q_old, Q_cur)                      # :824-825
        Q_cur = Q[fi]
        e_f, n_f, m_f = err[fi].astype(np.float64), nex[fi].astype(np.float64), ids_mask[fi]
        if f != 0:                                                    # :850 absolute frame number (quirk Q7)
            Q_old, Q_cur, sorted_ids = postproc.sort_people_sports2d(Q_old, np.array(Q_cur), max_dist=max_distance_m)
            e_s = np.full((P, K), np.nan)
            n_s = np.full((P, K), float(n_cams))
            m_s = np.full((P, K), allmask, dtype=np.uint32)
            for n in range(P):                                        # :855-864
                d = int(sorted_ids[n])
                if d >= 0:
                    e_s[n], n_s[n], m_s[n] = e_f[d], n_f[d], m_f[d]
            e_f, n_f, m_f = e_s, n_s, m_s
        Q_out[fi] = Q_cur[:P]
        e_out[fi], n_out[fi], m_out[fi] = e_f, n_f, m_f
    return Q_out, e_out, n_out, m_out


def _cam_exclusion_fractions(mask_rows, n_cams, K):
    """triangulation.py:933-943 from the excluded-camera bit masks of the kept frames."""
    frame_count = mask_rows.shape[0]
    total = frame_count * K
    counts = {}
    flat = mask_rows.reshape(-1).astype(np.uint64)
    for c in range(n_cams):
        counts[c] = int(np.count_nonzero((flat >> np.uint64(c)) & np.uint64(1))) / total
    return counts


def triangulate_all(config_dict):
    """Same contract as the reference: reads <session>/calib*/*.toml and the pose JSON folders of
    config_dict['project']['project_dir'], writes <project>/pose-3d/<name>[_P<n>]_<f0>-<f1>.trc."""
    project_dir = config_dict.get('project').get('project_dir')
    session_dir = _session_dir(project_dir)
    multi_person = config_dict.get('project').get('multi_person')
    pose_model = config_dict.get('pose').get('pose_model')
    frame_range = config_dict.get('project').get('frame_range')
    tcfg = config_dict.get('triangulation')
    likelihood_threshold = tcfg.get('likelihood_threshold_triangulation')
    error_threshold = tcfg.get('reproj_error_threshold_triangulation')
    min_cameras = tcfg.get('min_cameras_for_triangulation')
    interpolation_kind = tcfg.get('interpolation')
    interp_gap_smaller_than = tcfg.get('interp_if_gap_smaller_than')
    max_distance_m = tcfg.get('max_distance_m', None)
    remove_incomplete_frames = tcfg.get('remove_incomplete_frames', False)
    sections_to_keep = tcfg.get('sections_to_keep')
    min_chunk_size = tcfg.get('min_chunk_size', 10)
    fill_large_gaps_with = tcfg.get('fill_large_gaps_with')
    show_interp_indices = tcfg.get('show_interp_indices')
    undistort_points = tcfg.get('undistort_points')
    handle_LR_swap = tcfg.get('handle_LR_swap')
    make_c3d = tcfg.get('make_c3d')

    calib_file = calib_mod.find_calibration_file(session_dir)
    P = calib_mod.computeP(calib_file, undistort=undistort_points)
    calib_params = calib_mod.retrieve_calib_params(calib_file)

    keypoints_ids, keypoints_names, keypoints_idx_swapped = skeletons.keypoints(pose_model, config_dict)
    keypoints_nb = len(keypoints_ids)
    if keypoints_idx_swapped == list(range(keypoints_nb)) and not skeletons.swap_indices(keypoints_names)[1]:
        logging.warning('No left/right swap was performed.')

    pose_dir, json_dirs_names, json_files_names = select_pose_inputs(project_dir)
    n_cams = len(json_dirs_names)

    f_range = [[0, min([len(j) for j in json_files_names])] if frame_range in ('all', 'auto', []) else frame_range][0]
    if n_cams != len(P):
        raise Exception(f'Error: The number of cameras is not consistent: Found {len(P)} cameras in the calibration file, and {n_cams} cameras based on the number of pose folders.')

    # ---- JSON -> packed tensor (native parser) -> HIP engine.  One process: every frame at once.  Several
    # ranks: each one reads, parses and triangulates only its contiguous block of frames (the ingest is the
    # longest stage by far), then ONE all-gather reassembles the trajectory on every rank.
    maps = poseio.frame_file_map(json_files_names)
    rank, world = parallel.dist_info()
    n_frames = max(0, f_range[1] - f_range[0])
    lo, hi = parallel.shard_bounds(n_frames, rank, world)
    my_range = (f_range[0] + lo, f_range[0] + hi)
    if multi_person and world == 1:
        xyl, nb_persons = poseio.load_observations(pose_dir, json_dirs_names, maps, my_range, keypoints_ids, 0,
                                                   json_files_names=json_files_names, count_all_persons=True)
    else:
        nb_persons = 1
        if multi_person:                                   # the person count is a maximum over ALL files (:784)
            nb_persons = poseio.max_persons_sharded(pose_dir, json_dirs_names, json_files_names, rank, world)
    # this rank's stage; whatever it raises is agreed with the other ranks BEFORE the collective, so that a rank that
    # cannot read its files or has no memory left does not leave the others waiting in the all-gather
    local, failure = None, None
    try:
        if not (multi_person and world == 1):
            xyl = poseio.load_observations(pose_dir, json_dirs_names, maps, my_range, keypoints_ids, nb_persons)
        engine = _make_engine()
        engine.set_calibration(P, calib_params if undistort_points else None)
        prm = engine.tri_params(error_threshold, likelihood_threshold, min_cameras, undistort_points, handle_LR_swap)
        swap_arg = keypoints_idx_swapped if handle_LR_swap else None
        if world > 1 and hasattr(engine, 'triangulate_packed') and parallel.collective_device().type == 'cuda':
            # results stay on the GPU: the all-gather takes the packed device buffer as it is
            local = engine.triangulate_packed(xyl, prm, swap_arg, pad_blocks=parallel.largest_shard(n_frames, world) * nb_persons)
        else:
            local = engine.triangulate(xyl, prm, swap_arg)
        capped = engine.tri_stats(reset=True)['capped_units'] if hasattr(engine, 'tri_stats') else 0
        if capped:
            logging.warning(f'{capped} keypoint triangulations were stopped before a camera-subset level of more than 2^26 '
                            'subsets (the reference would have gone on for hours): they are reported as not triangulated. '
                            'Raise min_cameras_for_triangulation to bound the search.')
    except Exception as exc:                               # noqa: BLE001 -- re-raised on every rank by agree_ok
        if world == 1:
            raise
        failure = exc
    parallel.agree_ok(failure)
    gathered = parallel.gather_results(local, n_frames, nb_persons, keypoints_nb, host_copy_on=0 if world > 1 else None)
    if gathered is None:
        return []                                          # tracking, interpolation and the .trc files are rank 0's
    Qk, ek, nk, mk = gathered

    Q_rows, e_rows, n_rows, m_rows = track_persons(Qk, ek, nk, mk, f_range, multi_person, max_distance_m, n_cams)
    index = range(*f_range)
    Q_tot = [pd.DataFrame(Q_rows[:, n].reshape(len(index), keypoints_nb * 3), index=index) for n in range(nb_persons)]
    error_tot = [pd.DataFrame(e_rows[:, n].astype(np.float64), index=index) for n in range(nb_persons)]
    nb_cams_excluded_tot = [pd.DataFrame(n_rows[:, n].astype(np.float64), index=index) for n in range(nb_persons)]

    cam_excluded_count, interp_frames, non_interp_frames, f_range_trimmed, trc_paths = [], [], [], [], []
    for n in range(nb_persons):
        if interpolation_kind != 'none':                                      # :889-894
            try:
                Q_tot[n] = Q_tot[n].apply(postproc.interpolate_zeros_nans, axis=0, args=[interp_gap_smaller_than, interpolation_kind])
            except Exception:
                logging.warning(f'Interpolation was not possible for person {n}. This means that not enough points are available, which is often due to a bad calibration.')

        error_tot[n]['mean'] = error_tot[n].mean(axis=1, skipna=not remove_incomplete_frames)   # :897
        nb_cams_excluded_tot[n]['mean'] = nb_cams_excluded_tot[n].mean(axis=1)
        start, end = postproc.indices_of_first_last_non_nan_chunks(error_tot[n]['mean'], min_chunk_size=min_chunk_size,
                                                                   chunk_choice_method=sections_to_keep)
        f_range_trimmed.append([start, end])

        if end - start <= min_chunk_size:                                     # :903-910
            nb_cams_excluded_tot[n] = pd.DataFrame(columns=nb_cams_excluded_tot[n].columns)
            cam_excluded_count.append({})
            interp_frames.append([])
            non_interp_frames.append([])
            trc_paths.append('')
            logging.info(f'\nPerson {n}: Less than {min_chunk_size} valid frames in a row. Deleting person.')
            continue

        Q_tot[n] = Q_tot[n].iloc[start:end]                                  # :913-916
        error_tot[n] = error_tot[n].iloc[start:end]
        nb_cams_excluded_tot[n] = nb_cams_excluded_tot[n].iloc[start:end]
        masks_kept = m_rows[start:end, n]
        zero_nan_frames = np.where(Q_tot[n].iloc[:, ::3].T.eq(0) | ~np.isfinite(Q_tot[n].iloc[:, ::3].T))
        zero_nan_frames_per_kpt = [zero_nan_frames[1][np.where(zero_nan_frames[0] == k)[0]] for k in range(keypoints_nb)]
        zero_nan_frames_per_kpt = [z[(start < z) & (end > z)] for z in zero_nan_frames_per_kpt]

        if fill_large_gaps_with == 'last_value':                              # :922-926
            Q_tot[n] = Q_tot[n].ffill(axis=0).bfill(axis=0)
            Q_tot[n] = Q_tot[n].replace([np.nan, np.inf], 0)
        elif fill_large_gaps_with == 'zeros':
            Q_tot[n] = Q_tot[n].replace([np.nan, np.inf], 0)

        if rank == 0:
            trc_paths.append(trc.make_trc(config_dict, Q_tot[n], keypoints_names, id_person=n))   # :929
            if make_c3d:
                logging.warning('make_c3d: the c3d package is not available in this build; only the .trc file was written.')
        else:
            trc_paths.append('')

        cam_excluded_count.append(_cam_exclusion_fractions(masks_kept, n_cams, keypoints_nb))   # :933-943

        if show_interp_indices:                                               # :946-953
            gaps = [np.where(np.diff(zero_nan_frames_per_kpt[k]) > 1)[0] + 1 for k in range(keypoints_nb)]
            sequences = [np.split(zero_nan_frames_per_kpt[k], gaps[k]) for k in range(keypoints_nb)]
            interp_frames.append([[f'{seq[0]}:{seq[-1]}' for seq in seq_kpt if len(seq) <= interp_gap_smaller_than and len(seq) > 0] for seq_kpt in sequences])
            non_interp_frames.append([[f'{seq[0]}:{seq[-1]}' for seq in seq_kpt if len(seq) > interp_gap_smaller_than] for seq_kpt in sequences])
        else:
            interp_frames.append(None)
            non_interp_frames.append([])

    if np.all(np.diff(np.array(f_range_trimmed)) == 0):                       # :955-956
        raise Exception('No persons have been triangulated. Please check your calibration and your synchronization, or the triangulation parameters in Config.toml.')

    if rank == 0:
        recap_triangulate(config_dict, error_tot, nb_cams_excluded_tot, keypoints_names, cam_excluded_count,
                          interp_frames, non_interp_frames, f_range_trimmed, f_range, trc_paths, calib_file)
    return trc_paths


# ---- the stage's report (triangulation.py:255-360): message templates as data, one pass over a table of values -------
_MSG = {
    'participant': '\n\nPARTICIPANT {n}\n',
    'keypoint': 'Mean reprojection error for {name} is {px} px (~ {m} m), reached with {cams} excluded cameras. ',
    'none_needed': '  No frames needed to be interpolated.',
    'interpolated': '  Frames {spans} were interpolated.',
    'not_interpolated': '  Frames {spans} were not interpolated.',
    'interp_off': "  No frames were interpolated because 'interpolation_kind' was set to none. ",
    'overall': '\n--> Mean reprojection error for all points on frames {first} to {last} is {px} px, which roughly corresponds to {mm} mm. ',
    'thresholds': 'Cameras were excluded if likelihood was below {lik} and if the reprojection error was above {thr} px.',
    'gaps': 'Gaps were interpolated with {kind} method if smaller than {gap} frames. Larger gaps were filled with {filler}.',
    'excluded': 'In average, {cams} cameras had to be excluded to reach these thresholds.',
    'trimmed': '\nSome frames could not be correctly triangulated: trial trimmed between frames {span}.\n'
               'You might need to tweak the triangulation parameters in Config.toml (for example, try increasing "reproj_error_threshold_triangulation").',
    'stored': '3D coordinates are stored at {path}.',
    'c3d': 'All trc files have been converted to c3d.',
    'swap': 'Limb swapping was {state}.',
    'distortion': 'Lens distortions were {state}.',
}
_FILLERS = {'last_value': 'the last valid value', 'zeros': 'zeros'}


def _spans(ranges):
    """['3:7', '12:12'] -> '3 to 7, 12 to 12' (the reference prints the list and strips its punctuation)."""
    return ', '.join(r.replace(':', ' to ') for r in ranges)


def _camera_sentence(fractions):
    """'Camera A was excluded 13% of the time, Camera B: 12%, and Camera C: 9%.' -- cameras by falling share."""
    ranked = sorted(fractions.items(), key=lambda kv: kv[1])[::-1]
    parts = []
    for i, (cam, frac) in enumerate(ranked):
        pct = int(np.round(frac * 100))
        if i == 0:
            parts.append(f'Camera {cam} was excluded {pct}% of the time, ')
        elif i == len(ranked) - 1:
            parts.append(f'and Camera {cam}: {pct}%.')
        else:
            parts.append(f'Camera {cam}: {pct}%, ')
    return ''.join(parts)


def recap_triangulate(config_dict, error, nb_cams_excluded, keypoints_names, cam_excluded_count, interp_frames,
                      non_interp_frames, f_range_trimmed, f_range, trc_paths, calib_file):
    """The report of triangulation.py:255-360: per keypoint the mean reprojection error (px, and metres through the first
    camera's focal length and distance) and mean number of excluded cameras, the interpolated frame spans, the
    trial-wide means, the thresholds in force, each camera's share of exclusions, where the file went."""
    tcfg = config_dict.get('triangulation')
    calib = calib_mod.load_toml(calib_file)
    cal_keys = calib_mod.camera_keys(calib)
    cam_names = [calib[c].get('name') if calib[c].get('name') else c for c in cal_keys]
    first_cam = calib[cal_keys[0]]
    px_to_m = float(np.sqrt(np.sum(np.array(first_cam['translation'], dtype=np.float64) ** 2))) / first_cam['matrix'][0][0]
    kind = tcfg.get('interpolation')
    min_chunk = tcfg.get('min_chunk_size', 10)
    say = logging.info

    say('')
    persons = [n for n in range(len(error)) if f_range_trimmed[n][1] - f_range_trimmed[n][0] > min_chunk]
    for n in persons:
        first, last = f_range_trimmed[n]
        if len(error) > 1:
            say(_MSG['participant'].format(n=n))
        for idx, name in enumerate(keypoints_names):
            px = np.around(error[n].iloc[:, idx].mean(), decimals=1)
            say(_MSG['keypoint'].format(name=name, px=px, m=np.around(px * px_to_m, decimals=3),
                                        cams=np.around(nb_cams_excluded[n].iloc[:, idx].mean(), decimals=2)))
            if not tcfg.get('show_interp_indices'):
                continue
            if kind == 'none':
                say(_MSG['interp_off'])
                continue
            done, left = list(interp_frames[n][idx]), list(non_interp_frames[n][idx])
            if not done and not left:
                say(_MSG['none_needed'])
            if done:
                say(_MSG['interpolated'].format(spans=_spans(done)))
            if left:
                say(_MSG['not_interpolated'].format(spans=_spans(left)))
        px = np.around(error[n]['mean'].mean(), decimals=1)
        say(_MSG['overall'].format(first=first, last=last, px=px, mm=np.around(px * px_to_m * 1000, decimals=1)))
        say(_MSG['thresholds'].format(lik=tcfg.get('likelihood_threshold_triangulation'), thr=tcfg.get('reproj_error_threshold_triangulation')))
        if kind != 'none':
            say(_MSG['gaps'].format(kind=kind, gap=tcfg.get('interp_if_gap_smaller_than'),
                                    filler=_FILLERS.get(tcfg.get('fill_large_gaps_with'), 'NaNs')))
        say(_MSG['excluded'].format(cams=np.around(nb_cams_excluded[n]['mean'].mean(), decimals=2)))
        if len(range(first, last)) < len(range(*f_range)):
            logging.warning(_MSG['trimmed'].format(span=f_range_trimmed[n]))
        say(_camera_sentence({cam_names[i]: v for i, v in cam_excluded_count[n].items()}))
        say(_MSG['stored'].format(path=trc_paths[n]))

    say('\n\n')
    if tcfg.get('make_c3d'):
        say(_MSG['c3d'])
    say(_MSG['swap'].format(state='handled' if tcfg.get('handle_LR_swap') else 'not handled'))
    say(_MSG['distortion'].format(state='taken into account' if tcfg.get('undistort_points') else 'not taken into account'))
