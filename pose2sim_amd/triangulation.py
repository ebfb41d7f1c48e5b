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


def track_persons(Q, f_range, multi_person, max_distance_m):
    """triangulation.py:823-874 on the kernel's points, frame by frame: which detection every person slot takes.

    Q [F][P][K][3].  Returns (Q_rows [F][P][K][3], ids [F][P]): the points in tracked order and, per frame and slot, the
    detection it took (-1: none; the reference then records error NaN and every camera excluded, :855-864).  Persons
    0..P-1 only (:870-874)."""
    F, P, K, _ = Q.shape
    ids = np.tile(np.arange(P), (F, 1))
    if not multi_person:
        return Q, ids
    Q_out = np.empty((F, P, K, 3))
    Q_cur = np.full((P, K, 3), np.nan)
    Q_old = np.full((P, K, 3), np.nan)
    for fi, f in enumerate(range(*f_range)):
        nan_mask = np.isnan(Q_cur)
        Q_old = np.where(nan_mask, Q_old, Q_cur)                      # :824-825
        Q_cur = Q[fi]
        if f != 0:                                                    # :850 absolute frame number (quirk Q7)
            Q_old, Q_cur, sorted_ids = postproc.sort_people_sports2d(Q_old, np.array(Q_cur), max_dist=max_distance_m)
            ids[fi] = [int(sorted_ids[n]) for n in range(P)]
        Q_out[fi] = Q_cur[:P]
    return Q_out, ids


def in_tracked_order(table, ids, empty):
    """table [F][P][...] of per-detection values -> the rows of the person slots: slot n of frame f reads detection
    ids[f][n], `empty` where it took none."""
    took = ids >= 0
    rows = np.take_along_axis(table, np.where(took, ids, 0).reshape(ids.shape + (1,) * (table.ndim - 2)), axis=1)
    return np.where(took.reshape(ids.shape + (1,) * (table.ndim - 2)), rows, empty)


class PersonTrial:
    """What the stage keeps of one person after the frame-sequential steps: the kept section [start, end) of the trial
    (positions in f_range), the file written, and the figures the report prints."""

    def __init__(self, start, end):
        self.start, self.end = start, end
        self.kept = False
        self.trc_path = ''
        self.err_per_kpt = self.excl_per_kpt = None            # [K] means over the kept frames
        self.err_mean = self.excl_mean = None                  # means over the kept frames of the per-frame means
        self.cam_fractions = {}
        self.interpolated, self.not_interpolated = [], []


def finish_person(n, coords, err_frame, excl_frame, frames, settings, config_dict, keypoints_names, write):
    """triangulation.py:888-953 for one person, on arrays: coords [F][3 K], err_frame / excl_frame [F] (the frame's mean
    error and mean number of excluded cameras), frames [F] absolute frame numbers.  Interpolates the short gaps, keeps
    the section(s) of frames with a mean error, fills the long gaps and writes the .trc file (rank 0); the report's
    per-keypoint figures are added by table_figures or summed_figures."""
    if settings['interpolation'] != 'none':
        try:
            coords = postproc.interpolate_gaps(coords, frames, settings['max_gap'], settings['interpolation'])
        except Exception:
            logging.warning(f'Interpolation was not possible for person {n}. This means that not enough points are available, which is often due to a bad calibration.')
    start, end = postproc.indices_of_first_last_non_nan_chunks(err_frame, min_chunk_size=settings['min_chunk_size'],
                                                               chunk_choice_method=settings['sections_to_keep'])
    person = PersonTrial(start, end)
    if end - start <= settings['min_chunk_size']:                                                # :903-910
        logging.info(f'\nPerson {n}: Less than {settings["min_chunk_size"]} valid frames in a row. Deleting person.')
        return person
    person.kept = True
    coords = coords[start:end]
    done, left = postproc.gap_spans(coords[:, ::3], start, end, settings['max_gap'])             # before the long gaps are filled
    coords = postproc.fill_gaps(coords, settings['fill_large_gaps_with'])
    if write:
        seq_name = os.path.basename(os.path.realpath(config_dict.get('project').get('project_dir')))
        if config_dict.get('project').get('multi_person'):
            seq_name += f'_P{n}'
        person.trc_path = trc.write_trc(os.path.join(config_dict.get('project').get('project_dir'), 'pose-3d'), seq_name,
                                        frames[start:end], coords, keypoints_names, trc.resolve_frame_rate(config_dict))
        if settings['make_c3d']:
            logging.warning('make_c3d: the c3d package is not available in this build; only the .trc file was written.')
    person.err_mean = postproc.column_means(err_frame[start:end, None])[0]
    person.excl_mean = postproc.column_means(excl_frame[start:end, None])[0]
    if settings['show_interp_indices']:
        person.interpolated, person.not_interpolated = done, left
    else:
        person.interpolated, person.not_interpolated = None, []
    return person


def table_figures(person, err, n_excl, masks, n_cams):
    """The report's per-keypoint means and per-camera shares over the person's kept frames from the whole tables
    (one process): err / n_excl [F][K] float, masks [F][K] u32."""
    if person.kept:
        person.err_per_kpt = postproc.column_means(err[person.start:person.end])
        person.excl_per_kpt = postproc.column_means(n_excl[person.start:person.end])
        person.cam_fractions = postproc.camera_exclusion_fractions(masks[person.start:person.end], n_cams)


def summed_figures(person, sums, n):
    """The same figures from the column sums the ranks reduced (parallel.reduce_report_sums)."""
    if person.kept:
        with np.errstate(invalid='ignore', divide='ignore'):
            person.err_per_kpt = np.where(sums['err_count'][n] > 0, sums['err_sum'][n] / sums['err_count'][n], np.nan)
            person.excl_per_kpt = sums['excl_sum'][n] / sums['frames'][n]
            total = sums['frames'][n] * len(person.err_per_kpt)
            person.cam_fractions = {c: int(v) / total for c, v in enumerate(sums['cam_count'][n])}


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
    frames = np.arange(*f_range)
    settings = {'interpolation': interpolation_kind, 'max_gap': interp_gap_smaller_than, 'remove_incomplete_frames': remove_incomplete_frames,
                'sections_to_keep': sections_to_keep, 'min_chunk_size': min_chunk_size, 'fill_large_gaps_with': fill_large_gaps_with,
                'show_interp_indices': show_interp_indices, 'make_c3d': make_c3d}
    skipna = not remove_incomplete_frames                                      # :897
    allmask = np.uint32((1 << n_cams) - 1) if n_cams < 32 else np.uint32(0xFFFFFFFF)

    def sequential_steps(Qk, err_frames, excl_frames):
        """Tracking, then per person interpolation, kept section, gap filling and the file (rank 0)."""
        Q_rows, ids = track_persons(Qk, f_range, multi_person, max_distance_m)
        e_fr = in_tracked_order(err_frames, ids, np.nan)
        x_fr = in_tracked_order(excl_frames, ids, float(n_cams))
        people = [finish_person(n, Q_rows[:, n].reshape(len(frames), keypoints_nb * 3), e_fr[:, n], x_fr[:, n], frames, settings,
                                config_dict, keypoints_names, write=(rank == 0)) for n in range(nb_persons)]
        if all(p.end == p.start for p in people):                              # :955-956
            raise Exception('No persons have been triangulated. Please check your calibration and your synchronization, or the triangulation parameters in Config.toml.')
        return people, ids

    if world == 1:
        Qk, ek, nk, mk = local
        ek, nk = np.asarray(ek, dtype=np.float64), np.asarray(nk, dtype=np.float64)
        flat = lambda t: t.reshape(-1, t.shape[-1])                            # noqa: E731
        persons, ids = sequential_steps(Qk, postproc.frame_means(flat(ek), skipna=skipna).reshape(ek.shape[:2]),
                                        postproc.frame_means(flat(nk)).reshape(nk.shape[:2]))
        e_rows, n_rows, m_rows = in_tracked_order(ek, ids, np.nan), in_tracked_order(nk, ids, float(n_cams)), in_tracked_order(mk, ids, allmask)
        for n, person in enumerate(persons):
            table_figures(person, e_rows[:, n], n_rows[:, n], m_rows[:, n], n_cams)
    else:
        # the points and the per-frame means travel; the per-unit tables stay where they were computed and only their
        # column sums over the kept frames are exchanged once rank 0 has settled which frames those are
        Qk, row_means, tables = parallel.gather_trajectory(local, n_frames, nb_persons, keypoints_nb, skipna, host_copy_on=0)
        persons, verdict, failure = None, np.zeros(nb_persons * 2 + n_frames * nb_persons, dtype=np.int64), None
        if rank == 0:
            try:
                persons, ids = sequential_steps(Qk, row_means[:, :, 0], row_means[:, :, 1])
                verdict = np.concatenate([np.array([[p.start, p.end] for p in persons]).reshape(-1), ids.reshape(-1)])
            except Exception as exc:                       # noqa: BLE001 -- the other ranks wait in the broadcast below
                failure = exc
        parallel.agree_ok(failure)
        verdict = parallel.broadcast_ints(verdict, nb_persons * 2 + n_frames * nb_persons)
        sums = parallel.reduce_report_sums(tables, verdict[nb_persons * 2:].reshape(n_frames, nb_persons),
                                           verdict[:nb_persons * 2].reshape(nb_persons, 2), n_cams)
        if rank != 0:
            return []                                      # the files and the report are rank 0's
        for n, person in enumerate(persons):
            summed_figures(person, sums, n)
    if rank == 0:
        recap_triangulate(config_dict, persons, keypoints_names, f_range, calib_file)
    return [p.trc_path for p in persons]


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


def recap_triangulate(config_dict, persons, keypoints_names, f_range, calib_file):
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
    say = logging.info

    say('')
    for n, person in enumerate(persons):
        if not person.kept:
            continue
        if len(persons) > 1:
            say(_MSG['participant'].format(n=n))
        for idx, name in enumerate(keypoints_names):
            px = np.around(person.err_per_kpt[idx], decimals=1)
            say(_MSG['keypoint'].format(name=name, px=px, m=np.around(px * px_to_m, decimals=3),
                                        cams=np.around(person.excl_per_kpt[idx], decimals=2)))
            if not tcfg.get('show_interp_indices'):
                continue
            if kind == 'none':
                say(_MSG['interp_off'])
                continue
            done, left = list(person.interpolated[idx]), list(person.not_interpolated[idx])
            if not done and not left:
                say(_MSG['none_needed'])
            if done:
                say(_MSG['interpolated'].format(spans=_spans(done)))
            if left:
                say(_MSG['not_interpolated'].format(spans=_spans(left)))
        px = np.around(person.err_mean, decimals=1)
        say(_MSG['overall'].format(first=person.start, last=person.end, px=px, mm=np.around(px * px_to_m * 1000, decimals=1)))
        say(_MSG['thresholds'].format(lik=tcfg.get('likelihood_threshold_triangulation'), thr=tcfg.get('reproj_error_threshold_triangulation')))
        if kind != 'none':
            say(_MSG['gaps'].format(kind=kind, gap=tcfg.get('interp_if_gap_smaller_than'),
                                    filler=_FILLERS.get(tcfg.get('fill_large_gaps_with'), 'NaNs')))
        say(_MSG['excluded'].format(cams=np.around(person.excl_mean, decimals=2)))
        if len(range(person.start, person.end)) < len(range(*f_range)):
            logging.warning(_MSG['trimmed'].format(span=[person.start, person.end]))
        say(_camera_sentence({cam_names[i]: v for i, v in person.cam_fractions.items()}))
        say(_MSG['stored'].format(path=person.trc_path))

    say('\n\n')
    if tcfg.get('make_c3d'):
        say(_MSG['c3d'])
    say(_MSG['swap'].format(state='handled' if tcfg.get('handle_LR_swap') else 'not handled'))
    say(_MSG['distortion'].format(state='taken into account' if tcfg.get('undistort_points') else 'not taken into account'))
