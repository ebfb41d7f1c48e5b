"""Person association stage: drop-in for the reference's ``associate_all(config_dict)``
(personAssociation.py:642-808), both branches, on the MI355X.

Per frame the reference computes Pluecker rays, the pairwise epipolar affinity, ~20 full SVDs
(matchSVT) and then extracts proposals; here every frame of the trial goes to the HIP engine in ONE
call and only the order-sensitive proposal extraction (person_index_per_cam, :512-549, exact
``np.unique`` / ``argsort`` semantics) and the JSON rewrite (:552-580) stay on the host.

Single-person branch (:67-257): the brute-force search over "one person per camera" x "cameras
switched off" on one tracked keypoint runs in the engine for every frame at once
(``Engine.associate_single``); the host only gathers the tracked keypoint of every detected person and
rewrites the JSON files.  With ``undistort_points`` the reference's single-person code reprojects through
``range(len(Q_comb))`` = the first 4 kept cameras only (triangulate_comb :130-132) and raises IndexError as soon
as a combination keeps fewer than 4 cameras; that accident is not reproduced and the combination is refused
here with a clear message.
"""
import logging
import os

import numpy as np

from . import calib as calib_mod
from . import poseio, skeletons
from ._lib import P2S_MAX_COMBINATIONS, P2S_MAX_PERSONS_PER_CAM, P2S_MAX_PERSONS_TOTAL


def _make_engine():
    from .engine import Engine
    return Engine(int(os.environ.get('LOCAL_RANK', '0')))


def person_index_per_cam(affinity, cum_persons_per_view, min_cameras_for_triangulation):
    """personAssociation.py:512-549.  Row order of the result = person order in the rewritten JSON."""
    n_views = len(cum_persons_per_view) - 1
    rows = []
    for r in range(affinity.shape[0]):
        row = []
        for cam in range(n_views):
            block = affinity[r, cum_persons_per_view[cam]:cum_persons_per_view[cam + 1]]
            row += [np.argmax(block) if (len(block) > 0 and max(block) > 0) else -1]
        rows.append(row)
    return proposals_from_rows(np.array(rows, dtype=float), min_cameras_for_triangulation)


def proposals_from_rows(rows, min_cameras_for_triangulation):
    """Second half of person_index_per_cam (:528-549).  Which proposals come first is decided by np.unique and by
    np.argsort of their multiplicities, whose order among equal counts is unspecified (and not stable in NumPy 2.x's
    SIMD sorts): those two calls are made exactly as the reference makes them, so the person order is the reference's.
    What follows them is set logic: a proposal that reuses, for some camera, a person of ANY proposal before it
    (kept or not) is dropped; so is one seen by too few cameras."""
    if rows.size == 0:
        return np.array([])
    distinct, counts = np.unique(rows, axis=0, return_counts=True)
    ranked = distinct[np.argsort(counts)[::-1]]
    ranked[ranked == -1] = np.nan
    reused = np.zeros(ranked.shape, dtype=bool)
    for cam in range(ranked.shape[1]):                      # per camera: every occurrence of a person after its first
        col = ranked[:, cam]
        seen = ~np.isnan(col)
        idx = np.flatnonzero(seen)
        _, first = np.unique(col[idx], return_index=True)
        later = np.ones(len(idx), dtype=bool)
        later[first] = False
        reused[idx[later], cam] = True
    enough = (~np.isnan(ranked)).sum(axis=1) >= min_cameras_for_triangulation
    out = ranked[~reused.any(axis=1) & enough]
    return out if len(out) else np.array([])


def proposals_batch(affinity, n_persons, min_cameras_for_triangulation):
    """person_index_per_cam for every frame (csrc/p2s_proposals.cpp): the per-detection argmax rows (the Python double
    loop of :516-527), their distinct rows and multiplicities (np.unique, :530) and, after the ranking, the first-come and
    minimum-camera filters (:534-546) are native, all frames at once.  The ranking itself -- np.argsort of the
    multiplicities, whose order among equal counts is not specified and not stable in NumPy 2.x's SIMD sorts -- is the
    reference's own call on the array the reference would pass (remembered per distinct array: a trial has a handful).
    affinity [F][n_max][n_max], n_persons [F][C] -> list of float arrays [n_proposals][C] (NaN = unseen)."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    affinity = np.ascontiguousarray(affinity, dtype=np.float64)
    n_persons = np.ascontiguousarray(n_persons, dtype=np.int32)
    F, n_cams = n_persons.shape
    n_max = affinity.shape[1] if affinity.ndim == 3 else 0
    if F == 0 or n_max == 0:
        return [np.array([]) for _ in range(F)]
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)    # noqa: E731
    rows = np.full((F, n_max, n_cams), -1, dtype=np.int32)
    _lib.check(lib.p2s_assoc_argmax_rows(F, n_cams, n_max, ptr(affinity), ptr(n_persons), 0, ptr(rows)))
    totals = np.ascontiguousarray(n_persons.sum(axis=1), dtype=np.int32)
    uniq = np.empty((F, n_max, n_cams), dtype=np.int32)
    counts = np.zeros((F, n_max), dtype=np.int64)
    n_uniq = np.zeros(F, dtype=np.int32)
    _lib.check(lib.p2s_assoc_unique_rows(F, n_cams, n_max, ptr(rows), ptr(totals), 0, ptr(uniq), ptr(counts), ptr(n_uniq)))
    rank = np.zeros((F, n_max), dtype=np.int32)
    remembered = {}
    for f in range(F):
        n = int(n_uniq[f])
        if n == 0:
            continue
        c = counts[f, :n]
        key = c.tobytes()
        order = remembered.get(key)
        if order is None:
            order = remembered[key] = np.argsort(c)[::-1].astype(np.int32)      # :531, the reference's call
        rank[f, :n] = order
    props = np.empty((F, n_max, n_cams), dtype=np.int32)
    n_props = np.zeros(F, dtype=np.int32)
    _lib.check(lib.p2s_assoc_filter_rows(F, n_cams, n_max, ptr(uniq), ptr(n_uniq), ptr(rank), int(min_cameras_for_triangulation), 0,
                                         ptr(props), ptr(n_props)))
    as_float = np.where(props < 0, np.nan, props.astype(np.float64))
    empty = np.array([])
    return [as_float[f, :n_props[f]] if n_props[f] else empty for f in range(F)]


def rewrite_json_files(json_tracked_files_f, json_files_f, proposals, n_cams):
    """personAssociation.py:552-580 for one frame: people reordered by proposal, {} where a camera does not see the
    person, no output file where the source cannot be read -- one frame of the native batch writer below."""
    rewrite_json_files_batch([list(json_tracked_files_f)[:n_cams]], [list(json_files_f)[:n_cams]], [np.asarray(proposals, dtype=float)], n_cams)


def rewrite_json_files_batch(dst_files, src_files, proposals_per_frame, n_cams):
    """rewrite_json_files for every frame of the trial in one native call (csrc/p2s_rewrite.cpp): same text as
    json.dumps, same "remove the output on any error" rule, on host threads.
    dst_files / src_files: [F][n_cams] paths; proposals_per_frame[f]: array [n_proposals][n_cams] with NaN = unseen."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    src, dst, sel, sel_off = [], [], [], [0]
    for f, proposals in enumerate(proposals_per_frame):
        proposals = np.asarray(proposals, dtype=float).reshape(-1, n_cams) if np.asarray(proposals).size else np.zeros((0, n_cams))
        for cam in range(n_cams):
            src.append(os.fsencode(src_files[f][cam]))
            dst.append(os.fsencode(dst_files[f][cam]))
            col = proposals[:, cam]
            sel.extend(np.where(np.isnan(col), -1, col).astype(np.int64).tolist())
            sel_off.append(len(sel))
    n = len(src)
    if n == 0:
        return

    def pack(paths):
        off = np.zeros(len(paths) + 1, dtype=np.int64)
        np.cumsum([len(p) for p in paths], out=off[1:])
        return b''.join(paths), off
    sblob, soff = pack(src)
    dblob, doff = pack(dst)
    sel = np.ascontiguousarray(sel, dtype=np.int32)
    sel_off = np.ascontiguousarray(sel_off, dtype=np.int64)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a.size else None    # noqa: E731
    _lib.check(lib.p2s_json_rewrite_people(sblob, ptr(soff), dblob, ptr(doff), n, ptr(sel_off), ptr(sel), 0, None))


_RECAP = {
    'single': ('\n--> Mean reprojection error for {kpt} point on all frames is {px} px, which roughly corresponds to {mm} mm. ',
               '--> In average, {cams} cameras had to be excluded to reach the demanded {thr} px error threshold after excluding points with likelihood below {lik}.'),
    'multi': ('\n--> A person was reconstructed if the lines from cameras to their keypoints intersected within {recon} m and if the calculated affinity stayed above {aff}.',
              '--> Beware that people were sorted across cameras, but not across frames. This will be done in the triangulation stage.'),
    'stored': '\nTracked json files are stored in {path}.',
}


def recap_tracking(config_dict, error=0, nb_cams_excluded=0):
    """The report of personAssociation.py:583-639: single-person mode -- mean reprojection error of the tracked keypoint
    (px, and mm through the first camera's focal length and distance) and mean number of cameras switched off;
    multi-person mode -- the two thresholds in force; then where the files went."""
    project_dir = config_dict.get('project').get('project_dir')
    pcfg = config_dict.get('personAssociation')
    if config_dict.get('project').get('multi_person'):
        values = dict(recon=pcfg.get('multi_person').get('reconstruction_error_threshold'), aff=pcfg.get('multi_person').get('min_affinity'))
        lines = _RECAP['multi']
    else:
        session_dir = os.path.realpath(os.path.join(project_dir, '..'))
        session_dir = session_dir if 'Config.toml' in os.listdir(session_dir) else os.getcwd()
        calib = calib_mod.load_toml(calib_mod.find_calibration_file(session_dir))
        first_cam = calib[calib_mod.camera_keys(calib)[0]]
        px_to_mm = float(np.sqrt(np.sum(np.array(first_cam['translation'], dtype=np.float64) ** 2))) / first_cam['matrix'][0][0] * 1000
        px = np.around(np.nanmean(error), decimals=1)
        single = pcfg.get('single_person')
        values = dict(kpt=single.get('tracked_keypoint'), px=px, mm=np.around(px * px_to_mm, decimals=1),
                      cams=np.around(np.mean(nb_cams_excluded), decimals=2), thr=single.get('reproj_error_threshold_association'),
                      lik=single.get('likelihood_threshold_association', 0.3))
        lines = _RECAP['single']
    for line in lines:
        logging.info(line.format(**values))
    logging.info(_RECAP['stored'].format(path=os.path.realpath(os.path.join(project_dir, 'pose-associated'))))


def single_person_candidates(paths, k3):
    """Per file: persons_combinations' count (personAssociation.py:84-91: people whose x values are not all
    NaN; 0 when the file cannot be read or a person has no keypoint list) and, for candidate i < count, the
    values [k3:k3+3] of read_json's i-th person (people with >= 3 values, :260-274), NaN when there is none.
    -> (counts int32 [n_files], tracked float64 [sum(counts)][3])."""
    from .ingest import P2S_JSON_PERSON_NO_LIST, P2S_JSON_PERSON_NOT_NUMERIC, JsonBatch
    with JsonBatch(paths) as batch:
        lengths = batch.person_lengths
        n_files = len(paths)
        file_of = np.repeat(np.arange(n_files, dtype=np.int64), np.maximum(batch.counts, 0))
        person_of = (np.arange(len(file_of), dtype=np.int64) - batch.person_base[file_of]).astype(np.int32)
        broken = np.zeros(n_files, dtype=bool)
        broken[file_of[(lengths == P2S_JSON_PERSON_NO_LIST) | (lengths == P2S_JSON_PERSON_NOT_NUMERIC)]] = True
        width = int(max(lengths.max(initial=0), k3 + 3))
        vals, _ = batch.gather_people(file_of, person_of, width, np.float64)
        xs = vals[:, 0::3]
        in_list = np.arange(xs.shape[1])[None, :] * 3 < np.maximum(lengths, 0)[:, None]
        all_nan = (np.isnan(xs) | ~in_list).all(axis=1)                       # all([]) is True: empty lists drop out
        counted = ~all_nan & ~broken[file_of]
        counts = np.bincount(file_of[counted], minlength=n_files).astype(np.int32)
        listed = (lengths >= 3) & ~broken[file_of]                           # read_json's list
        n_listed = np.bincount(file_of[listed], minlength=n_files)
        first = np.zeros(n_files + 1, dtype=np.int64)
        np.cumsum(n_listed, out=first[1:])
        listed_rows = np.flatnonzero(listed)
        cand_file = np.repeat(np.arange(n_files, dtype=np.int64), counts)
        cand_base = np.zeros(n_files + 1, dtype=np.int64)
        np.cumsum(counts, out=cand_base[1:])
        cand_j = np.arange(len(cand_file), dtype=np.int64) - cand_base[cand_file]
        tracked = np.full((len(cand_file), 3), np.nan)
        has = cand_j < n_listed[cand_file]
        src = listed_rows[(first[cand_file] + cand_j)[has]]
        ok = lengths[src] >= k3 + 3
        tracked[np.flatnonzero(has)[ok]] = vals[src[ok], k3:k3 + 3]
    return counts, tracked


def _associate_single_person(config_dict, frames_src, frames_dst, n_cams, P_all, calib_params):
    """Single-person branch of associate_all (:745-755, :772-781) over every frame in one engine call."""
    pose_model = config_dict.get('pose').get('pose_model')
    tracked_keypoint = config_dict.get('personAssociation').get('single_person').get('tracked_keypoint')
    error_threshold_tracking = config_dict.get('personAssociation').get('single_person').get('reproj_error_threshold_association')
    likelihood_threshold = config_dict.get('personAssociation').get('likelihood_threshold_association')
    min_cameras_for_triangulation = config_dict.get('triangulation').get('min_cameras_for_triangulation')
    if config_dict.get('triangulation').get('undistort_points'):
        raise NotImplementedError('single-person association with undistort_points: the reference reprojects through '
                                  'range(len(Q_comb)) = the first 4 kept cameras (personAssociation.py:130-132) and raises '
                                  'IndexError as soon as a combination keeps fewer than 4; that behaviour is not reproduced. '
                                  'Run the association on the distorted points (undistort_points = false).')
    try:
        tracked_keypoint_id = skeletons.node_id_by_name(pose_model, tracked_keypoint, config_dict)
        assert tracked_keypoint_id                       # id None and id 0 both land in the fallback, :749
    except Exception:
        tracked_keypoint_id = 0
        rows = skeletons.model_rows(pose_model, config_dict)
        tracked_keypoint_name = next(r[0] for r in rows if r[1] == 0)
        logging.warning(f'{tracked_keypoint} not found in {pose_model}, consider editing tracked_keypoint in Config.toml. Tracking {tracked_keypoint_name} instead.')
    k3 = tracked_keypoint_id * 3

    # the tracked keypoint of every candidate person: index i of a camera is read_json's i-th person (:200-202),
    # the number of candidates is persons_combinations' own count (:84-91); one native parse of every file
    F = len(frames_src)
    n_persons, tracked = single_person_candidates([p for src in frames_src for p in src], k3)
    n_persons = n_persons.reshape(F, n_cams)
    if n_persons.max(initial=0) > P2S_MAX_PERSONS_PER_CAM:
        raise ValueError(f'a camera holds more than {P2S_MAX_PERSONS_PER_CAM} detections in one frame')
    if np.prod(np.maximum(n_persons, 1).astype(np.float64), axis=1).max(initial=0) > P2S_MAX_COMBINATIONS:
        raise ValueError(f'a frame holds more than {P2S_MAX_COMBINATIONS} person combinations')
    tracked = np.asarray(tracked, dtype=np.float64).reshape(-1, 3)

    engine = _make_engine()
    engine.set_calibration(P_all, calib_params)
    comb, err, _ = engine.associate_single(n_persons, tracked, error_threshold_tracking, likelihood_threshold,
                                           min_cameras_for_triangulation)

    error_min_tot, cameras_off_tot, proposals_all = [], [], []
    for fi in range(F):
        proposal = np.where(comb[fi] < 0, np.nan, comb[fi].astype(float))
        if not np.isinf(err[fi]):
            error_min_tot.append(err[fi])
        cameras_off_tot.append(float(np.count_nonzero(np.isnan(proposal))))
        proposals_all.append([proposal])
    rewrite_json_files_batch(frames_dst, frames_src, proposals_all, n_cams)      # every file of the trial, host threads
    return error_min_tot, cameras_off_tot


def associate_all(config_dict):
    """Same contract as the reference: reads <project>/pose/<cam>_json, writes
    <project>/pose-associated/<cam>_json with the people of every file reordered consistently
    across cameras."""
    project_dir = config_dict.get('project').get('project_dir')
    session_dir = os.path.realpath(os.path.join(project_dir, '..'))
    session_dir = session_dir if 'Config.toml' in os.listdir(session_dir) else os.getcwd()
    multi_person = config_dict.get('project').get('multi_person')
    pose_model = config_dict.get('pose').get('pose_model')
    min_cameras_for_triangulation = config_dict.get('triangulation').get('min_cameras_for_triangulation')
    reconstruction_error_threshold = config_dict.get('personAssociation').get('multi_person').get('reconstruction_error_threshold')
    min_affinity = config_dict.get('personAssociation').get('multi_person').get('min_affinity')
    frame_range = config_dict.get('project').get('frame_range')
    undistort_points = config_dict.get('triangulation').get('undistort_points')

    calib_file = calib_mod.find_calibration_file(session_dir)
    pose_dir = os.path.join(project_dir, 'pose')
    poseSync_dir = os.path.join(project_dir, 'pose-sync')
    poseTracked_dir = os.path.join(project_dir, 'pose-associated')

    P_all = calib_mod.computeP(calib_file, undistort=undistort_points)
    calib_params = calib_mod.retrieve_calib_params(calib_file)
    skeletons.model_rows(pose_model, config_dict)          # NameError for an unknown model, like :695-711

    pose_listdirs_names = next(os.walk(pose_dir))[1]
    try:
        pose_listdirs_names = poseio.sort_stringlist_by_last_number(pose_listdirs_names)
        os.listdir(os.path.join(pose_dir, pose_listdirs_names[0]))[0]
    except Exception:
        raise ValueError(f'No json files found in {pose_dir} subdirectories. Make sure you run Pose2Sim.poseEstimation() first.')
    json_dirs_names = [k for k in pose_listdirs_names if 'json' in k]
    try:
        json_files_names = poseio.list_json_files(poseSync_dir, json_dirs_names)
    except Exception:
        try:
            json_files_names = poseio.list_json_files(pose_dir, json_dirs_names)
        except Exception:
            raise ValueError(f'No json files found in {pose_dir} nor {poseSync_dir} subdirectories. Make sure you run Pose2Sim.poseEstimation() first.')

    os.makedirs(poseTracked_dir, exist_ok=True)               # (several ranks may arrive here together)
    for k in json_dirs_names:
        try:
            os.mkdir(os.path.join(poseTracked_dir, k))
        except Exception:
            pass

    f_range = [[0, max([len(j) for j in json_files_names])] if frame_range in ('all', 'auto', []) else frame_range][0]   # max, :736
    n_cams = len(json_dirs_names)
    if n_cams != len(P_all):
        raise Exception(f'Error: The number of cameras is not consistent:\
                    Found {len(P_all)} cameras in the calibration file,\
                    and {n_cams} cameras based on the number of pose folders.')

    maps = poseio.frame_file_map(json_files_names)
    # Frames are independent and every frame has its own output files: with several ranks (torch.distributed)
    # each one reads, associates and rewrites only its contiguous block, and no result has to be exchanged.
    from . import parallel
    rank, world = parallel.dist_info()
    lo, hi = parallel.shard_bounds(max(0, f_range[1] - f_range[0]), rank, world)
    frames = range(f_range[0] + lo, f_range[0] + hi)

    if not multi_person:
        logging.info('\nSingle-person analysis selected.')
        # (always from pose/: the reference's os.path.exist typo at :764)
        names = [[maps[c].get(f, 'none') for c in range(n_cams)] for f in frames]
        frames_src = [[os.path.join(pose_dir, json_dirs_names[c], nm[c]) for c in range(n_cams)] for nm in names]
        frames_dst = [[os.path.join(poseTracked_dir, json_dirs_names[c], nm[c]) for c in range(n_cams)] for nm in names]
        # whatever this rank's share raises is agreed with the others BEFORE the collective: all ranks raise together
        # instead of one leaving and the rest waiting in the gather
        error_min_tot, cameras_off_tot, failure = [], [], None
        try:
            error_min_tot, cameras_off_tot = _associate_single_person(config_dict, frames_src, frames_dst, n_cams,
                                                                      P_all, calib_params)
        except Exception as exc:                           # noqa: BLE001 -- re-raised on every rank by agree_ok
            if world == 1:
                raise
            failure = exc
        parallel.agree_ok(failure)
        error_min_tot, cameras_off_tot = parallel.gather_lists(error_min_tot, cameras_off_tot)   # recap over all frames
        if rank == 0:
            recap_tracking(config_dict, error_min_tot, cameras_off_tot)
        return
    logging.info('\nMulti-person analysis selected.')

    # ---- read every frame (always from pose/: the reference's os.path.exist typo at :764) --------
    src_files, dst_files = [], []
    for f in frames:
        names = [maps[c].get(f, 'none') for c in range(n_cams)]
        src_files.append([os.path.join(pose_dir, json_dirs_names[c], names[c]) for c in range(n_cams)])
        dst_files.append([os.path.join(poseTracked_dir, json_dirs_names[c], names[c]) for c in range(n_cams)])
    n_people, rows, Kj3 = poseio.read_people_batch([p for src in src_files for p in src])     # one native parse
    n_persons = n_people.reshape(len(frames), n_cams)
    if n_persons.sum(axis=1).max(initial=0) > P2S_MAX_PERSONS_TOTAL:
        raise ValueError(f'a frame holds more than {P2S_MAX_PERSONS_TOTAL} detections over all cameras')
    Kj = (Kj3 or 3) // 3
    kpts = rows.reshape(-1, Kj, 3)

    # ---- rays, affinity and matchSVT of every frame: one call into the HIP engine ------------------
    engine = _make_engine()
    engine.set_calibration(P_all, calib_params)
    prm = engine.assoc_params(reconstruction_error_threshold, min_affinity, min_cameras_for_triangulation)
    affinity = engine.associate(n_persons, kpts, prm)

    proposals_all = proposals_batch(affinity, n_persons, min_cameras_for_triangulation)        # host threads
    rewrite_json_files_batch(dst_files, src_files, proposals_all, n_cams)         # every file of this rank's block, host threads

    if rank == 0:
        recap_tracking(config_dict)
