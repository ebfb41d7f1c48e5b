"""Quality metrics of a .trc file: mirror of Pose2Sim/Utilities/trc_evaluate.py:114-340 (bone-length consistency,
trajectory smoothness, missing data, left/right symmetry) with the per-frame quantities and sums computed by the HIP
engine (p2s_trc_metrics_host) -- the parity-independent sanity check of a triangulation run."""
import numpy as np

# (parent, child, display name) of the HALPE_26 segments the reference evaluates (trc_evaluate.py:37-65)
HALPE_26_BONES = tuple(
    [('Hip', f'{s}Hip', f'Hip-{s}Hip') for s in 'R'] + [(f'R{a}', f'R{b}', f'R {n}') for a, b, n in
     (('Hip', 'Knee', 'Thigh'), ('Knee', 'Ankle', 'Shank'), ('Ankle', 'BigToe', 'Foot'), ('BigToe', 'SmallToe', 'Toe'), ('Ankle', 'Heel', 'Heel'))] +
    [('Hip', 'LHip', 'Hip-LHip')] + [(f'L{a}', f'L{b}', f'L {n}') for a, b, n in
     (('Hip', 'Knee', 'Thigh'), ('Knee', 'Ankle', 'Shank'), ('Ankle', 'BigToe', 'Foot'), ('BigToe', 'SmallToe', 'Toe'), ('Ankle', 'Heel', 'Heel'))] +
    [('Hip', 'Neck', 'Trunk'), ('Neck', 'Head', 'Neck-Head')] +
    [(p, c, n) for s in 'RL' for p, c, n in
     (('Neck', f'{s}Shoulder', f'Neck-{s}Shoulder'), (f'{s}Shoulder', f'{s}Elbow', f'{s} Upper Arm'), (f'{s}Elbow', f'{s}Wrist', f'{s} Forearm'))])

# (left segment, right segment, display name) (trc_evaluate.py:68-78)
SYMMETRIC_BONE_PAIRS = (('Hip-LHip', 'Hip-RHip', 'Hip'),) + tuple((f'L {n}', f'R {n}', n) for n in ('Thigh', 'Shank', 'Foot', 'Toe', 'Heel')) + \
    (('Neck-LShoulder', 'Neck-RShoulder', 'Shoulder'),) + tuple((f'L {n}', f'R {n}', n) for n in ('Upper Arm', 'Forearm'))


def load_trc_as_marker_array(trc_path):
    """trc_evaluate.py:83-111 / common.py:178-199 -> (marker_names, time [F], xyz [F][K][3], fps)."""
    with open(trc_path, 'r') as fh:
        lines = fh.readlines()
    marker_names = lines[3].strip().split('\t')[2::3]
    fps = float(lines[2].split('\t')[0])
    data = np.genfromtxt(trc_path, skip_header=5, delimiter='\t')[:, 1:]
    if data.ndim == 1:
        data = data.reshape(1, -1)
    K = len(marker_names)
    return marker_names, data[:, 0], np.ascontiguousarray(data[:, 1:1 + 3 * K].reshape(len(data), K, 3)), fps


def evaluate_arrays(xyz, marker_names, fps, engine, bones=None):
    """The four metric tables of evaluate_single for xyz [F][K][3]."""
    bones = HALPE_26_BONES if bones is None else bones
    index = {n: i for i, n in enumerate(marker_names)}
    present = [(p, c, n) for p, c, n in bones if p in index and c in index]
    pairs = np.array([[index[p], index[c]] for p, c, _ in present], dtype=np.int32).reshape(-1, 2)
    F = xyz.shape[0]
    bone_len, bone_stats, accel, missing = engine.trc_metrics(xyz, pairs)

    bone_results = []
    for (p, c, name), (mean, sd, n) in zip(present, bone_stats):
        n = int(n)
        if n == 0:
            bone_results.append({'name': name, 'parent': p, 'child': c, 'mean': np.nan, 'sd': np.nan, 'cv': np.nan, 'n_valid': 0})
        else:
            bone_results.append({'name': name, 'parent': p, 'child': c, 'mean': mean, 'sd': sd,
                                 'cv': (sd / mean * 100) if mean > 0 else np.nan, 'n_valid': n})
    smooth_results = []
    for m, name in enumerate(marker_names):
        valid = accel[m][~np.isnan(accel[m])] if F >= 3 else np.empty(0)
        if len(valid) == 0:
            smooth_results.append({'name': name, 'accel_median': np.nan, 'accel_p95': np.nan, 'accel_median_si': np.nan,
                                   'accel_p95_si': np.nan, 'n_valid': 0})
            continue
        med, p95 = float(np.median(valid)), float(np.percentile(valid, 95))
        smooth_results.append({'name': name, 'accel_median': med, 'accel_p95': p95, 'accel_median_si': med * fps * fps,
                               'accel_p95_si': p95 * fps * fps, 'n_valid': len(valid)})
    missing_results = [{'name': name, 'n_total': F, 'n_missing': int(missing[m]),
                        'missing_pct': int(missing[m]) / F * 100 if F > 0 else 0.0} for m, name in enumerate(marker_names)]
    by_name = {r['name']: r for r in bone_results}
    symmetry_results = []
    for left, right, pair in SYMMETRIC_BONE_PAIRS:
        if left not in by_name or right not in by_name:
            continue
        lm, rm = by_name[left]['mean'], by_name[right]['mean']
        if np.isnan(lm) or np.isnan(rm):
            diff = np.nan
        else:
            avg = (lm + rm) / 2
            diff = abs(lm - rm) / avg * 100 if avg > 0 else np.nan
        symmetry_results.append({'pair_name': pair, 'left_name': left, 'right_name': right, 'left_mean': lm, 'right_mean': rm, 'diff_pct': diff})
    return bone_results, smooth_results, missing_results, symmetry_results


def evaluate_single(trc_path, engine=None):
    """trc_evaluate.py:283-340: the same dictionary (tables + summary) for one .trc file."""
    if engine is None:
        from .filtering import _make_engine
        engine = _make_engine()
    marker_names, time_arr, xyz, fps = load_trc_as_marker_array(trc_path)
    bone_results, smooth_results, missing_results, symmetry_results = evaluate_arrays(xyz, marker_names, fps, engine)
    cvs = [b['cv'] for b in bone_results if not np.isnan(b['cv'])]
    worst = max(bone_results, key=lambda b: b['cv'] if not np.isnan(b['cv']) else -1) if bone_results else None
    p95s = [s['accel_p95'] for s in smooth_results if not np.isnan(s['accel_p95'])]
    total = sum(m['n_total'] for m in missing_results)
    diffs = [s['diff_pct'] for s in symmetry_results if not np.isnan(s['diff_pct'])]
    return {'trc_path': trc_path, 'n_frames': len(time_arr), 'n_markers': len(marker_names), 'fps': fps,
            'bone_results': bone_results, 'smooth_results': smooth_results, 'missing_results': missing_results,
            'symmetry_results': symmetry_results,
            'summary': {'mean_cv': np.mean(cvs) if cvs else np.nan,
                        'worst_bone': worst['name'] if worst else '',
                        'worst_cv': worst['cv'] if worst and not np.isnan(worst['cv']) else np.nan,
                        'mean_accel_p95': np.mean(p95s) if p95s else np.nan,
                        'overall_nan_pct': sum(m['n_missing'] for m in missing_results) / total * 100 if total > 0 else 0.0,
                        'mean_lr_diff': np.mean(diffs) if diffs else np.nan}}
