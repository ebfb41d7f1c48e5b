"""Sequential host steps after the kernel: person tracking across frames, gap interpolation,
valid-section selection, gap filling.  They run on the gathered result on rank 0 (SURVEY.md
section 8e: these steps carry state from frame to frame and do not shard).
"""
import numpy as np
from scipy import interpolate
from scipy.optimize import linear_sum_assignment


# common.py:406-424
def pad_shape(arr, target_len, fill_value=np.nan):
    if len(arr) < target_len:
        pad = np.full((target_len - len(arr),) + arr.shape[1:], fill_value)
        return np.concatenate((arr, pad))
    return arr


# common.py:1037-1136 (scores=None branch, the one triangulation.py:852 uses)
def sort_people_sports2d(keyptpre, keypt, max_dist=None):
    """Associate persons between the previous and the current frame by mean keypoint distance and
    the Hungarian algorithm.  Returns (sorted_prev_keypoints, sorted_keypoints, sorted_ids)."""
    n_prev, n_curr = len(keyptpre), len(keypt)
    if n_prev == 0 and n_curr == 0:
        return np.array([]), np.array([])
    if n_prev == 0:
        return np.array([]), keypt
    diff = keypt[np.newaxis, :, :, :] - keyptpre[:, np.newaxis, :, :]
    per_kpt = np.sqrt(np.nansum(diff ** 2, axis=3))        # an all-NaN keypoint sums to 0, so nothing here is NaN
    dist_matrix = np.nanmean(per_kpt, axis=2)
    dist_matrix = np.nan_to_num(dist_matrix, nan=1e10, posinf=1e10)
    pre_ids, curr_ids = linear_sum_assignment(dist_matrix)
    if max_dist is not None:
        valid = [(p, c) for p, c in zip(pre_ids, curr_ids) if dist_matrix[p, c] <= max_dist]
    else:
        valid = list(zip(pre_ids, curr_ids))
    associated = set(c for _, c in valid)
    unassociated = [i for i in range(n_curr) if i not in associated]
    n_total = n_prev + len(unassociated)
    sorted_keypoints = np.full((n_total,) + keypt.shape[1:], np.nan)
    sorted_ids = np.full(n_total, -1)
    for p, c in valid:
        sorted_keypoints[p] = keypt[c]
        sorted_ids[p] = c
    for new_idx, c in enumerate(unassociated):
        sorted_keypoints[n_prev + new_idx] = keypt[c]
        sorted_ids[n_prev + new_idx] = c
    keyptpre_padded = pad_shape(keyptpre, n_total, fill_value=np.nan)
    sorted_prev = np.where(np.isnan(sorted_keypoints) & ~np.isnan(keyptpre_padded), keyptpre_padded, sorted_keypoints)
    return sorted_prev, sorted_keypoints, sorted_ids


# common.py:669-712
def interpolate_zeros_nans(col, *args):
    """Interpolate the zeros / NaNs of a pandas column unless more than N are contiguous (common.py:669-712).
    Same calls into scipy's interp1d as the reference (so the same numbers), with the index bookkeeping done
    on arrays instead of Python lists: 78 columns x 100 k frames went from ~5 s to well under one."""
    kind = None
    if len(args) == 2:
        N, kind = args
    elif len(args) == 1:
        N = np.inf
        kind = args[0]
    else:
        N = np.inf
    vals = np.asarray(col, dtype=np.float64)
    labels = np.asarray(col.index)
    mask = ~(np.isnan(vals) | (vals == 0))
    if int(mask.sum()) <= 4:
        return col
    idx_good = labels[mask]
    if kind is None:
        f_interp = interpolate.interp1d(idx_good, vals[mask], kind='linear', bounds_error=False)
    else:
        f_interp = interpolate.interp1d(idx_good, vals[mask], kind=kind, fill_value='extrapolate', bounds_error=False)
    out = np.where(mask, vals, f_interp(labels))
    # runs of consecutive bad labels longer than N go back to NaN (:704-710)
    bad_pos = np.flatnonzero(~mask)
    if bad_pos.size:
        bad_labels = labels[bad_pos]
        starts = np.concatenate(([0], np.flatnonzero(np.diff(bad_labels) > 1) + 1))
        lengths = np.diff(np.concatenate((starts, [bad_pos.size])))
        for s0, ln in zip(starts[lengths > N], lengths[lengths > N]):
            out[bad_pos[s0:s0 + ln]] = np.nan
    import pandas as pd
    return pd.Series(out, index=col.index, name=getattr(col, 'name', None))


# triangulation.py:93-148
def indices_of_first_last_non_nan_chunks(series, min_chunk_size=10, chunk_choice_method='largest'):
    """(start, end) positions of the kept section(s) of consecutive non-NaN values."""
    min_chunk_size = 10 if min_chunk_size is None else min_chunk_size
    ok = ~np.isnan(np.asarray(series.values if hasattr(series, 'values') else series, dtype=np.float64))
    # runs of True
    padded = np.concatenate([[False], ok, [False]]).astype(np.int8)
    d = np.diff(padded)
    starts = np.flatnonzero(d == 1)
    ends = np.flatnonzero(d == -1)
    valid = [(int(s), int(e)) for s, e in zip(starts, ends) if e - s >= min_chunk_size]
    if not valid:
        return 0, 0
    if chunk_choice_method not in ['largest', 'all', 'first', 'last']:
        chunk_choice_method = 'all'
    if chunk_choice_method == 'largest':
        # list.sort is stable: among equally long runs the earliest wins
        return sorted(valid, key=lambda r: r[1] - r[0], reverse=True)[0]
    if chunk_choice_method == 'all':
        return valid[0][0], valid[-1][1]
    if chunk_choice_method == 'first':
        return valid[0]
    return valid[-1]
