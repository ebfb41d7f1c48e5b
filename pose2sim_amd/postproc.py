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


def _interpolate_column(vals, labels, max_gap, kind):
    """One column of interpolate_zeros_nans (common.py:669-712) on arrays: vals [F] with labels [F] (the frame numbers
    interp1d sees as abscissae, increasing).  Samples that are NaN or 0 are replaced by scipy's interp1d through the others
    ('extrapolate' beyond the ends) unless they sit in a run of more than max_gap of them; with 4 or fewer good samples
    the column comes back untouched (None).  Returns (positions, values): the samples to overwrite -- the interpolant is
    evaluated where it is needed only (the same numbers as evaluating it everywhere and keeping those)."""
    good = ~(np.isnan(vals) | (vals == 0))
    if int(good.sum()) <= 4:
        return None
    bad_pos = np.flatnonzero(~good)
    if not bad_pos.size:
        return bad_pos, vals[:0]
    if kind is None:
        f_interp = interpolate.interp1d(labels[good], vals[good], kind='linear', bounds_error=False, assume_sorted=True)
    else:
        f_interp = interpolate.interp1d(labels[good], vals[good], kind=kind, fill_value='extrapolate', bounds_error=False, assume_sorted=True)
    new = f_interp(labels[bad_pos])
    # runs of consecutive bad labels longer than max_gap stay NaN (:704-710)
    bad_labels = labels[bad_pos]
    starts = np.concatenate(([0], np.flatnonzero(np.diff(bad_labels) > 1) + 1))
    lengths = np.diff(np.concatenate((starts, [bad_pos.size])))
    for s0, ln in zip(starts[lengths > max_gap], lengths[lengths > max_gap]):
        new[s0:s0 + ln] = np.nan
    return bad_pos, new


def interpolate_zeros_nans(col, *args):
    """The reference's per-column entry (common.py:669-712) for a pandas column: args = (N, kind) | (kind,) | ()."""
    import pandas as pd
    max_gap, kind = (args if len(args) == 2 else (np.inf, args[0] if args else None))
    vals = np.array(col, dtype=np.float64)
    patch = _interpolate_column(vals, np.asarray(col.index), max_gap, kind)
    if patch is None:
        return col
    vals[patch[0]] = patch[1]
    return pd.Series(vals, index=col.index, name=getattr(col, 'name', None))


def interpolate_gaps(coords, labels, max_gap, kind):
    """Every column of coords [F][3 K] through the gap interpolation, IN PLACE (triangulation.py:889-894: all columns or,
    if one of them fails, none -- the caller logs the reference's warning; the overwritten samples are put back then).
    Only the gaps are touched: at 100 k frames the table is 60 MB and a copy of it costs more than the interpolation."""
    undo = []
    try:
        for c in range(coords.shape[1]):
            patch = _interpolate_column(coords[:, c], labels, max_gap, kind)
            if patch is not None and patch[0].size:
                undo.append((c, patch[0], coords[patch[0], c]))
                coords[patch[0], c] = patch[1]
    except Exception:
        for c, pos, old in undo:
            coords[pos, c] = old
        raise
    return coords


def fill_gaps(coords, how):
    """triangulation.py:922-926.  'last_value': every gap takes the last valid value before it, leading gaps the first
    valid one after them, and what is still missing (a column never seen) or +inf becomes 0; 'zeros': NaN and +inf
    become 0; anything else leaves the gaps.  IN PLACE, column by column (one frame-long temporary at a time)."""
    if how not in ('last_value', 'zeros'):
        return coords
    out = coords
    F = out.shape[0]
    rows = np.arange(F)
    for c in range(out.shape[1]):
        col = out[:, c]
        missing = np.isnan(col)
        if how == 'last_value' and missing.any() and not missing.all():
            last = np.maximum.accumulate(np.where(missing, -1, rows))          # row of the last valid sample, or -1
            first = int(np.argmin(missing))                                    # leading gaps: the first valid sample
            col[missing] = col[np.where(last >= 0, last, first)[missing]]
            missing = np.isnan(col)
        col[missing | (col == np.inf)] = 0
    return out


def gap_spans(x_coords, first, last, max_gap):
    """triangulation.py:917-919, 946-953: per keypoint, the runs of frames (positions in the trimmed trial) whose X is 0
    or not finite, as 'a:b' strings -- those short enough to have been interpolated, and the others.  Positions are kept
    only if they lie strictly between `first` and `last`, the trial's own bounds (the reference compares positions in
    the trimmed trial with positions in the whole one)."""
    done, left = [], []
    for k in range(x_coords.shape[1]):
        pos = np.flatnonzero((x_coords[:, k] == 0) | ~np.isfinite(x_coords[:, k]))
        pos = pos[(pos > first) & (pos < last)]
        runs = np.split(pos, np.flatnonzero(np.diff(pos) > 1) + 1)
        done.append([f'{r[0]}:{r[-1]}' for r in runs if 0 < len(r) <= max_gap])
        left.append([f'{r[0]}:{r[-1]}' for r in runs if len(r) > max_gap])
    return done, left


def camera_exclusion_fractions(mask_rows, n_cams):
    """triangulation.py:933-943: per camera, the share of (frame, keypoint) units that excluded it, from the kernels'
    excluded-camera bit masks [F'][K]."""
    flat = mask_rows.reshape(-1).astype(np.uint64)
    total = flat.size
    return {c: int(np.count_nonzero((flat >> np.uint64(c)) & np.uint64(1))) / total for c in range(n_cams)}


def frame_means(table, skipna=True):
    """Row means with pandas' DataFrame.mean(axis=1) arithmetic (what the reference trims and reports by)."""
    import pandas as pd
    return pd.DataFrame(table).mean(axis=1, skipna=skipna).to_numpy()


def column_means(table):
    """Column means with pandas' DataFrame.mean() arithmetic (NaN skipped)."""
    import pandas as pd
    return pd.DataFrame(table).mean().to_numpy()


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
