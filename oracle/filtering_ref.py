"""TEST INFRASTRUCTURE -- CPU restatement of the reference's Butterworth filtering and trc_evaluate metrics.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(pose2sim_amd/) never does.  Restates, with the reference's own third-party calls (SciPy / NumPy are the algorithm
there, they are present on the GPU box and pinned by the goldens recorded from the reference):

* butterworth_filter_1d   Pose2Sim/filtering.py:437-471
* compute_bone_lengths / compute_smoothness / compute_missing_data / compute_symmetry
                          Pose2Sim/Utilities/trc_evaluate.py:114-280
"""
import numpy as np
from scipy import signal


def butterworth_filter_1d(col, order, cutoff, frame_rate):
    """filtering.py:437-471: zero-phase low-pass on every run of valid samples (not NaN, not 0) longer than padlen."""
    order, cutoff = int(order), int(cutoff)
    b, a = signal.butter(order / 2, cutoff / (frame_rate / 2), 'low', analog=False)        # :456
    padlen = 3 * max(len(a), len(b))                                                        # :457
    out = np.array(col, dtype=np.float64)
    mask = np.isnan(out) | (out == 0)                                                       # :461
    good = np.where(~mask)[0]
    gaps = np.where(np.diff(good) > 1)[0] + 1
    for seq in np.split(good, gaps):                                                        # :463-470
        if len(seq) > padlen:
            out[seq] = signal.filtfilt(b, a, out[seq])
    return out


def butterworth_filter(data, order, cutoff, frame_rate):
    """Q_coords.apply(butterworth_filter_1d, axis=0): data [n_frames][n_cols]."""
    data = np.asarray(data, dtype=np.float64)
    return np.stack([butterworth_filter_1d(data[:, c], order, cutoff, frame_rate) for c in range(data.shape[1])], axis=1)


def bone_lengths(xyz, bones):
    """trc_evaluate.py:114-156.  xyz [F][K][3]; bones: (parent index, child index).  -> per bone (mean, sd, cv, n_valid)
    and the per-frame lengths [n_bones][F]."""
    stats, lens = [], []
    for p, c in bones:
        l = np.linalg.norm(xyz[:, c] - xyz[:, p], axis=1)
        l[l == 0.0] = np.nan
        lens.append(l)
        n = int(np.sum(~np.isnan(l)))
        if n == 0:
            stats.append((np.nan, np.nan, np.nan, 0))
            continue
        mean, sd = np.nanmean(l), np.nanstd(l)
        stats.append((mean, sd, (sd / mean * 100) if mean > 0 else np.nan, n))
    return stats, np.array(lens).reshape(len(bones), xyz.shape[0])


def smoothness(xyz, fps):
    """trc_evaluate.py:159-207 per marker: (median, p95, median_si, p95_si, n_valid) of |second difference|."""
    out = []
    F, K = xyz.shape[:2]
    for m in range(K):
        pos = xyz[:, m]
        if F < 3:
            out.append((np.nan, np.nan, np.nan, np.nan, 0))
            continue
        acc = np.linalg.norm(pos[2:] - 2 * pos[1:-1] + pos[:-2], axis=1)
        valid = acc[~np.isnan(acc)]
        if len(valid) == 0:
            out.append((np.nan, np.nan, np.nan, np.nan, 0))
            continue
        med, p95 = float(np.median(valid)), float(np.percentile(valid, 95))
        out.append((med, p95, med * fps * fps, p95 * fps * fps, len(valid)))
    return out


def missing_data(xyz):
    """trc_evaluate.py:210-238 per marker: (n_total, n_missing, missing_pct)."""
    F, K = xyz.shape[:2]
    out = []
    for m in range(K):
        n = int(np.sum(np.any(np.isnan(xyz[:, m]), axis=1)))
        out.append((F, n, n / F * 100 if F > 0 else 0.0))
    return out
