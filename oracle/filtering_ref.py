"""TEST INFRASTRUCTURE -- CPU restatement of the reference's Butterworth filtering and trc_evaluate metrics.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(pose2sim_amd/) never does.  Restates, with the reference's own third-party calls (SciPy / NumPy are the algorithm
there, they are present on the GPU box and pinned by the goldens recorded from the reference):

* butterworth_filter_1d   Pose2Sim/filtering.py:437-471
* hampel_filter           Pose2Sim/filtering.py:63-85
* one_euro_filter_1d      Pose2Sim/filtering.py:87-160
* butterworth_on_speed_filter_1d  Pose2Sim/filtering.py:474-510
* gaussian_filter_1d      Pose2Sim/filtering.py:513-529
* median_filter_1d        Pose2Sim/filtering.py:561-577
* kalman_filter_1d        Pose2Sim/filtering.py:316-434 -- PARITY UNPINNED: the reference calls filterpy (KalmanFilter.batch_filter,
                          rts_smoother, Q_discrete_white_noise; version not pinned in pyproject.toml), which is not importable
                          here; its published algorithm is restated and no golden vector exists
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


def hampel_filter(col, window_size=7, n_sigma=2):
    """filtering.py:63-85: sliding median / MAD outlier replacement; the window is read from the INPUT column."""
    col = np.asarray(col, dtype=np.float64)
    out = col.copy()
    half = window_size // 2
    for i in range(half, len(col) - half):                                                 # :74
        window = col[i - half:i + half + 1]
        median = np.median(window)
        mad = np.median(np.abs(window - median))
        if mad != 0:                                                                       # :79 (true for NaN)
            z = 0.6745 * (col[i] - median) / mad
            if np.abs(z) > n_sigma:
                out[i] = median
    return out


def _runs(valid):
    good = np.where(valid)[0]
    return np.split(good, np.where(np.diff(good) > 1)[0] + 1)


def one_euro_filter_1d(col, frame_rate, min_cutoff=2.5, beta=0.9, d_cutoff=1.0):
    """filtering.py:87-160: zero-phase (forward, then backward) one-euro filter of every run of >= 2 non-NaN samples."""
    dt = 1.0 / frame_rate

    def smoothing_factor(dt, cutoff):                                                      # :106-111
        r = 2 * np.pi * cutoff * dt
        return r / (r + 1)

    def apply_filter(data):                                                                # :113-142
        if len(data) < 2:
            return data
        filtered = [data[0]]
        x_prev, dx_prev = data[0], 0.0
        for i in range(1, len(data)):
            x = data[i]
            alpha_d = smoothing_factor(dt, d_cutoff)
            dx = (x - x_prev) / dt
            dx_hat = alpha_d * dx + (1 - alpha_d) * dx_prev
            cutoff = min_cutoff + beta * abs(dx_hat)
            alpha = smoothing_factor(dt, cutoff)
            x_hat = alpha * x + (1 - alpha) * x_prev
            filtered.append(x_hat)
            x_prev, dx_prev = x_hat, dx_hat
        return np.array(filtered)

    out = np.array(col, dtype=np.float64)
    for seq in _runs(~np.isnan(out)):                                                      # :144-158
        if len(seq) >= 2:
            fwd = apply_filter(out[seq])
            out[seq] = apply_filter(fwd[::-1])[::-1]
    return out


def butterworth_on_speed_filter_1d(col, order, cutoff, frame_rate):
    """filtering.py:474-510: the Butterworth filter on the first difference, integrated back (pandas semantics:
    fillna fills EVERY NaN of the difference with half its second value, cumsum skips NaN)."""
    import pandas as pd
    order, cutoff = int(order), int(cutoff)
    b, a = signal.butter(order / 2, cutoff / (frame_rate / 2), 'low', analog=False)
    padlen = 3 * max(len(a), len(b))
    col = pd.Series(np.asarray(col, dtype=np.float64))
    d = col.diff()
    d = d.fillna(d.iloc[1] / 2)                                                            # :494
    mask = np.isnan(d) | d.eq(0)
    for seq in _runs(~mask.to_numpy()):
        if len(seq) > padlen:
            d[seq] = signal.filtfilt(b, a, d[seq])
    return (d.cumsum() + col.iloc[0]).to_numpy()                                          # :508


def gaussian_filter_1d(col, sigma_kernel):
    """filtering.py:513-529."""
    from scipy.ndimage import gaussian_filter1d
    return gaussian_filter1d(np.asarray(col, dtype=np.float64), int(sigma_kernel))


def median_filter_1d(col, kernel_size):
    """filtering.py:561-577."""
    return signal.medfilt(np.asarray(col, dtype=np.float64), kernel_size=kernel_size)


def kalman_filter_1d(col, frame_rate, trust_ratio, smooth=True):
    """filtering.py:316-434 with filterpy's published algorithms restated (PARITY UNPINNED, see the header): constant-
    acceleration model of one coordinate (state position / velocity / acceleration), measurement noise 20, process noise
    20 * trust_ratio, predict + update per sample (KalmanFilter.batch_filter), then the Rauch-Tung-Striebel smoother."""
    measurement_noise = 20
    process_noise = measurement_noise * int(trust_ratio)
    out = np.array(col, dtype=np.float64)
    dt = 1 / frame_rate
    F = np.array([[1.0, dt, dt ** 2 / 2], [0.0, 1.0, dt], [0.0, 0.0, 1.0]])                 # :355-359
    H = np.array([[1.0, 0.0, 0.0]])
    R = np.array([[float(measurement_noise ** 2)]])
    var = process_noise ** 2
    Q = np.array([[.25 * dt ** 4, .5 * dt ** 3, .5 * dt ** 2], [.5 * dt ** 3, dt ** 2, dt], [.5 * dt ** 2, dt, 1.0]]) * var   # Q_discrete_white_noise(3)
    for seq in _runs(~(np.isnan(out) | (out == 0))):
        if len(seq) < 4:                                                                   # :428
            continue
        z = out[seq]
        x = np.array([z[0], np.diff(z, 1)[0] / dt, np.diff(np.diff(z) / dt)[0] / dt])       # :343-351
        P = np.eye(3) * measurement_noise                                                   # :377
        xs, Ps = [], []
        I = np.eye(3)
        for zk in z:
            x = F @ x                                                                       # predict
            P = F @ P @ F.T + Q
            y = zk - (H @ x)                                                                # update
            S = H @ P @ H.T + R
            K = P @ H.T @ np.linalg.inv(S)
            x = x + (K @ y)
            IKH = I - K @ H
            P = IKH @ P @ IKH.T + K @ R @ K.T
            xs.append(x.copy()); Ps.append(P.copy())
        xs, Ps = np.array(xs), np.array(Ps)
        if smooth:                                                                          # rts_smoother
            for k in range(len(z) - 2, -1, -1):
                Pp = F @ Ps[k] @ F.T + Q
                Kk = Ps[k] @ F.T @ np.linalg.inv(Pp)
                xs[k] += Kk @ (xs[k + 1] - F @ xs[k])
                Ps[k] += Kk @ (Ps[k + 1] - Pp) @ Kk.T
        out[seq] = xs[:, 0]
    return out


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
