"""Goldens of the remaining column filters of the filtering stage, recorded from the reference -> filter_units2.npz

* hampel_filter (filtering.py:63-85), one_euro_filter_1d (:87-160), butterworth_on_speed_filter_1d (:474-510),
  gaussian_filter_1d (:513-529), median_filter_1d (:561-577) on columns with NaN / zero gaps, spikes and short runs;
* filter_all (:728-830) with reject_outliers = true (the shipped Demo_SinglePerson configuration: Hampel, then
  Butterworth order 4 at 6 Hz) and with each of the other types, on a written .trc file: the text it produces.

Kalman, GCV spline and LOESS are not recorded: they need filterpy / statsmodels, which are not importable here.
"""
import os
import sys
import tempfile

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_filter as g1  # noqa: E402


def column(rng, L, frame_rate, mode):
    t = np.arange(L) / frame_rate
    col = 1.2 + 0.4 * np.sin(2 * np.pi * 1.1 * t) + 0.05 * np.sin(2 * np.pi * 17 * t) + rng.normal(0, 0.01, L)
    if L > 20:
        spikes = rng.random(L) < 0.03
        col[spikes] += rng.normal(0, 0.3, int(spikes.sum()))                  # what the Hampel filter is for
    if mode == 1 and L > 20:
        col[rng.random(L) < 0.03] = np.nan
    elif mode == 2 and L > 40:
        g = int(rng.integers(5, L - 30)); col[g:g + int(rng.integers(1, 25))] = np.nan
        col[rng.random(L) < 0.02] = 0.0
    elif mode == 3:
        col[:int(rng.integers(1, max(2, L // 3)))] = np.nan
    elif mode == 4 and L > 60:
        for _ in range(4):
            g = int(rng.integers(0, L - 10)); col[g:g + int(rng.integers(1, 4))] = np.nan
    elif mode == 5 and L > 30:
        col[10:14] = col[9]                                                   # a flat stretch: zero MAD, zero speed
    return col


def gen():
    filt, _ = g1.load_filtering()
    import logging
    logging.disable(logging.CRITICAL)
    rng = np.random.default_rng(777)
    out = {}
    n = 0
    for case in range(48):
        L = int(rng.choice([2, 5, 9, 12, 30, 31, 40, 120, 400, 1500]))
        frame_rate = int(rng.choice([30, 60, 100, 120]))
        mode = case % 6
        col = column(rng, L, frame_rate, mode)
        order = int(rng.choice([2, 4, 4, 6])); cutoff = int(rng.choice([3, 6, 10]))
        sigma = int(rng.choice([1, 2, 3, 5])); ksize = int(rng.choice([3, 5, 9, 15]))
        mc, beta, dc = float(rng.choice([1.0, 2.5, 5.0])), float(rng.choice([0.0, 0.5, 0.9])), float(rng.choice([0.5, 1.0, 2.0]))
        cfg = g1.filter_config('.', order, cutoff, frame_rate)
        cfg['filtering']['butterworth_on_speed'] = {'order': order, 'cut_off_frequency': cutoff}
        cfg['filtering']['gaussian'] = {'sigma_kernel': sigma}
        cfg['filtering']['median'] = {'kernel_size': ksize}
        cfg['filtering']['one_euro'] = {'cut_off_frequency': mc, 'beta': beta, 'd_cut_off_frequency': dc}
        out[f'col{n}_in'] = col
        out[f'col{n}_prm'] = np.array([order, cutoff, frame_rate, sigma, ksize], dtype=np.int64)
        out[f'col{n}_euro'] = np.array([mc, beta, dc])
        out[f'col{n}_hampel'] = np.asarray(filt.hampel_filter(pd.Series(col.copy())), dtype=np.float64)
        out[f'col{n}_one_euro'] = np.asarray(filt.one_euro_filter_1d(cfg, frame_rate, pd.Series(col.copy())), dtype=np.float64)
        if L >= 2:
            out[f'col{n}_speed'] = np.asarray(filt.butterworth_on_speed_filter_1d(cfg, frame_rate, pd.Series(col.copy())), dtype=np.float64)
        out[f'col{n}_gauss'] = np.asarray(filt.gaussian_filter_1d(cfg, frame_rate, pd.Series(col.copy())), dtype=np.float64)
        if not np.isnan(col).any() and ksize <= L:
            out[f'col{n}_median'] = np.asarray(filt.median_filter_1d(cfg, frame_rate, pd.Series(col.copy())), dtype=np.float64)
        n += 1
    out['n_cols'] = np.array(n)

    # ---- filter_all on files: the shipped demo configuration (Hampel + Butterworth) and the other types ------------------
    n = 0
    for (frames, rate, first, ftype, reject, fill) in ((240, 60, 0, 'butterworth', True, False), (150, 30, 17, 'butterworth_on_speed', True, False),
                                                     (200, 60, 0, 'gaussian', False, False), (200, 60, 5, 'one_euro', True, False),
                                                     (180, 100, 0, 'median', False, True)):
        with tempfile.TemporaryDirectory() as tmp:
            trial = os.path.join(tmp, 'trial')
            os.makedirs(os.path.join(trial, 'pose-3d'))
            name, text = g1.synthetic_trc_text(frames, rate, seed=900 + n, first_frame=first)
            if fill:                                                          # medfilt: no NaN (a gap-filled trial)
                lines = text.split('\n')
                rows = pd.DataFrame([r.split('\t') for r in lines[5:] if r]).replace('', np.nan).astype(float)
                rows = rows.ffill().bfill().fillna(0.0)
                rows[0] = rows[0].astype(int)
                import io
                buf = io.StringIO(); rows.to_csv(buf, sep='\t', index=False, header=None, lineterminator='\n')
                text = '\n'.join(lines[:5]) + '\n' + buf.getvalue()
            path = os.path.join(trial, 'pose-3d', name)
            with open(path, 'w') as fh:
                fh.write(text)
            cfg = g1.filter_config(trial, 4, 6, rate)
            cfg['filtering']['type'] = ftype
            cfg['filtering']['reject_outliers'] = reject
            cfg['filtering']['median'] = {'kernel_size': 5}
            filt.filter_all(cfg)
            produced = sorted(f for f in os.listdir(os.path.join(trial, 'pose-3d')) if 'filt' in f)
            assert len(produced) == 1, produced
            out[f'file{n}_name'] = np.array(name); out[f'file{n}_text'] = np.array(text)
            out[f'file{n}_type'] = np.array(ftype); out[f'file{n}_reject'] = np.array(reject); out[f'file{n}_rate'] = np.array(rate)
            out[f'file{n}_out_name'] = np.array(produced[0])
            out[f'file{n}_out_text'] = np.array(open(os.path.join(trial, 'pose-3d', produced[0])).read())
        n += 1
    out['n_files'] = np.array(n)
    np.savez_compressed(os.path.join(HERE, 'filter_units2.npz'), **out)
    print('filter_units2.npz:', len(out), 'arrays')


if __name__ == '__main__':
    gen()
