"""Goldens of the filtering stage and of trc_evaluate, recorded from the reference -> filter_units.npz

* butterworth_filter_1d (filtering.py:437-471) on columns with NaN / zero gaps, short runs, several orders, cut-offs
  and frame rates;
* filter_all (filtering.py:728-830) on a written .trc file: the text of the file it produces;
* compute_bone_lengths / compute_smoothness / compute_missing_data / compute_symmetry and evaluate_single's summary
  (Utilities/trc_evaluate.py:114-340) on that .trc file and on its filtered version.

The reference's filtering module imports statsmodels and filterpy at its top (LOESS and Kalman filters); both are
absent from this image (ordinary ModuleNotFoundError) and get empty stand-ins here, next to ref_shim's.
"""
import importlib
import io
import logging
import os
import sys
import tempfile
import types

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402


def load_filtering():
    ref_shim.install()
    for name in ('statsmodels', 'statsmodels.nonparametric', 'statsmodels.nonparametric.smoothers_lowess', 'filterpy',
                 'filterpy.kalman', 'filterpy.common'):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules['statsmodels.nonparametric.smoothers_lowess'].lowess = None
    sys.modules['filterpy.kalman'].KalmanFilter = None
    sys.modules['filterpy.common'].Q_discrete_white_noise = None
    return importlib.import_module('Pose2Sim.filtering'), importlib.import_module('Pose2Sim.Utilities.trc_evaluate')


def filter_config(project_dir, order, cutoff, frame_rate):
    return {'project': {'project_dir': project_dir, 'frame_rate': frame_rate, 'frame_range': 'auto'},
            'pose': {'vid_img_extension': 'mp4'},
            'filtering': {'type': 'butterworth', 'display_figures': False, 'save_filt_plots': False, 'filter': True,
                          'reject_outliers': False, 'make_c3d': False,
                          'butterworth': {'order': order, 'cut_off_frequency': cutoff},
                          'kalman': {'trust_ratio': 500, 'smooth': True},
                          'gcv_spline': {'cut_off_frequency': 'auto', 'smoothing_factor': 1.0},
                          'one_euro': {'cut_off_frequency': 2.5, 'beta': 0.9, 'd_cut_off_frequency': 1.0},
                          'butterworth_on_speed': {'order': 4, 'cut_off_frequency': 10},
                          'gaussian': {'sigma_kernel': 2}, 'loess': {'nb_values_used': 30}, 'median': {'kernel_size': 9}}}


def synthetic_trc_text(n_frames, frame_rate, seed, first_frame=0):
    """A HALPE_26 .trc as triangulate_all writes it (header of make_trc, triangulation.py:195-199), with gaps."""
    from pose2sim_amd import skeletons, synth
    ids, names, _ = skeletons.keypoints('HALPE_26')
    rng = np.random.default_rng(seed)
    Q = synth.make_points3d(n_frames, 1, len(ids), seed=seed)[:, 0] + rng.normal(0, 0.004, (n_frames, len(ids), 3))
    Q[rng.random((n_frames, len(ids))) < 0.01] = np.nan                       # isolated missing points
    g0 = n_frames // 3
    Q[g0:g0 + 7, 5] = np.nan                                                  # a gap that splits a column
    Q[:12, 9] = np.nan                                                        # a column that starts late
    Q[n_frames - 20:n_frames - 15, 11] = 0.0                                  # zeros count as invalid too
    Q[:, 17] = np.nan                                                         # a marker never seen
    name = f'trial_{first_frame}-{first_frame + n_frames}.trc'
    head = ['PathFileType\t4\t(X/Y/Z)\t' + name,
            'DataRate\tCameraRate\tNumFrames\tNumMarkers\tUnits\tOrigDataRate\tOrigDataStartFrame\tOrigNumFrames',
            '\t'.join(map(str, [frame_rate, frame_rate, n_frames, len(ids), 'm', frame_rate, first_frame, n_frames])),
            'Frame#\tTime\t' + '\t\t\t'.join(names) + '\t\t',
            '\t\t' + '\t'.join(f'X{i + 1}\tY{i + 1}\tZ{i + 1}' for i in range(len(ids)))]
    df = pd.DataFrame(Q.reshape(n_frames, -1), index=range(first_frame, first_frame + n_frames))
    df.insert(0, 't', df.index / frame_rate)
    buf = io.StringIO()
    df.to_csv(buf, sep='\t', index=True, header=None, lineterminator='\n')
    return name, '\n'.join(head) + '\n' + buf.getvalue()


def gen():
    filt, tev = load_filtering()
    logging.disable(logging.CRITICAL)
    rng = np.random.default_rng(2024)
    out = {}

    # ---- columns through butterworth_filter_1d -------------------------------------------------------------------------
    n = 0
    for case in range(40):
        L = int(rng.choice([5, 12, 30, 31, 40, 120, 400, 1500]))
        order = int(rng.choice([2, 4, 4, 6, 8]))
        frame_rate = int(rng.choice([30, 60, 100, 120]))
        cutoff = int(rng.choice([3, 6, 10]))
        t = np.arange(L) / frame_rate
        col = 1.2 + 0.4 * np.sin(2 * np.pi * 1.1 * t) + 0.05 * np.sin(2 * np.pi * 17 * t) + rng.normal(0, 0.01, L)
        mode = case % 5
        if mode == 1 and L > 20:
            col[rng.random(L) < 0.03] = np.nan
        elif mode == 2 and L > 40:
            g = int(rng.integers(5, L - 30)); col[g:g + int(rng.integers(1, 25))] = np.nan
            col[rng.random(L) < 0.02] = 0.0
        elif mode == 3:
            col[:int(rng.integers(1, max(2, L // 3)))] = np.nan
            col[L - int(rng.integers(1, max(2, L // 4))):] = 0.0
        elif mode == 4 and L > 60:
            for _ in range(4):
                g = int(rng.integers(0, L - 10)); col[g:g + int(rng.integers(1, 4))] = np.nan
        cfg = filter_config('.', order, cutoff, frame_rate)
        res = filt.butterworth_filter_1d(cfg, frame_rate, pd.Series(col.copy()))
        out[f'col{n}_in'] = col
        out[f'col{n}_prm'] = np.array([order, cutoff, frame_rate], dtype=np.int64)
        out[f'col{n}_out'] = np.asarray(res, dtype=np.float64)
        n += 1
    out['n_cols'] = np.array(n)

    # ---- filter_all and trc_evaluate on files ----------------------------------------------------------------------------
    n = 0
    for (frames, rate, order, cutoff, first) in ((240, 60, 4, 6, 0), (90, 30, 2, 3, 17)):
        with tempfile.TemporaryDirectory() as tmp:
            trial = os.path.join(tmp, 'trial')
            os.makedirs(os.path.join(trial, 'pose-3d'))
            name, text = synthetic_trc_text(frames, rate, seed=500 + n, first_frame=first)
            path = os.path.join(trial, 'pose-3d', name)
            with open(path, 'w') as fh:
                fh.write(text)
            cfg = filter_config(trial, order, cutoff, rate)
            filt.filter_all(cfg)
            produced = sorted(f for f in os.listdir(os.path.join(trial, 'pose-3d')) if 'filt' in f)
            assert len(produced) == 1, produced
            out[f'file{n}_name'] = np.array(name); out[f'file{n}_text'] = np.array(text)
            out[f'file{n}_prm'] = np.array([order, cutoff, rate], dtype=np.int64)
            out[f'file{n}_out_name'] = np.array(produced[0])
            out[f'file{n}_out_text'] = np.array(open(os.path.join(trial, 'pose-3d', produced[0])).read())
            for tag, p in (('raw', path), ('filt', os.path.join(trial, 'pose-3d', produced[0]))):
                ev = tev.evaluate_single(p)
                out[f'file{n}_{tag}_bones'] = np.array([[b['mean'], b['sd'], b['cv'], b['n_valid']] for b in ev['bone_results']], dtype=np.float64)
                out[f'file{n}_{tag}_bone_names'] = np.array([b['name'] for b in ev['bone_results']])
                out[f'file{n}_{tag}_smooth'] = np.array([[s['accel_median'], s['accel_p95'], s['accel_median_si'], s['accel_p95_si'], s['n_valid']]
                                                         for s in ev['smooth_results']], dtype=np.float64)
                out[f'file{n}_{tag}_missing'] = np.array([[m['n_total'], m['n_missing'], m['missing_pct']] for m in ev['missing_results']], dtype=np.float64)
                out[f'file{n}_{tag}_sym'] = np.array([[s['left_mean'], s['right_mean'], s['diff_pct']] for s in ev['symmetry_results']], dtype=np.float64)
                out[f'file{n}_{tag}_sym_names'] = np.array([s['pair_name'] for s in ev['symmetry_results']])
                sm = ev['summary']
                out[f'file{n}_{tag}_summary'] = np.array([sm['mean_cv'], sm['worst_cv'], sm['mean_accel_p95'], sm['overall_nan_pct'], sm['mean_lr_diff']])
                out[f'file{n}_{tag}_worst'] = np.array(sm['worst_bone'])
        n += 1
    out['n_files'] = np.array(n)
    np.savez_compressed(os.path.join(HERE, 'filter_units.npz'), **out)
    print('filter_units.npz:', len(out), 'arrays')


if __name__ == '__main__':
    gen()
