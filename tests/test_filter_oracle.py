"""Filtering stage and trc_evaluate on CPU: the oracle (oracle/filtering_ref.py) against the goldens recorded from the
reference (tests/golden/filter_units.npz <- make_golden_filter.py), and the host mirror (pose2sim_amd/filtering.py,
trc_evaluate.py) with an oracle-backed test double standing in for the HIP engine: the .trc file it writes must equal
the reference's byte for byte."""
import os

import numpy as np
import pytest

from oracle import filtering_ref as fr


class OracleFilterEngine:
    """Test double of Engine.butterworth / Engine.trc_metrics (no GPU in this container)."""

    def butterworth(self, data, b, a, zi):
        from scipy import signal
        data = np.asarray(data, dtype=np.float64)
        out = data.copy()
        padlen = 3 * max(len(a), len(b))
        for c in range(data.shape[1]):
            col = out[:, c]
            good = np.where(~(np.isnan(col) | (col == 0)))[0]
            for seq in np.split(good, np.where(np.diff(good) > 1)[0] + 1):
                if len(seq) > padlen:
                    col[seq] = signal.filtfilt(b, a, col[seq])
        return out

    def filter_columns(self, kind, data, params):
        data = np.asarray(data, dtype=np.float64)
        params = np.asarray(params, dtype=np.float64).reshape(-1)
        per_col = {1: lambda c: fr.hampel_filter(c, 7, params[0]),
                   2: lambda c: __import__('scipy.ndimage', fromlist=['correlate1d']).correlate1d(c, params, mode='reflect'),
                   3: lambda c: fr.median_filter_1d(c, int(params[0])),
                   4: lambda c: fr.one_euro_filter_1d(c, 1.0 / params[0], params[1], params[2], params[3]),
                   5: lambda c: fr.kalman_filter_1d(c, 1.0 / params[0], int(round(params[2] / params[1])), bool(params[3]))}[kind]
        if data.shape[1] == 0:
            return data.copy()
        return np.stack([per_col(data[:, c]) for c in range(data.shape[1])], axis=1)

    def trc_metrics(self, xyz, bones):
        stats, lens = fr.bone_lengths(xyz, [tuple(b) for b in bones])
        F, K = xyz.shape[:2]
        accel = np.full((K, max(F - 2, 0)), np.nan)
        if F >= 3:
            for m in range(K):
                accel[m] = np.linalg.norm(xyz[2:, m] - 2 * xyz[1:-1, m] + xyz[:-2, m], axis=1)
        missing = np.array([m[1] for m in fr.missing_data(xyz)], dtype=np.int64)
        return lens, np.array([[s[0], s[1], s[3]] for s in stats]).reshape(len(bones), 3), accel, missing


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'filter_units.npz'))


def test_oracle_columns_match_the_reference(gold):
    for i in range(int(gold['n_cols'])):
        order, cutoff, rate = (int(v) for v in gold[f'col{i}_prm'])
        got = fr.butterworth_filter_1d(gold[f'col{i}_in'], order, cutoff, rate)
        assert np.array_equal(got, gold[f'col{i}_out'], equal_nan=True), i


def test_coefficients_are_the_reference_calls():
    from scipy import signal
    from pose2sim_amd import filtering
    b, a, zi = filtering.butterworth_coefficients(4, 6, 60)
    rb, ra = signal.butter(2.0, 6 / 30.0, 'low', analog=False)
    assert np.array_equal(b, rb) and np.array_equal(a, ra) and a[0] == 1.0
    assert np.array_equal(zi, signal.lfilter_zi(rb, ra)) and len(zi) == len(b) - 1


@pytest.fixture
def work_dir():
    """A scratch directory whose path does not contain 'filt': the reference skips every .trc whose PATH does
    (filtering.py:777), and pytest's tmp_path is named after the test."""
    import shutil
    import tempfile
    from pathlib import Path
    d = tempfile.mkdtemp(prefix='p2s_bw_')
    yield Path(d)
    shutil.rmtree(d, ignore_errors=True)


def _write_trial(tmp_path, gold, i):
    trial = tmp_path / f'trial{i}'
    (trial / 'pose-3d').mkdir(parents=True)
    (trial / 'pose-3d' / str(gold[f'file{i}_name'])).write_text(str(gold[f'file{i}_text']))
    order, cutoff, rate = (int(v) for v in gold[f'file{i}_prm'])
    cfg = {'project': {'project_dir': str(trial), 'frame_rate': rate, 'frame_range': 'auto'}, 'pose': {'vid_img_extension': 'mp4'},
           'filtering': {'type': 'butterworth', 'filter': True, 'reject_outliers': False,
                         'butterworth': {'order': order, 'cut_off_frequency': cutoff}}}
    return trial, cfg


def test_filter_all_writes_the_reference_file(work_dir, gold):
    tmp_path = work_dir
    from pose2sim_amd import filtering
    for i in range(int(gold['n_files'])):
        trial, cfg = _write_trial(tmp_path, gold, i)
        paths = filtering.filter_all(cfg, engine=OracleFilterEngine())
        assert [os.path.basename(p) for p in paths] == [str(gold[f'file{i}_out_name'])]
        assert open(paths[0]).read() == str(gold[f'file{i}_out_text'])


def test_filter_types_outside_the_build_are_refused(work_dir, gold):
    tmp_path = work_dir
    from pose2sim_amd import filtering
    trial, cfg = _write_trial(tmp_path, gold, 0)
    for t in ('gcv_spline', 'loess'):
        cfg['filtering']['type'] = t
        with pytest.raises(NotImplementedError):
            filtering.filter_all(cfg, engine=OracleFilterEngine())
    cfg['filtering']['type'] = 'no_such_filter'
    with pytest.raises(KeyError):                      # the reference's filter_mapping[filter_type]
        filtering.filter_all(cfg, engine=OracleFilterEngine())


@pytest.fixture(scope='module')
def gold2(golden_dir):
    return np.load(os.path.join(golden_dir, 'filter_units2.npz'))


def test_oracle_of_the_other_filters_matches_the_reference(gold2):
    """hampel, one_euro, butterworth_on_speed, gaussian, median: 48 columns each (21 for medfilt, which wants no NaN),
    bit for bit against what the reference computed (tests/golden/make_golden_filter2.py)."""
    g = gold2
    for i in range(int(g['n_cols'])):
        col = g[f'col{i}_in']
        order, cutoff, rate, sigma, ksize = (int(v) for v in g[f'col{i}_prm'])
        mc, beta, dc = g[f'col{i}_euro']
        assert np.array_equal(fr.hampel_filter(col), g[f'col{i}_hampel'], equal_nan=True), ('hampel', i)
        assert np.array_equal(fr.one_euro_filter_1d(col, rate, mc, beta, dc), g[f'col{i}_one_euro'], equal_nan=True), ('one_euro', i)
        if f'col{i}_speed' in g.files:
            assert np.array_equal(fr.butterworth_on_speed_filter_1d(col, order, cutoff, rate), g[f'col{i}_speed'], equal_nan=True), ('speed', i)
        assert np.array_equal(fr.gaussian_filter_1d(col, sigma), g[f'col{i}_gauss'], equal_nan=True), ('gauss', i)
        if f'col{i}_median' in g.files:
            assert np.array_equal(fr.median_filter_1d(col, ksize), g[f'col{i}_median'], equal_nan=True), ('median', i)


def _write_trial2(tmp_path, g, i):
    trial = tmp_path / f'trial2_{i}'
    (trial / 'pose-3d').mkdir(parents=True)
    (trial / 'pose-3d' / str(g[f'file{i}_name'])).write_text(str(g[f'file{i}_text']))
    cfg = {'project': {'project_dir': str(trial), 'frame_rate': int(g[f'file{i}_rate']), 'frame_range': 'auto'},
           'pose': {'vid_img_extension': 'mp4'},
           'filtering': {'type': str(g[f'file{i}_type']), 'filter': True, 'reject_outliers': bool(g[f'file{i}_reject']), 'make_c3d': False,
                         'butterworth': {'order': 4, 'cut_off_frequency': 6}, 'butterworth_on_speed': {'order': 4, 'cut_off_frequency': 10},
                         'one_euro': {'cut_off_frequency': 2.5, 'beta': 0.9, 'd_cut_off_frequency': 1.0},
                         'gaussian': {'sigma_kernel': 2}, 'median': {'kernel_size': 5}}}
    return trial, cfg


def test_filter_all_with_outlier_rejection_and_every_type_writes_the_reference_file(work_dir, gold2):
    """The shipped Demo_SinglePerson filtering table (reject_outliers = true, Butterworth order 4 at 6 Hz,
    Config.toml:214-226) and the other filter types: the file equals the reference's byte for byte."""
    from pose2sim_amd import filtering
    g = gold2
    kinds = set()
    for i in range(int(g['n_files'])):
        trial, cfg = _write_trial2(work_dir, g, i)
        paths = filtering.filter_all(cfg, engine=OracleFilterEngine())
        assert [os.path.basename(p) for p in paths] == [str(g[f'file{i}_out_name'])]
        assert open(paths[0]).read() == str(g[f'file{i}_out_text']), (i, cfg['filtering']['type'])
        kinds.add((cfg['filtering']['type'], cfg['filtering']['reject_outliers']))
    assert ('butterworth', True) in kinds and len(kinds) == 5


def test_outlier_rejection_without_a_filter_fails_like_the_reference(work_dir, gold2):
    from pose2sim_amd import filtering
    trial, cfg = _write_trial2(work_dir, gold2, 0)
    cfg['filtering']['filter'] = False
    with pytest.raises(UnboundLocalError):             # filtering.py:804-815 uses Q_filt, which only the filter assigns
        filtering.filter_all(cfg, engine=OracleFilterEngine())


def _check_evaluation(ev, gold, pre):
    assert [b['name'] for b in ev['bone_results']] == [str(s) for s in gold[pre + 'bone_names']]
    got = np.array([[b['mean'], b['sd'], b['cv'], b['n_valid']] for b in ev['bone_results']], dtype=np.float64)
    np.testing.assert_allclose(got, gold[pre + 'bones'], rtol=1e-12, atol=0, equal_nan=True)
    got = np.array([[s['accel_median'], s['accel_p95'], s['accel_median_si'], s['accel_p95_si'], s['n_valid']] for s in ev['smooth_results']])
    np.testing.assert_allclose(got, gold[pre + 'smooth'], rtol=1e-12, atol=0, equal_nan=True)
    got = np.array([[m['n_total'], m['n_missing'], m['missing_pct']] for m in ev['missing_results']], dtype=np.float64)
    np.testing.assert_allclose(got, gold[pre + 'missing'], rtol=1e-15, equal_nan=True)
    assert [s['pair_name'] for s in ev['symmetry_results']] == [str(s) for s in gold[pre + 'sym_names']]
    got = np.array([[s['left_mean'], s['right_mean'], s['diff_pct']] for s in ev['symmetry_results']])
    np.testing.assert_allclose(got, gold[pre + 'sym'], rtol=1e-11, equal_nan=True)
    sm = ev['summary']
    np.testing.assert_allclose([sm['mean_cv'], sm['worst_cv'], sm['mean_accel_p95'], sm['overall_nan_pct'], sm['mean_lr_diff']],
                               gold[pre + 'summary'], rtol=1e-11, equal_nan=True)
    assert sm['worst_bone'] == str(gold[pre + 'worst'])


def test_trc_evaluate_matches_the_reference(tmp_path, gold):
    from pose2sim_amd import trc_evaluate
    for i in range(int(gold['n_files'])):
        for tag, name, text in (('raw', gold[f'file{i}_name'], gold[f'file{i}_text']),
                                ('filt', gold[f'file{i}_out_name'], gold[f'file{i}_out_text'])):
            p = tmp_path / f'{i}_{tag}_{name}'
            p.write_text(str(text))
            ev = trc_evaluate.evaluate_single(str(p), engine=OracleFilterEngine())
            _check_evaluation(ev, gold, f'file{i}_{tag}_')


def test_kalman_stage_runs_and_reports(work_dir, gold, caplog):
    """type = 'kalman' (PARITY UNPINNED: filterpy is not importable): the stage runs, names its file
    and prints the reference's line for it (filtering.py:703); the numbers are the oracle restatement's here."""
    import logging
    from pose2sim_amd import filtering
    trial, cfg = _write_trial(work_dir, gold, 0)
    cfg['filtering']['type'] = 'kalman'
    cfg['filtering']['kalman'] = {'trust_ratio': 500, 'smooth': True}
    with caplog.at_level(logging.INFO):
        paths = filtering.filter_all(cfg, engine=OracleFilterEngine())
    assert len(paths) == 1 and paths[0].endswith('_filt_kalman.trc')
    assert ('--> Filter type: Kalman smoother. Measurements trusted 500 times as much as previous data, assuming a constant '
            'acceleration process.') in caplog.text
    from pose2sim_amd import trc
    frames, times, data, markers, header = trc.load_trc(paths[0])
    assert np.isfinite(data).any()


def test_kalman_oracle_tracks_a_smooth_signal():
    """Sanity of the restatement itself (it is all that pins the kernel): a noisy parabola comes back closer to the truth
    with the smoother than with the filter alone, runs shorter than 4 samples and gaps are left alone."""
    rng = np.random.default_rng(2)
    t = np.arange(400) / 60.0
    truth = 1.0 + 0.8 * t - 0.3 * t * t
    col = truth + rng.normal(0, 0.004, t.size)
    col[100:103] = np.nan
    col[103:106] = [0.5, 0.6, 0.7]                      # a run of 3 between gaps
    col[106] = 0.0
    filt = fr.kalman_filter_1d(col, 60, 500, smooth=False)
    smo = fr.kalman_filter_1d(col, 60, 500, smooth=True)
    ok = np.r_[5:100, 110:395]
    assert np.abs(smo[ok] - truth[ok]).mean() < np.abs(col[ok] - truth[ok]).mean()
    assert np.abs(smo[ok] - truth[ok]).mean() <= np.abs(filt[ok] - truth[ok]).mean() * 1.05
    assert np.isnan(smo[100:103]).all() and np.array_equal(smo[103:106], [0.5, 0.6, 0.7]) and smo[106] == 0.0
