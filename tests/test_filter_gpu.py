"""GPU parity of the Butterworth filter kernel and of the trc_evaluate metrics kernel, through the C-ABI
(p2s_butterworth_host, p2s_trc_metrics_host), against the goldens recorded from the reference
(tests/golden/filter_units.npz) and against the oracle on larger seeded inputs.

Bars: filtered samples within 1e-9 relative to max(1, |value|) of scipy.signal.filtfilt (the kernel runs the same
recurrence in the same order without contraction; in practice the difference is 0 or an ulp), untouched samples
bit-identical; bone statistics and accelerations within 1e-12 relative (sums are reduced in a different order), missing
counts and n_valid identical."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-9


@pytest.fixture(scope='module')
def engine():
    import __graft_entry__ as entry
    entry.build_hip()
    from pose2sim_amd.engine import Engine
    eng = Engine(0)
    yield eng
    eng.close()


@pytest.fixture(scope='module')
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, 'filter_units.npz'))


def _close(got, want, what):
    assert np.array_equal(np.isnan(got), np.isnan(want)), f'{what}: NaN pattern'
    ok = ~np.isnan(want)
    if ok.any():
        d = np.abs(got[ok] - want[ok]) / np.maximum(1.0, np.abs(want[ok]))
        assert d.max() <= TOL, f'{what}: {d.max():.3e}'


def test_golden_columns(engine, gold):
    from pose2sim_amd import filtering
    for i in range(int(gold['n_cols'])):
        order, cutoff, rate = (int(v) for v in gold[f'col{i}_prm'])
        col = gold[f'col{i}_in']
        got = filtering.butterworth_filter(col.reshape(-1, 1), order, cutoff, rate, engine)[:, 0]
        _close(got, gold[f'col{i}_out'], f'column {i}')
        untouched = gold[f'col{i}_out'] == col
        assert np.array_equal(got[untouched], col[untouched])


def test_matrix_against_the_oracle(engine):
    """A .trc-sized matrix (20 000 frames x 156 columns = two persons) with gaps of every kind, all orders 2..8."""
    from oracle import filtering_ref as fr
    from pose2sim_amd import filtering
    rng = np.random.default_rng(12)
    F, ncol = 20_000, 156
    t = np.arange(F)[:, None] / 60.0
    data = 1.0 + 0.5 * np.sin(2 * np.pi * (0.5 + rng.random(ncol)) * t) + rng.normal(0, 0.01, (F, ncol))
    data[rng.random((F, ncol)) < 0.002] = np.nan
    data[rng.random((F, ncol)) < 0.001] = 0.0
    for c in range(0, ncol, 7):
        g = int(rng.integers(0, F - 400)); data[g:g + int(rng.integers(1, 400)), c] = np.nan
    data[:, 3] = np.nan
    data[:50, 5] = 0.0
    for order, cutoff in ((2, 6), (4, 6), (8, 10), (16, 12)):
        got = filtering.butterworth_filter(data, order, cutoff, 60, engine)
        want = fr.butterworth_filter(data, order, cutoff, 60)
        _close(got, want, f'order {order}')


def test_edge_shapes(engine):
    from oracle import filtering_ref as fr
    from pose2sim_amd import filtering
    from pose2sim_amd._lib import P2sError
    assert filtering.butterworth_filter(np.zeros((0, 3)), 4, 6, 60, engine).shape == (0, 3)
    assert filtering.butterworth_filter(np.zeros((5, 0)), 4, 6, 60, engine).shape == (5, 0)
    rng = np.random.default_rng(3)
    for F in (1, 9, 10, 11, 65):                       # padlen = 9 for order 4: runs of 10 are the first filtered
        data = rng.normal(1, 0.1, (F, 70))
        _close(filtering.butterworth_filter(data, 4, 6, 60, engine), fr.butterworth_filter(data, 4, 6, 60), f'F={F}')
    with pytest.raises(P2sError):
        engine.butterworth(np.ones((20, 2)), np.ones(12), np.ones(12), np.ones(11))       # order beyond the kernel's 8


def test_filter_all_file(engine, gold):
    """filter_all on the GPU: the reference's header lines and frame / time columns exactly, coordinates within TOL."""
    import shutil
    import tempfile
    from pose2sim_amd import filtering
    root = tempfile.mkdtemp(prefix='p2s_bw_')
    try:
        for i in range(int(gold['n_files'])):
            trial = os.path.join(root, f'trial{i}')
            os.makedirs(os.path.join(trial, 'pose-3d'))
            with open(os.path.join(trial, 'pose-3d', str(gold[f'file{i}_name'])), 'w') as fh:
                fh.write(str(gold[f'file{i}_text']))
            order, cutoff, rate = (int(v) for v in gold[f'file{i}_prm'])
            cfg = {'project': {'project_dir': trial, 'frame_rate': rate, 'frame_range': 'auto'}, 'pose': {'vid_img_extension': 'mp4'},
                   'filtering': {'type': 'butterworth', 'filter': True, 'reject_outliers': False,
                                 'butterworth': {'order': order, 'cut_off_frequency': cutoff}}}
            paths = filtering.filter_all(cfg, engine=engine)
            assert [os.path.basename(p) for p in paths] == [str(gold[f'file{i}_out_name'])]
            got, want = open(paths[0]).read().split('\n'), str(gold[f'file{i}_out_text']).split('\n')
            assert got[:5] == want[:5] and len(got) == len(want)
            for gl, wl in zip(got[5:], want[5:]):
                gf, wf = gl.split('\t'), wl.split('\t')
                assert gf[:2] == wf[:2] and len(gf) == len(wf)
                g = np.array([float(v) if v else np.nan for v in gf[2:]])
                w = np.array([float(v) if v else np.nan for v in wf[2:]])
                _close(g, w, f'file {i}')
    finally:
        shutil.rmtree(root, ignore_errors=True)


def test_trc_evaluate(engine, gold, tmp_path):
    from pose2sim_amd import trc_evaluate
    for i in range(int(gold['n_files'])):
        for tag, name, text in (('raw', gold[f'file{i}_name'], gold[f'file{i}_text']),
                                ('filt', gold[f'file{i}_out_name'], gold[f'file{i}_out_text'])):
            p = tmp_path / f'{i}_{tag}_{name}'
            p.write_text(str(text))
            ev = trc_evaluate.evaluate_single(str(p), engine=engine)
            pre = f'file{i}_{tag}_'
            got = np.array([[b['mean'], b['sd'], b['cv'], b['n_valid']] for b in ev['bone_results']], dtype=np.float64)
            np.testing.assert_allclose(got, gold[pre + 'bones'], rtol=1e-10, atol=0, equal_nan=True)
            assert np.array_equal(got[:, 3], gold[pre + 'bones'][:, 3])
            got = np.array([[s['accel_median'], s['accel_p95'], s['accel_median_si'], s['accel_p95_si'], s['n_valid']] for s in ev['smooth_results']])
            np.testing.assert_allclose(got, gold[pre + 'smooth'], rtol=1e-10, atol=1e-18, equal_nan=True)
            got = np.array([[m['n_total'], m['n_missing'], m['missing_pct']] for m in ev['missing_results']], dtype=np.float64)
            assert np.array_equal(got, gold[pre + 'missing'])
            sm = ev['summary']
            np.testing.assert_allclose([sm['mean_cv'], sm['worst_cv'], sm['mean_accel_p95'], sm['overall_nan_pct'], sm['mean_lr_diff']],
                                       gold[pre + 'summary'], rtol=1e-9, equal_nan=True)
            assert sm['worst_bone'] == str(gold[pre + 'worst'])


def test_metrics_at_trial_size(engine):
    """100 000 frames x 26 markers against the NumPy restatement."""
    from oracle import filtering_ref as fr
    from pose2sim_amd import skeletons, synth, trc_evaluate
    ids, names, _ = skeletons.keypoints('HALPE_26')
    F = 100_000
    xyz = synth.make_points3d(F, 1, len(ids), seed=9)[:, 0] + np.random.default_rng(1).normal(0, 0.003, (F, len(ids), 3))
    xyz[np.random.default_rng(2).random((F, len(ids))) < 0.01] = np.nan
    index = {n: i for i, n in enumerate(names)}
    pairs = np.array([[index[p], index[c]] for p, c, _ in trc_evaluate.HALPE_26_BONES], dtype=np.int32)
    lens, stats, accel, missing = engine.trc_metrics(xyz, pairs)
    rstats, rlens = fr.bone_lengths(xyz, [tuple(p) for p in pairs])
    np.testing.assert_allclose(lens, rlens, rtol=1e-15, equal_nan=True)
    np.testing.assert_allclose(stats[:, :2], np.array([[s[0], s[1]] for s in rstats]), rtol=1e-10)
    assert np.array_equal(stats[:, 2], np.array([s[3] for s in rstats], dtype=np.float64))
    assert np.array_equal(missing, np.array([m[1] for m in fr.missing_data(xyz)]))
    for m in (0, 7, 25):
        want = np.linalg.norm(xyz[2:, m] - 2 * xyz[1:-1, m] + xyz[:-2, m], axis=1)
        np.testing.assert_allclose(accel[m], want, rtol=1e-15, atol=1e-18, equal_nan=True)


# ---- the other column filters (filtering.py:63-160, 474-577) ------------------------------------------------------------
@pytest.fixture(scope='module')
def gold2(golden_dir):
    return np.load(os.path.join(golden_dir, 'filter_units2.npz'))


def test_golden_columns_of_the_other_filters(engine, gold2):
    """48 reference columns through every kernel: Hampel and median bit for bit (they select samples), one-euro, Gaussian
    and Butterworth-on-speed within TOL (the same operations in the same order; in practice 0 or an ulp)."""
    from pose2sim_amd import filtering
    g = gold2
    for i in range(int(g['n_cols'])):
        col = g[f'col{i}_in'].reshape(-1, 1)
        order, cutoff, rate, sigma, ksize = (int(v) for v in g[f'col{i}_prm'])
        mc, beta, dc = g[f'col{i}_euro']
        assert np.array_equal(filtering.hampel_filter(col, engine)[:, 0], g[f'col{i}_hampel'], equal_nan=True), ('hampel', i)
        _close(filtering.one_euro_filter(col, rate, mc, beta, dc, engine)[:, 0], g[f'col{i}_one_euro'], f'one_euro {i}')
        _close(filtering.gaussian_filter(col, sigma, engine)[:, 0], g[f'col{i}_gauss'], f'gauss {i}')
        if f'col{i}_speed' in g.files:
            _close(filtering.butterworth_on_speed_filter(col, order, cutoff, rate, engine)[:, 0], g[f'col{i}_speed'], f'speed {i}')
        if f'col{i}_median' in g.files:
            assert np.array_equal(filtering.median_filter(col, ksize, engine)[:, 0], g[f'col{i}_median']), ('median', i)


def test_other_filters_on_a_matrix_against_the_oracle(engine):
    """20 000 frames x 156 columns with spikes, NaN and zero gaps: every kernel against the NumPy / SciPy restatement."""
    from oracle import filtering_ref as fr
    from pose2sim_amd import filtering
    from pose2sim_amd._lib import P2sError
    rng = np.random.default_rng(21)
    F, ncol = 20_000, 156
    t = np.arange(F)[:, None] / 60.0
    data = 1.0 + 0.5 * np.sin(2 * np.pi * (0.5 + rng.random(ncol)) * t) + rng.normal(0, 0.01, (F, ncol))
    spikes = rng.random((F, ncol)) < 0.01
    data[spikes] += rng.normal(0, 0.5, int(spikes.sum()))
    clean = data.copy()
    data[rng.random((F, ncol)) < 0.002] = np.nan
    data[rng.random((F, ncol)) < 0.001] = 0.0
    for c in range(0, ncol, 7):
        g = int(rng.integers(0, F - 400)); data[g:g + int(rng.integers(1, 400)), c] = np.nan
    data[:, 3] = np.nan
    cols = list(range(0, ncol, 5))                               # the oracle's Python loops: a sample of the columns
    got = filtering.hampel_filter(data, engine)
    for c in cols:
        assert np.array_equal(got[:, c], fr.hampel_filter(data[:, c]), equal_nan=True), c
    assert (got != data)[~np.isnan(data)].sum() > 1000           # it did replace spikes
    got = filtering.one_euro_filter(data, 60, 2.5, 0.9, 1.0, engine)
    for c in cols[:8]:
        _close(got[:, c], fr.one_euro_filter_1d(data[:, c], 60, 2.5, 0.9, 1.0), f'one_euro column {c}')
    for sigma in (1, 3):
        got = filtering.gaussian_filter(data, sigma, engine)
        for c in cols:
            _close(got[:, c], fr.gaussian_filter_1d(data[:, c], sigma), f'gauss {sigma} column {c}')
    got = filtering.butterworth_on_speed_filter(data, 4, 10, 60, engine)
    for c in cols[:8]:
        _close(got[:, c], fr.butterworth_on_speed_filter_1d(data[:, c], 4, 10, 60), f'speed column {c}')
    for k in (3, 9, 15):
        got = filtering.median_filter(clean, k, engine)
        for c in cols:
            assert np.array_equal(got[:, c], fr.median_filter_1d(clean[:, c], k)), (k, c)
    with pytest.raises(P2sError):                                # scipy's answer for NaN depends on its selection algorithm
        filtering.median_filter(data, 5, engine)
    for F2 in (1, 2, 5, 6, 7, 8):                                # windows longer than the data
        small = rng.normal(1, 0.1, (F2, 70))
        assert np.array_equal(filtering.hampel_filter(small, engine), np.stack([fr.hampel_filter(small[:, c]) for c in range(70)], 1))
        _close(filtering.gaussian_filter(small, 2, engine), np.stack([fr.gaussian_filter_1d(small[:, c], 2) for c in range(70)], 1), f'gauss F={F2}')
        _close(filtering.one_euro_filter(small, 30, engine=engine), np.stack([fr.one_euro_filter_1d(small[:, c], 30) for c in range(70)], 1), f'euro F={F2}')


def test_filter_all_with_outlier_rejection_on_the_gpu(engine, gold2):
    """filter_all with the shipped demo configuration (reject_outliers = true + Butterworth) and the other types: the
    reference's header lines and frame / time columns exactly, coordinates within TOL."""
    import shutil
    import tempfile
    from pose2sim_amd import filtering
    g = gold2
    root = tempfile.mkdtemp(prefix='p2s_f2_')
    try:
        for i in range(int(g['n_files'])):
            trial = os.path.join(root, f'trial{i}')
            os.makedirs(os.path.join(trial, 'pose-3d'))
            with open(os.path.join(trial, 'pose-3d', str(g[f'file{i}_name'])), 'w') as fh:
                fh.write(str(g[f'file{i}_text']))
            cfg = {'project': {'project_dir': trial, 'frame_rate': int(g[f'file{i}_rate']), 'frame_range': 'auto'}, 'pose': {'vid_img_extension': 'mp4'},
                   'filtering': {'type': str(g[f'file{i}_type']), 'filter': True, 'reject_outliers': bool(g[f'file{i}_reject']),
                                 'butterworth': {'order': 4, 'cut_off_frequency': 6}, 'butterworth_on_speed': {'order': 4, 'cut_off_frequency': 10},
                                 'one_euro': {'cut_off_frequency': 2.5, 'beta': 0.9, 'd_cut_off_frequency': 1.0},
                                 'gaussian': {'sigma_kernel': 2}, 'median': {'kernel_size': 5}}}
            paths = filtering.filter_all(cfg, engine=engine)
            assert [os.path.basename(p) for p in paths] == [str(g[f'file{i}_out_name'])]
            got, want = open(paths[0]).read().split('\n'), str(g[f'file{i}_out_text']).split('\n')
            assert got[:5] == want[:5] and len(got) == len(want)
            for gl, wl in zip(got[5:], want[5:]):
                gf, wf = gl.split('\t'), wl.split('\t')
                assert gf[:2] == wf[:2] and len(gf) == len(wf)
                a = np.array([float(v) if v else np.nan for v in gf[2:]])
                b = np.array([float(v) if v else np.nan for v in wf[2:]])
                _close(a, b, f'file {i} ({cfg["filtering"]["type"]})')
    finally:
        shutil.rmtree(root, ignore_errors=True)


def test_kalman_kernel_against_the_restatement(engine):
    """p2s_kalman_kernel against oracle/filtering_ref.kalman_filter_1d -- PARITY UNPINNED: the reference takes the filter
    and the smoother from filterpy, which is not importable here, so no golden exists; both sides follow filterpy's
    published algorithm (predict, Joseph-form update, rts_smoother).  Runs of every length around the 4-sample limit,
    NaN and zero gaps, a column that is all NaN, filter and smoother, two trust ratios; 1e-9 relative."""
    from oracle import filtering_ref as fr
    from pose2sim_amd import filtering
    rng = np.random.default_rng(33)
    F, ncol = 3_000, 70
    t = np.arange(F)[:, None] / 60.0
    data = 1.0 + 0.5 * np.sin(2 * np.pi * (0.3 + rng.random(ncol)) * t) + rng.normal(0, 0.005, (F, ncol))
    data[rng.random((F, ncol)) < 0.01] = np.nan
    data[rng.random((F, ncol)) < 0.004] = 0.0
    data[:, 5] = np.nan
    data[10:13, 6] = np.nan; data[16, 6] = np.nan; data[21, 6] = 0.0           # runs of 3, 4 and more
    for trust, smooth in ((500, True), (500, False), (20, True)):
        got = filtering.kalman_filter(data, 60, trust, smooth, engine)
        for c in range(0, ncol, 3):
            _close(got[:, c], fr.kalman_filter_1d(data[:, c], 60, trust, smooth), f'kalman trust {trust} smooth {smooth} column {c}')
        assert np.array_equal(got[:, 5], data[:, 5], equal_nan=True)
    for F2 in (1, 3, 4, 5):
        small = rng.normal(1, 0.1, (F2, 9))
        want = np.stack([fr.kalman_filter_1d(small[:, c], 30, 500, True) for c in range(9)], 1)
        _close(filtering.kalman_filter(small, 30, 500, True, engine), want, f'kalman F={F2}')


def test_random_filter_cases():
    """A short run of tests/sweeps/fuzz_filters.py inside the suite: 25 random (length 1..600, columns, gaps, parameters)
    cases through the seven column filters against oracle/filtering_ref.py."""
    import importlib.util
    import __graft_entry__ as entry
    entry.build_hip()
    spec = importlib.util.spec_from_file_location('fuzz_filters', os.path.join(os.path.dirname(__file__), 'sweeps', 'fuzz_filters.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, worst = mod.run(25, 7, verbose=False)
    assert bad == 0, worst
