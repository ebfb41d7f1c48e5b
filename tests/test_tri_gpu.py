"""GPU parity of the fused triangulation kernel, through the C-ABI.

Bars (BASELINE.json north_star): 3D within 1e-4 mm = 1e-7 m of the reference, identical
inlier-camera selections (n_excl, excluded-camera mask, NaN pattern) -- checked against
  * the fixtures recorded from the reference (tests/golden/tri_units.npz),
  * the CPU oracle on fresh seeded inputs,
  * size-independent properties at BASELINE config sizes.
The reprojection error leaves the kernel as float32: tolerance 2e-6 relative to max(1 px, err) -- float32 rounding
(6e-8) plus what a 1e-9 m difference in Q moves a pixel error by (~3.5e-7 px at f/z ~ 350 px/m).
Every test runs on both triangulation paths: the one-launch kernel with the in-wave subset search (the default where
it applies: pinhole, no L/R swap, <= 16 cameras) and the streaming + work-list search pair.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL_Q = 1e-7        # metres, absolute, for every unit within FAR_M of the origin
TOL_Q_REL = 1e-9    # relative to |Q| beyond FAR_M (two-camera units that land kilometres away)
FAR_M = 100.0
TOL_E = 2e-6        # px, relative to max(1, err)


@pytest.fixture(scope='module', params=['auto', 'worklist', 'onetile', 'twotiles', 'noscreen'])
def engine(request):
    """auto: the pooled kernel (persistent waves, fp32 screen + fp64 evaluation of the survivors) where it applies, else the
    one-launch kernel, else the work-list pair; noscreen: the pooled kernel with every candidate sent to the fp64
    evaluation; twotiles / onetile: the one-launch kernel of round 2; worklist: the pair everywhere."""
    import __graft_entry__ as entry
    entry.build_hip()
    from pose2sim_amd.engine import Engine
    eng = Engine(0)
    eng.set_tuning(Engine.TUNE_TRI_PATH, {'auto': Engine.TRI_PATH_AUTO, 'worklist': Engine.TRI_PATH_WORKLIST,
                                          'onetile': Engine.TRI_PATH_ONE_TILE, 'twotiles': Engine.TRI_PATH_TWO_TILES,
                                          'noscreen': Engine.TRI_PATH_POOLED}[request.param])
    if request.param == 'noscreen':
        eng.set_tuning(Engine.TUNE_SCREEN, 0)
    yield eng
    eng.close()


def _cal_from_group(g):
    from pose2sim_amd import cvmath
    C = int(g['C'])
    return {'K': [g['K'][c] for c in range(C)], 'dist': [g['dist'][c] for c in range(C)],
            'R': [g['R'][c] for c in range(C)], 'R_mat': [cvmath.rodrigues(g['R'][c]) for c in range(C)],
            'T': [g['T'][c] for c in range(C)], 'optim_K': [g['optim_K'][c] for c in range(C)]}


def _compare(Q, err, nex, mask, Qr, er, nr, mr, what=''):
    Q = Q.reshape(-1, 3); err = err.reshape(-1); nex = nex.reshape(-1); mask = mask.reshape(-1)
    Qr = Qr.reshape(-1, 3); er = np.asarray(er, dtype=np.float64).reshape(-1)
    nr = np.asarray(nr).reshape(-1); mr = np.asarray(mr).reshape(-1)
    bad = np.flatnonzero(np.isnan(err) != np.isnan(er))
    assert bad.size == 0, f'{what}: NaN pattern differs at units {bad[:8]}'
    bad = np.flatnonzero(nex.astype(np.int64) != nr.astype(np.int64))
    assert bad.size == 0, f'{what}: nb_cams_excluded differs at {bad[:8]}: {nex[bad[:8]]} vs {nr[bad[:8]]}'
    bad = np.flatnonzero(mask.astype(np.uint32) != mr.astype(np.uint32))
    assert bad.size == 0, f'{what}: excluded-camera mask differs at {bad[:8]}: {mask[bad[:8]]} vs {mr[bad[:8]]}'
    ok = ~np.isnan(er)
    assert np.isnan(Q[~ok]).all()
    if ok.any():
        # 1e-7 m ABSOLUTE for every unit within 100 m of the origin; 1e-9 relative beyond.  The relative part is needed,
        # and only by accepted garbage: with two cameras, noise can make two rays nearly parallel and the "point" lands
        # 1.3 - 15 km away with a pixel error under the threshold (units 387186, 1417924, 1484274 of the two-camera test,
        # measured |dQ| 1.6e-7 - 7.9e-7 m = 1e-11 - 6e-10 relative, on both kernel paths alike,
        # tests/sweeps/c2_far_units.py); at that depth the reference's own SVD is no better determined.  The count of
        # units that needed the relative bar is printed, so that a growing count is visible.
        dist = np.linalg.norm(Qr[ok], axis=1)
        dabs = np.abs(Q[ok] - Qr[ok]).max(axis=1)
        near = dist <= FAR_M
        n_far = int((~near).sum())
        if n_far:
            print(f'{what}: {n_far} unit(s) beyond {FAR_M:.0f} m judged by the relative bar '
                  f'(max |dQ|/|Q| = {(dabs[~near] / dist[~near]).max():.2e})')
            assert (dabs[~near] <= TOL_Q_REL * dist[~near]).all(), \
                f'{what}: max |dQ|/|Q| = {(dabs[~near] / dist[~near]).max():.3e} beyond {FAR_M:.0f} m'
        dq = float(dabs[near].max()) if near.any() else 0.0
        assert dq <= TOL_Q, f'{what}: max |dQ| = {dq:.3e} m (absolute, units within {FAR_M:.0f} m)'
        de = np.abs(err[ok].astype(np.float64) - er[ok]) / np.maximum(1.0, np.abs(er[ok]))
        assert de.max() <= TOL_E, f'{what}: error differs by {de.max():.3e}'
        return dq
    return 0.0


def test_golden_units(engine, golden_dir):
    """Every recorded reference unit (2.4k units, C in 2..16, swap / undistort / zero likelihoods)."""
    from pose2sim_amd import skeletons
    _, _, swap = skeletons.keypoints('HALPE_26')
    z = np.load(os.path.join(golden_dir, 'tri_units.npz'))
    worst = 0.0
    for i in range(int(z['n_groups'])):
        g = {k[len(f'g{i}_'):]: z[k] for k in z.files if k.startswith(f'g{i}_')}
        C = int(g['C'])
        engine.set_calibration([g['P'][c] for c in range(C)], _cal_from_group(g))
        prm = engine.tri_params(float(g['thr']), float(g['lik_thr']), int(g['min_cams']),
                                bool(g['undistort']), bool(g['lr_swap']))
        xyl = g['raw_xyl']                     # [F][1][C][K][3] float32
        Q, err, nex, mask = engine.triangulate(xyl, prm, swap)
        worst = max(worst, _compare(Q, err, nex, mask, g['Q'], g['err'], g['n_excl'], g['mask'], f'group {i}'))
    print(f'golden: worst |dQ| = {worst:.3e} m')


@pytest.mark.parametrize('C,min_cams,lr_swap,undistort,dtype64', [
    (4, 2, False, False, False),
    (8, 2, False, False, False),
    (8, 3, True, False, False),
    (5, 2, False, True, False),
    (6, 2, True, True, False),
    (8, 2, False, False, True),
    (3, 2, True, False, True),
    (4, 2, False, False, True),
    (2, 2, False, False, True),
    (13, 4, False, False, False),
    (16, 2, False, False, False),
])
def test_against_oracle(engine, C, min_cams, lr_swap, undistort, dtype64):
    from oracle import triangulation_ref as tr
    from pose2sim_amd import skeletons, synth
    _, _, swap = skeletons.keypoints('HALPE_26')
    F = 40
    wl = synth.make_config(F, C, 26, 1, seed=11 + C, undistort=undistort, lr_swap=lr_swap, swap_idx=swap,
                           p_lowlik=0.08, p_outlier=0.06, p_missing_cam=0.02)
    xyl = wl['xyl']
    if dtype64:     # values that are NOT float32-representable must take the float64 path unchanged
        rng = np.random.default_rng(5)
        xyl = xyl.astype(np.float64) + rng.uniform(-1e-4, 1e-4, xyl.shape) * (xyl.astype(np.float64) != 0)
    cams = wl['cams']
    engine.set_calibration(wl['P'], cams)
    prm = engine.tri_params(12.0, 0.3, min_cams, undistort, lr_swap)
    Q, err, nex, mask = engine.triangulate(xyl, prm, swap)
    Qr, er, nr, mr = tr.triangulate_batch(xyl, wl['P'], cams, swap, 0.3, 12.0, min_cams, lr_swap, undistort)
    _compare(Q, err, nex, mask, Qr, er, nr, mr, f'C={C}')


def test_edge_cases(engine):
    """Empty input, all-missing units, fewer cameras than min_cameras, ragged last tile, K = 1."""
    from oracle import triangulation_ref as tr
    from pose2sim_amd import synth
    wl = synth.make_config(3, 4, 5, 1, seed=3)
    engine.set_calibration(wl['P'])
    prm = engine.tri_params(15.0, 0.3, 2)
    # empty
    Q, err, nex, mask = engine.triangulate(np.zeros((0, 4, 5, 3), np.float32), prm)
    assert Q.shape == (0, 5, 3) and err.shape == (0, 5)
    # everything missing / below threshold
    x = wl['xyl'].copy()
    x[0] = np.nan
    x[1, :, :, :, 2] = 0.1
    Q, err, nex, mask = engine.triangulate(x, prm)
    assert np.isnan(Q[0]).all() and np.isnan(err[0]).all() and (nex[0] == 4).all() and (mask[0] == 0xF).all()
    assert np.isnan(Q[1]).all() and (nex[1] == 4).all()
    Qr, er, nr, mr = tr.triangulate_batch(x, wl['P'], None, list(range(5)), 0.3, 15.0, 2)
    _compare(Q, err, nex, mask, Qr, er, nr, mr, 'missing')
    # min_cameras above the camera count: no level runs (triangulation.py:408, 595-596)
    prm5 = engine.tri_params(15.0, 0.3, 5)
    Q, err, nex, mask = engine.triangulate(wl['xyl'], prm5)
    assert np.isnan(Q).all() and (nex == 4).all() and (mask == 0xF).all()
    # ragged tiles: every n_blocks from 1 to 2 tiles + 1, K = 1
    wl1 = synth.make_config(300, 4, 1, 1, seed=4, p_outlier=0.1)
    engine.set_calibration(wl1['P'])
    geo = engine.tri_geometry(1)
    Qa, ea, na, ma = engine.triangulate(wl1['xyl'], prm)
    for nb in (1, 2, geo['blocks_per_tile'] - 1, geo['blocks_per_tile'] + 1, 299):
        nb = max(1, min(nb, 300))
        Q, err, nex, mask = engine.triangulate(wl1['xyl'][:nb], prm)
        assert np.array_equal(Q, Qa[:nb], equal_nan=True) and np.array_equal(mask, ma[:nb])
    Qr, er, nr, mr = tr.triangulate_batch(wl1['xyl'][:60], wl1['P'], None, [0], 0.3, 15.0, 2)
    _compare(Qa[:60], ea[:60], na[:60], ma[:60], Qr, er, nr, mr, 'K=1')


def test_non_finite_coordinates_with_a_valid_likelihood(engine):
    """A camera whose coordinates are inf or NaN while its likelihood passes the threshold.  The reference keeps it among
    the valid cameras: level 0 cannot be solved (every reprojection is NaN -> euclidean_distance's all-NaN rule -> inf), every
    subset that keeps the camera stays at inf, and np.nanargmin (triangulation.py:500-503) finds the subset without it one
    level later.  The kernels take such a camera as a missing detection at level 0 (classify_and_accumulate): the same point,
    error, exclusion count and excluded-camera set, on every kernel path.  Units left with fewer finite cameras than
    min_cameras are not triangulated either way; there the reference reports the first cameras of its last level and
    the kernels every camera."""
    from oracle import triangulation_ref as tr
    from pose2sim_amd import synth
    for C, min_cams in ((6, 2), (8, 3), (12, 4)):
        wl = synth.make_config(40, C, 7, 1, seed=70 + C, p_outlier=0.0, p_lowlik=0.0, p_missing_cam=0.0)
        x = wl['xyl'].copy()
        x[0:8, 0, 1, :, 0] = np.inf                      # x of camera 1
        x[8:16, 0, 2, :, 1] = -np.inf                    # y of camera 2
        x[16:24, 0, 0, :, 0] = np.nan                    # NaN coordinate, likelihood intact
        x[24:32, 0, 3, :, 0] = np.inf
        x[24:32, 0, 4, :, 1] = np.nan                    # two such cameras: the reference goes to level 2
        x[32:36, 0, :, :, 0] = np.inf                    # every camera
        engine.set_calibration(wl['P'])
        prm = engine.tri_params(15.0, 0.3, min_cams)
        with np.errstate(all='ignore'):
            Qr, er, nr, mr = tr.triangulate_batch(x, wl['P'], None, list(range(7)), 0.3, 15.0, min_cams)
        Q, err, nex, mask = engine.triangulate(x, prm)
        rest = np.r_[0:32, 36:40]
        _compare(Q[rest], err[rest], nex[rest], mask[rest], np.asarray(Qr)[rest], np.asarray(er)[rest], np.asarray(nr)[rest],
                 np.asarray(mr)[rest], f'non-finite coordinates, C={C}')
        assert not np.isnan(err[0:32]).any()
        assert (nex[0:24] == 1).all() and (nex[24:32] == 2).all() and (mask[24:32] == 0b11000).all()
        assert np.isnan(np.asarray(er)[32:36]).all() and np.isnan(err[32:36]).all() and np.isnan(Q[32:36]).all()
        assert (nex[32:36] == C).all() and (mask[32:36] == (1 << C) - 1).all()


def test_bad_arguments(engine):
    from pose2sim_amd._lib import P2sError
    from pose2sim_amd import synth
    wl = synth.make_config(2, 4, 5, 1, seed=3)
    engine.set_calibration(wl['P'])
    with pytest.raises(P2sError):
        engine.triangulate(wl['xyl'], engine.tri_params(15.0, 0.3, 0))          # min_cameras < 1
    with pytest.raises(P2sError):
        engine.triangulate(wl['xyl'], engine.tri_params(15.0, 0.3, 2, undistort=True))   # no K/dist given
    with pytest.raises(P2sError):
        engine.triangulate(wl['xyl'], engine.tri_params(15.0, 0.3, 2, lr_swap=True))     # no swap_idx
    with pytest.raises(P2sError):
        engine.triangulate(wl['xyl'][:, :, :3], engine.tri_params(15.0, 0.3, 2))        # camera count mismatch


def test_full_size_properties(engine):
    """BASELINE config 2 size (8 cams x 26 kpts x 100k frames): properties that need no oracle run --
    bit-identical repeat, frame-permutation equivariance, clean-data recovery of the generating 3D
    points -- plus an oracle spot check on a random sample of frames."""
    from oracle import triangulation_ref as tr
    from pose2sim_amd import skeletons, synth
    _, _, swap = skeletons.keypoints('HALPE_26')
    F, C, K = 100_000, 8, 26
    wl = synth.make_config(F, C, K, 1, seed=2)
    engine.set_calibration(wl['P'])
    prm = engine.tri_params(15.0, 0.3, 2)
    xyl = wl['xyl']
    Q, err, nex, mask = engine.triangulate(xyl, prm)
    Q2, err2, nex2, mask2 = engine.triangulate(xyl, prm)
    assert np.array_equal(Q, Q2, equal_nan=True) and np.array_equal(err, err2, equal_nan=True)
    assert np.array_equal(nex, nex2) and np.array_equal(mask, mask2)
    perm = np.random.default_rng(0).permutation(F)
    Qp, errp, nexp, maskp = engine.triangulate(xyl[perm], prm)
    assert np.array_equal(Qp, Q[perm], equal_nan=True) and np.array_equal(maskp, mask[perm])
    assert np.array_equal(nexp, nex[perm]) and np.array_equal(errp, err[perm], equal_nan=True)
    # popcount(mask) == n_excl without L/R swap and zero likelihoods
    pop = np.unpackbits(mask.view(np.uint8).reshape(-1, 4), axis=1).sum(axis=1).reshape(mask.shape)
    assert np.array_equal(pop, nex)
    # accepted units have error <= threshold and sit near the generating point (noise 1.5 px ~ cm)
    ok = ~np.isnan(err)
    assert ok.mean() > 0.99 and (err[ok] <= 15.0).all()
    dist = np.linalg.norm(Q.reshape(F, K, 3) - wl['Q3d'].reshape(F, K, 3), axis=-1)[ok.reshape(F, K)]
    assert np.median(dist) < 0.01 and np.percentile(dist, 99) < 0.05
    # oracle spot check
    sel = np.sort(np.random.default_rng(1).choice(F, 60, replace=False))
    Qr, er, nr, mr = tr.triangulate_batch(xyl[sel], wl['P'], None, swap, 0.3, 15.0, 2)
    _compare(Q[sel], err[sel], nex[sel], mask[sel], Qr, er, nr, mr, 'cfg2 sample')


def test_noise_free_recovers_ground_truth(engine):
    from pose2sim_amd import synth
    wl = synth.make_config(500, 6, 26, 1, seed=9, noise_px=0.0, p_lowlik=0.0, p_outlier=0.0, p_missing_cam=0.0)
    engine.set_calibration(wl['P'])
    # float64 observations carry the exact projections
    cams = wl['cams']
    from pose2sim_amd import cvmath
    xyl = wl['xyl'].astype(np.float64)
    for c in range(6):
        uv = cvmath.project_points(wl['Q3d'].reshape(-1, 3), cams['R_mat'][c], cams['T'][c], cams['K'][c], np.zeros(4))
        xyl[:, 0, c, :, :2] = uv.reshape(500, 26, 2)
    Q, err, nex, mask = engine.triangulate(xyl, engine.tri_params(15.0, 0.3, 2))
    assert np.abs(Q.reshape(-1, 3) - wl['Q3d'].reshape(-1, 3)).max() < 1e-9
    assert (nex == 0).all() and err.max() < 1e-6


@pytest.mark.parametrize('C,F', [(4, 200_000), (12, 190_000)])
def test_multi_chunk_calls_equal_the_whole_oracle_run(engine, C, F):
    """More than 2^22 units in one call: several (level-0, search) kernel pairs with alternating work lists and
    the search of chunk i running beside the level-0 pass of chunk i+1 (direct kernel for C <= 8, tiled above).
    EVERY unit is checked against the C oracle, and the call equals the same data given chunk by chunk."""
    from oracle import tri_oracle
    from pose2sim_amd import skeletons, synth
    _, _, swap = skeletons.keypoints('HALPE_26')
    K = 26
    wl = synth.make_config(F, C, K, 1, seed=40 + C, p_outlier=0.04)
    engine.set_calibration(wl['P'])
    prm = engine.tri_params(15.0, 0.3, 2)
    xyl = wl['xyl']
    assert F * K > 1 << 22
    Q, err, nex, mask = engine.triangulate(xyl, prm)
    cut = (1 << 22) // K // 7 * 7 + 3                        # not on a chunk boundary
    for lo, hi in ((0, cut), (cut, F)):
        Qp, ep, np_, mp = engine.triangulate(xyl[lo:hi], prm)
        assert np.array_equal(Qp, Q[lo:hi], equal_nan=True) and np.array_equal(ep, err[lo:hi], equal_nan=True)
        assert np.array_equal(np_, nex[lo:hi]) and np.array_equal(mp, mask[lo:hi])
    threads = min(64, len(os.sched_getaffinity(0)))
    Qr, er, nr, mr = tri_oracle.triangulate_batch(xyl.astype(np.float64), wl['P'], None, swap, 0.3, 15.0, 2, threads=threads)
    _compare(Q, err, nex, mask, Qr, er, nr, mr, f'{F * K} units, C={C}')


@pytest.mark.parametrize('C', [2, 3])
def test_two_camera_geometry_every_unit_against_the_oracle(engine, C):
    """Two cameras left = one weakly observed direction in the normal matrix: the case that needs the
    second-order stop rule and the refinement step of smallest_eigvec.  1.5 M units, all checked."""
    from oracle import tri_oracle
    from pose2sim_amd import skeletons, synth
    _, _, swap = skeletons.keypoints('HALPE_26')
    F, K = 60_000, 26
    wl = synth.make_config(F, C, K, 1, seed=5 + C)
    engine.set_calibration(wl['P'])
    Q, err, nex, mask = engine.triangulate(wl['xyl'], engine.tri_params(15.0, 0.3, 2))
    threads = min(64, len(os.sched_getaffinity(0)))
    Qr, er, nr, mr = tri_oracle.triangulate_batch(wl['xyl'].astype(np.float64), wl['P'], None, swap, 0.3, 15.0, 2, threads=threads)
    dq = _compare(Q, err, nex, mask, Qr, er, nr, mr, f'C={C}')
    assert dq <= 2e-8, dq


@pytest.mark.parametrize('name,F,C,min_cams,undistort,lr_swap,f64,gen', [
    ('c32 undistort swap', 200, 32, 2, True, True, False, dict(p_outlier=0.01, p_lowlik=0.03)),
    ('c32 min28', 1000, 32, 28, False, False, False, dict(p_outlier=0.03)),
    ('c24 min20 swap', 1000, 24, 20, False, True, False, {}),
    ('c17 f64 undistort', 1500, 17, 12, True, False, True, {}),
    ('c9', 10_000, 9, 2, False, False, False, {}),
])
def test_many_cameras_every_unit_against_the_oracle(engine, name, F, C, min_cams, undistort, lr_swap, f64, gen):
    """Camera counts above the direct kernel's 8 (tiled level-0 kernel, search groups of up to 64 lanes and several
    rounds per level, swap and undistortion at C = 32 as in BASELINE configs[4])."""
    from oracle import tri_oracle
    from pose2sim_amd import skeletons, synth
    _, _, swap = skeletons.keypoints('HALPE_26')
    wl = synth.make_config(F, C, 26, 1, seed=100 + C, undistort=undistort, lr_swap=lr_swap, swap_idx=swap, **gen)
    x64 = wl['xyl'].astype(np.float64) + (1e-9 if f64 else 0.0)
    engine.set_calibration(wl['P'], wl['cams'] if undistort else None)
    prm = engine.tri_params(15.0, 0.3, min_cams, undistort, lr_swap)
    Q, err, nex, mask = engine.triangulate(x64 if f64 else wl['xyl'], prm, swap if lr_swap else None)
    threads = min(64, len(os.sched_getaffinity(0)))
    Qr, er, nr, mr = tri_oracle.triangulate_batch(x64, wl['P'], wl['cams'] if undistort else None, swap, 0.3, 15.0, min_cams,
                                                  lr_swap, undistort, threads=threads)
    _compare(Q, err, nex, mask, Qr, er, nr, mr, name)


def test_cfg4_shape_every_unit_against_the_oracle(engine):
    """BASELINE configs[3]'s shape: 16 cameras x COCO_133 (131 keypoints in skeleton order), min_cameras = 3.
    K = 131 makes the unit -> (block, keypoint) split, the 64-unit tiles and the 16-byte alignment of the wave-wide
    result stores fall differently from K = 26; the frame counts give a last tile that is not full and (through a
    second call on a slice that starts at an odd block) result pointers that are not 16-byte aligned."""
    from oracle import tri_oracle
    from pose2sim_amd import skeletons, synth
    ids, names, swap = skeletons.keypoints('COCO_133')
    K = len(ids)
    assert K == 131
    F, C = 9_001, 16
    wl = synth.make_config(F, C, K, 1, seed=404, p_outlier=0.03, p_lowlik=0.05, p_missing_cam=0.01)
    engine.set_calibration(wl['P'])
    prm = engine.tri_params(15.0, 0.3, 3)
    Q, err, nex, mask = engine.triangulate(wl['xyl'], prm)
    threads = min(64, len(os.sched_getaffinity(0)))
    Qr, er, nr, mr = tri_oracle.triangulate_batch(wl['xyl'].astype(np.float64), wl['P'], None, swap, 0.3, 15.0, 3, threads=threads)
    _compare(Q, err, nex, mask, Qr, er, nr, mr, 'C=16 K=131 min_cams=3')
    # odd first block, odd block count: the same numbers through the unaligned store path
    Qs, es, ns, ms = engine.triangulate(wl['xyl'][1:778], prm)
    assert np.array_equal(Qs, Q[1:778], equal_nan=True) and np.array_equal(es, err[1:778], equal_nan=True)
    assert np.array_equal(ns, nex[1:778]) and np.array_equal(ms, mask[1:778])


@pytest.mark.parametrize('singles_pct', [0, 8, 50, 100])
@pytest.mark.parametrize('C,p_outlier,F', [(8, 0.30, 1_237), (6, 0.20, 2_001), (8, 0.03, 5_003)])
def test_pooled_search_slots_and_tile_pairing(C, p_outlier, F, singles_pct):
    """The one-launch kernel's bookkeeping, every unit against the C oracle: far more searching units per wave than
    its 32 slots hold (30 % gross outliers: further rounds that read the observations again), tile counts that are not
    a multiple of the 8 XCD ranges or of the pair size, and every split of a range into paired and single tiles."""
    import __graft_entry__ as entry
    entry.build_hip()
    from oracle import tri_oracle
    from pose2sim_amd.engine import Engine
    from pose2sim_amd import synth
    wl = synth.make_config(F, C, 26, 1, seed=500 + C, p_outlier=p_outlier, p_lowlik=0.05, p_missing_cam=0.01)
    eng = Engine(0)
    try:
        eng.set_tuning(Engine.TUNE_POOL_SINGLES_PCT, singles_pct)
        eng.set_calibration(wl['P'])
        prm = eng.tri_params(15.0, 0.3, 2)
        Q, err, nex, mask = eng.triangulate(wl['xyl'], prm)
    finally:
        eng.close()
    threads = min(64, len(os.sched_getaffinity(0)))
    Qr, er, nr, mr = tri_oracle.triangulate_batch(wl['xyl'].astype(np.float64), wl['P'], None, list(range(26)), 0.3, 15.0, 2, threads=threads)
    _compare(Q, err, nex, mask, Qr, er, nr, mr, f'C={C} outliers {p_outlier} singles {singles_pct}%')


@pytest.mark.parametrize('name,F,C,K,min_cams,f64,gen', [
    ('cfg2 mix', 6_001, 8, 26, 2, False, {}),
    ('heavy outliers', 1_500, 8, 26, 2, False, dict(p_outlier=0.30, p_lowlik=0.05, p_missing_cam=0.01)),
    ('min_cams 4, many low-confidence', 3_000, 8, 26, 4, False, dict(p_outlier=0.08, p_lowlik=0.25)),
    ('6 cameras', 4_000, 6, 26, 2, False, dict(p_outlier=0.10)),
    ('4 cameras, weak geometry', 6_000, 4, 26, 2, False, dict(p_outlier=0.08, p_lowlik=0.10)),
    ('3 cameras', 6_000, 3, 26, 2, False, dict(p_outlier=0.10)),
    ('5 cameras x 131 keypoints', 800, 5, 131, 3, False, dict(p_outlier=0.10, p_missing_cam=0.05)),
])
@pytest.mark.parametrize('tiles', [2, 3, 4, 5, 6])
def test_screen_changes_nothing(name, F, C, K, min_cams, f64, gen, tiles):
    """The pooled kernel's fp32 screen decides which camera subsets reach the fp64 evaluation and nothing else: with the
    screen off (every candidate evaluated in fp64) every output bit is the same, on seven workloads and for 2, 3 and 4
    tiles pooled per wave -- and fewer subsets were evaluated in fp64 with it on.  Every unit against the C oracle as well."""
    import __graft_entry__ as entry
    entry.build_hip()
    from oracle import tri_oracle
    from pose2sim_amd.engine import Engine
    from pose2sim_amd import synth
    wl = synth.make_config(F, C, K, 1, seed=900 + C + K, **gen)
    outs, evals = {}, {}
    for screen in (1, 0):
        eng = Engine(0)
        try:
            eng.set_tuning(Engine.TUNE_TRI_PATH, Engine.TRI_PATH_POOLED)
            eng.set_tuning(Engine.TUNE_POOL_TILES, tiles)
            eng.set_tuning(Engine.TUNE_SCREEN, screen)
            eng.set_calibration(wl['P'])
            prm = eng.tri_params(15.0, 0.3, min_cams)
            eng.tri_stats(reset=True)
            outs[screen] = eng.triangulate(wl['xyl'], prm)
            evals[screen] = eng.tri_stats(reset=True)
        finally:
            eng.close()
    for a, b, what in zip(outs[1], outs[0], ('Q', 'err', 'n_excl', 'mask')):
        assert a.tobytes() == b.tobytes(), f'{name}: {what} differs between screen on and off'
    assert evals[1]['screened_subsets'] == evals[0]['screened_subsets'] == evals[0]['subsets_evaluated']
    if evals[0]['subsets_evaluated'] > 1000:
        assert evals[1]['subsets_evaluated'] < 0.5 * evals[0]['subsets_evaluated'], f'{name}: the screen let {evals[1]} of {evals[0]} through'
    threads = min(64, len(os.sched_getaffinity(0)))
    Qr, er, nr, mr = tri_oracle.triangulate_batch(wl['xyl'].astype(np.float64), wl['P'], None, list(range(K)), 0.3, 15.0, min_cams, threads=threads)
    _compare(*outs[1], Qr, er, nr, mr, f'{name}, {tiles} tiles')
    print(f'{name}, {tiles} tiles: {evals[1]["subsets_evaluated"]} of {evals[0]["subsets_evaluated"]} subsets evaluated in fp64')


@pytest.mark.parametrize('C,min_cams,lr_swap,undistort,deep_min', [(32, 24, False, False, 16384), (32, 25, True, True, 16384),
                                                                   (24, 18, True, False, 100), (20, 14, False, True, 100), (12, 3, False, False, 10)])
def test_exact_pruning_changes_nothing(C, min_cams, lr_swap, undistort, deep_min):
    """The long levels (64 lanes per unit in the search kernel, and the deep rounds) drop candidates whose partial error
    sum over the most suspicious cameras already exceeds what can still matter: the outputs are bit for bit those of
    the run that evaluates every camera of every candidate, and the counters show that cameras were in fact skipped."""
    import __graft_entry__ as entry
    entry.build_hip()
    from pose2sim_amd import skeletons, synth
    from pose2sim_amd.engine import Engine
    _, _, swap = skeletons.keypoints('HALPE_26')
    wl = synth.make_config(300, C, 26, 1, seed=700 + C, undistort=undistort, lr_swap=lr_swap, swap_idx=swap,
                           p_outlier=0.12, p_lowlik=0.04, p_missing_cam=0.02)
    eng = Engine(0)
    try:
        eng.set_tuning(Engine.TUNE_TRI_PATH, Engine.TRI_PATH_WORKLIST)
        eng.set_tuning(Engine.TUNE_DEEP_MIN_SUBSETS, deep_min)
        eng.set_calibration(wl['P'], wl['cams'] if undistort else None)
        prm = eng.tri_params(6.0, 0.3, min_cams, undistort, lr_swap)
        eng.set_tuning(Engine.TUNE_DEEP_PRUNE, 0)
        ref = eng.triangulate(wl['xyl'], prm, swap if lr_swap else None)
        st0 = eng.tri_stats(reset=True)
        eng.set_tuning(Engine.TUNE_DEEP_PRUNE, 1)
        got = eng.triangulate(wl['xyl'], prm, swap if lr_swap else None)
        st1 = eng.tri_stats(reset=True)
    finally:
        eng.close()
    for a, b in zip(ref, got):
        assert np.array_equal(a, b, equal_nan=True)
    assert st0['pruned_subsets'] == 0 and st0['subsets_evaluated'] == st1['subsets_evaluated']
    assert st1['pruned_subsets'] > 0
    assert st1['pruned_camera_errors'] < C * st1['pruned_subsets']           # cameras were skipped


def test_unaligned_device_outputs(engine):
    """p2s_triangulate_device with result pointers that are only element-aligned (the packed result buffer of a
    caller need not start the float32 / uint32 / uint8 arrays on 16 bytes): same numbers as the aligned call.
    Device memory comes from the HIP runtime directly (hipMalloc through ctypes), no torch in this process."""
    import ctypes as C
    from pose2sim_amd import synth
    from pose2sim_amd.engine import P2S_F32
    hip = C.CDLL('libamdhip64.so')
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    F, Cn, K = 333, 8, 26
    wl = synth.make_config(F, Cn, K, 1, seed=77, p_outlier=0.05)
    engine.set_calibration(wl['P'])
    prm = engine.tri_params(15.0, 0.3, 2)
    Q, err, nex, mask = engine.triangulate(wl['xyl'], prm)
    n = F * K
    xyl = np.ascontiguousarray(wl['xyl'])
    d_x, d_o = C.c_void_p(), C.c_void_p()
    total = n * 33 + 64
    assert hip.hipMalloc(C.byref(d_x), xyl.nbytes) == 0 and hip.hipMalloc(C.byref(d_o), total) == 0
    try:
        assert hip.hipMemcpy(d_x, xyl.ctypes.data_as(C.c_void_p), xyl.nbytes, 1) == 0      # host -> device
        assert hip.hipMemset(d_o, 0, total) == 0
        base = d_o.value + 8                                       # Q 8-byte aligned only
        pe, pm, px = base + n * 24, base + n * 28, base + n * 32 + 3
        engine.triangulate_device(F, K, P2S_F32, d_x.value, None, prm, base, pe, px, pm)
        engine.synchronize()
        h = np.empty(total, dtype=np.uint8)
        assert hip.hipMemcpy(h.ctypes.data_as(C.c_void_p), d_o, total, 2) == 0              # device -> host
    finally:
        hip.hipFree(d_x); hip.hipFree(d_o)
    o = 8
    Qd = h[o:o + n * 24].copy().view(np.float64).reshape(F, K, 3); o += n * 24
    ed = h[o:o + n * 4].copy().view(np.float32).reshape(F, K); o += n * 4
    md = h[o:o + n * 4].copy().view(np.uint32).reshape(F, K); o += n * 4
    xd = h[o + 3:o + 3 + n].reshape(F, K)
    assert np.array_equal(Qd, Q.reshape(F, K, 3), equal_nan=True) and np.array_equal(ed, err.reshape(F, K), equal_nan=True)
    assert np.array_equal(md, mask.reshape(F, K)) and np.array_equal(xd, nex.reshape(F, K))


def test_search_valve_boundary():
    """The work-list search does not enter a level with more subsets than the valve (2^26 in production: C(32, 11)
    and beyond).  With the valve lowered to C(8, 4) = 70 nothing changes; at 69 exactly the units that reach level 4
    are cut: they come back as not triangulated, are counted, and every other unit is untouched."""
    import __graft_entry__ as entry
    entry.build_hip()
    from pose2sim_amd import synth
    from pose2sim_amd.engine import Engine
    wl = synth.make_config(3000, 8, 26, 1, seed=99, p_outlier=0.30, p_lowlik=0.0, p_missing_cam=0.0)
    eng = Engine(0)
    try:
        eng.set_tuning(Engine.TUNE_TRI_PATH, Engine.TRI_PATH_WORKLIST)
        eng.set_calibration(wl['P'])
        prm = eng.tri_params(4.0, 0.3, 2)
        ref = eng.triangulate(wl['xyl'], prm)
        assert eng.tri_stats(reset=True)['capped_units'] == 0
        eng.set_tuning(Engine.TUNE_MAX_SUBSETS, 70)
        same = eng.triangulate(wl['xyl'], prm)
        assert eng.tri_stats(reset=True)['capped_units'] == 0
        for a, b in zip(ref, same):
            assert np.array_equal(a, b, equal_nan=True)
        eng.set_tuning(Engine.TUNE_MAX_SUBSETS, 69)
        cut = eng.triangulate(wl['xyl'], prm)
        capped = eng.tri_stats(reset=True)['capped_units']
        differ = ~(np.isclose(ref[0], cut[0], equal_nan=True, rtol=0, atol=0).all(axis=-1) & (ref[2] == cut[2]) & (ref[3] == cut[3]))
        assert capped > 0 and differ.sum() == capped
        assert np.isnan(cut[0][differ]).all() and np.isnan(cut[1][differ]).all()
        assert (cut[2][differ] == 3).all()                      # nb_cams_excluded of the last level that ran (no camera was out beforehand)
        assert (ref[2][differ] >= 4).all()                      # the full search went on to level 4 for exactly these units
    finally:
        eng.close()


@pytest.mark.parametrize('C,min_cams,lr_swap,undistort,f64', [
    (8, 2, False, False, False),
    (9, 2, True, False, False),
    (7, 2, True, True, False),
    (12, 3, False, False, True),
])
def test_deep_levels_spread_over_the_gpu(C, min_cams, lr_swap, undistort, f64):
    """Levels with more subsets than the deep threshold leave the search kernel's wave and are evaluated chunk by
    chunk by the whole GPU (p2s_tri_deep.hip: plan / eval / reduce rounds).  With the threshold lowered to 10 subsets
    every level from the second on takes that road (several rounds per unit, L/R swap candidates and undistortion
    included): every unit against the C oracle, and bit-identical to the all-in-one-wave search."""
    import __graft_entry__ as entry
    entry.build_hip()
    from oracle import tri_oracle
    from pose2sim_amd import skeletons, synth
    from pose2sim_amd.engine import Engine
    _, _, swap = skeletons.keypoints('HALPE_26')
    wl = synth.make_config(1500, C, 26, 1, seed=300 + C, undistort=undistort, lr_swap=lr_swap, swap_idx=swap,
                           p_outlier=0.12, p_lowlik=0.05, p_missing_cam=0.02)
    x64 = wl['xyl'].astype(np.float64) + (1e-9 if f64 else 0.0)
    x = x64 if f64 else wl['xyl']
    eng = Engine(0)
    try:
        eng.set_tuning(Engine.TUNE_TRI_PATH, Engine.TRI_PATH_WORKLIST)
        eng.set_calibration(wl['P'], wl['cams'] if undistort else None)
        prm = eng.tri_params(8.0, 0.3, min_cams, undistort, lr_swap)
        eng.set_tuning(Engine.TUNE_DEEP_MIN_SUBSETS, 0)
        ref = eng.triangulate(x, prm, swap if lr_swap else None)
        eng.set_tuning(Engine.TUNE_DEEP_MIN_SUBSETS, 10)
        eng.tri_stats(reset=True)
        got = eng.triangulate(x, prm, swap if lr_swap else None)
        for a, b in zip(ref, got):
            assert np.array_equal(a, b, equal_nan=True)
        threads = min(64, len(os.sched_getaffinity(0)))
        Qr, er, nr, mr = tri_oracle.triangulate_batch(x64, wl['P'], wl['cams'] if undistort else None, swap, 0.3, 8.0, min_cams,
                                                      lr_swap, undistort, threads=threads)
        _compare(got[0], got[1], got[2], got[3], Qr, er, nr, mr, f'deep C={C}')
        assert (nr.reshape(-1) - np.isnan(x64[..., 2]).sum(axis=2).reshape(-1) >= 2).any()     # some unit did go past level 1
    finally:
        eng.close()


@pytest.mark.parametrize('modes,seed', [(False, 11), (True, 12)])
def test_random_parameter_sets(modes, seed):
    """A short run of tests/sweeps/fuzz_params.py inside the suite: 60 random (cameras, keypoints, frames, min_cameras,
    thresholds, contamination, exact zero likelihoods) sets -- with undistortion, L/R swap, float64 observations and up to
    24 cameras switched on at random in the second run -- every unit against the C oracle on two kernel paths each.
    (The sweep itself: 1 400 sets, profiles/r03/fuzz_params_*.log.)"""
    import importlib.util
    import __graft_entry__ as entry
    entry.build_hip()
    spec = importlib.util.spec_from_file_location('fuzz_params', os.path.join(os.path.dirname(__file__), 'sweeps', 'fuzz_params.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, worst = mod.run(60, seed, modes, verbose=False)
    assert bad == 0 and worst <= TOL_Q, (bad, worst)
