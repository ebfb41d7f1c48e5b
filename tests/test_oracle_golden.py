"""Pin the CPU oracle against fixtures recorded from the reference (tests/golden/make_golden.py)."""
import os

import numpy as np

from oracle import triangulation_ref as tr


def _groups(golden_dir):
    z = np.load(os.path.join(golden_dir, 'tri_units.npz'))
    n = int(z['n_groups'])
    for i in range(n):
        yield i, {k[len(f'g{i}_'):]: z[k] for k in z.files if k.startswith(f'g{i}_')}


def _cal(g):
    C = int(g['C'])
    return {'K': [g['K'][c] for c in range(C)], 'dist': [g['dist'][c] for c in range(C)],
            'R': [g['R'][c] for c in range(C)], 'T': [g['T'][c] for c in range(C)],
            'optim_K': [g['optim_K'][c] for c in range(C)]}


def test_unit_oracle_matches_reference(golden_dir):
    """triangulate_unit == triangulation_from_best_cameras on 2.4k recorded units:
    identical discrete outputs, Q within 1e-7 m (1e-4 mm), error within 1e-9 px."""
    total = 0
    for i, g in _groups(golden_dir):
        C = int(g['C'])
        P = [g['P'][c] for c in range(C)]
        cal = _cal(g)
        # C=16 levels are slow in pure Python: subsample the widest groups
        step = 1 if C <= 8 else 3
        for u in range(0, len(g['err']), step):
            Q, e, ne, ids = tr.triangulate_unit(g['coords'][u], g['coords_sw'][u], P, cal,
                                                float(g['thr']), int(g['min_cams']),
                                                bool(g['lr_swap']), bool(g['undistort']))
            assert ne == g['n_excl'][u], (i, u)
            assert tr.excluded_mask(ids) == int(g['mask'][u]), (i, u)
            assert np.isnan(e) == np.isnan(g['err'][u]), (i, u)
            if not np.isnan(e):
                assert abs(e - g['err'][u]) <= 1e-9 * max(1.0, abs(e)), (i, u)
                assert np.max(np.abs(Q - g['Q'][u])) <= 1e-7, (i, u, Q, g['Q'][u])
            else:
                assert np.isnan(Q).all()
            total += 1
    assert total > 1500


def test_c_oracle_matches_reference(golden_dir):
    """The compiled restatement (oracle/tri_oracle.c) on the same recorded units, fed through the
    batch entry point (raw float32 observations -> undistort -> mask -> search)."""
    from oracle import tri_oracle
    from pose2sim_amd import cvmath, skeletons
    _, _, swap = skeletons.keypoints('HALPE_26')
    worst = 0.0
    for i, g in _groups(golden_dir):
        C = int(g['C'])
        cams = _cal(g)
        cams['R_mat'] = [cvmath.rodrigues(r) for r in cams['R']]
        Q, e, ne, mask = tri_oracle.triangulate_batch(
            g['raw_xyl'], [g['P'][c] for c in range(C)], cams, swap, float(g['lik_thr']), float(g['thr']),
            int(g['min_cams']), bool(g['lr_swap']), bool(g['undistort']), threads=4)
        Q = Q.reshape(-1, 3); e = e.reshape(-1); ne = ne.reshape(-1); mask = mask.reshape(-1)
        assert np.array_equal(ne, g['n_excl']), i
        assert np.array_equal(mask, g['mask']), i
        assert np.array_equal(np.isnan(e), np.isnan(g['err'])), i
        ok = ~np.isnan(e)
        assert np.abs(e[ok] - g['err'][ok]).max() <= 1e-9 * max(1.0, np.abs(e[ok]).max()), i
        worst = max(worst, np.abs(Q[ok] - g['Q'][ok]).max())
    assert worst <= 1e-7, worst


def test_c_oracle_matches_numpy_oracle_on_fresh_data():
    from oracle import tri_oracle
    from pose2sim_amd import skeletons, synth
    _, _, swap = skeletons.keypoints('HALPE_26')
    for C, mc, sw, und in [(4, 2, False, False), (6, 3, True, True), (8, 2, True, False)]:
        wl = synth.make_config(12, C, 26, 1, seed=40 + C, undistort=und, lr_swap=sw, swap_idx=swap,
                               p_lowlik=0.1, p_outlier=0.08, p_missing_cam=0.03)
        a = tr.triangulate_batch(wl['xyl'], wl['P'], wl['cams'], swap, 0.3, 12.0, mc, sw, und)
        b = tri_oracle.triangulate_batch(wl['xyl'], wl['P'], wl['cams'], swap, 0.3, 12.0, mc, sw, und)
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
        assert np.array_equal(np.isnan(a[1]), np.isnan(b[1]))
        ok = ~np.isnan(a[1])
        assert np.abs(a[0][ok] - b[0][ok]).max() <= 1e-9


# ------------------------------------------------------------------------------------------------
def _assoc_groups(golden_dir):
    z = np.load(os.path.join(golden_dir, 'assoc_frames.npz'))
    for i in range(int(z['n_groups'])):
        yield i, {k[len(f'g{i}_'):]: z[k] for k in z.files if k.startswith(f'g{i}_')}


def assoc_frames_of(g):
    """Rebuild per-frame people lists and calibration from an association fixture group."""
    from pose2sim_amd import cvmath
    C = int(g['C'])
    cal = {'K': [g['K'][c] for c in range(C)], 'inv_K': [np.linalg.inv(g['K'][c]) for c in range(C)],
           'R_mat': [cvmath.rodrigues(g['R'][c]) for c in range(C)], 'T': [g['T'][c] for c in range(C)],
           'dist': [g['dist'][c] for c in range(C)], 'optim_K': [g['K'][c] for c in range(C)]}
    frames, row = [], 0
    for f in range(g['n_persons'].shape[0]):
        per_cam = []
        for c in range(C):
            n = int(g['n_persons'][f, c])
            per_cam.append([g['kpts'][row + i].ravel() for i in range(n)])
            row += n
        frames.append(per_cam)
    return cal, frames


def test_association_oracle_matches_reference(golden_dir):
    """affinity, matchSVT result and proposals of 232 recorded frames (C 3..8, up to 45 detections)."""
    from oracle import association_ref as ar
    n = 0
    for i, g in _assoc_groups(golden_dir):
        cal, frames = assoc_frames_of(g)
        C = int(g['C'])
        for f, per_cam in enumerate(frames):
            N = sum(len(p) for p in per_cam)
            aff, res, props = ar.associate_frame(per_cam, cal, float(g['recon_thr']), float(g['min_aff']), int(g['min_cams']))
            assert np.allclose(aff, g['affinity'][f, :N, :N], rtol=0, atol=1e-12), (i, f)
            assert np.allclose(res, g['result'][f, :N, :N], rtol=0, atol=1e-9), (i, f)
            k = int(g['n_props'][f])
            props = np.asarray(props, dtype=float).reshape(-1, C) if np.asarray(props).size else np.zeros((0, C))
            assert props.shape[0] == k, (i, f)
            assert np.array_equal(props, g['proposals'][f, :k], equal_nan=True), (i, f)
            n += 1
    assert n > 200


# ------------------------------------------------------------------------------------------------
def single_frames_of(z, name):
    """Per-frame people lists and calibration of a single-person association fixture."""
    from pose2sim_amd import cvmath
    C = z[f'{name}_n_persons'].shape[1]
    K = z[f'{name}_K']
    P = [np.hstack([K[c], np.zeros((3, 1))]) @ np.vstack([np.hstack([cvmath.rodrigues(z[f'{name}_R'][c]), z[f'{name}_T'][c].reshape(3, 1)]), [0, 0, 0, 1]])
         for c in range(C)]
    frames, row = [], 0
    for f in range(z[f'{name}_n_persons'].shape[0]):
        per_cam = []
        for c in range(C):
            n = int(z[f'{name}_n_persons'][f, c])
            per_cam.append([z[f'{name}_kpts'][row + i].ravel() for i in range(n)])
            row += n
        frames.append(per_cam)
    return P, frames


def test_single_person_oracle_matches_reference(golden_dir):
    from oracle import association_single_ref as sr
    z = np.load(os.path.join(golden_dir, 'e2e_single.npz'))
    n = 0
    for name in z['cases']:
        name = str(name)
        P, frames = single_frames_of(z, name)
        for f, per_cam in enumerate(frames):
            combos = sr.persons_combinations([len(p) for p in per_cam])
            e, cb, Q = sr.best_persons_and_cameras(per_cam, combos, P, 18, float(z[f'{name}_thr']), int(z[f'{name}_min_cams']), 0.3)
            assert np.array_equal(cb, z[f'{name}_best_comb'][f], equal_nan=True), (name, f, cb, z[f'{name}_best_comb'][f])
            assert abs(e - z[f'{name}_best_err'][f]) <= 1e-9 * max(1.0, abs(e)), (name, f)
            assert np.allclose(Q, z[f'{name}_best_Q'][f], rtol=0, atol=1e-9, equal_nan=True), (name, f)
            n += 1
    assert n == 50
