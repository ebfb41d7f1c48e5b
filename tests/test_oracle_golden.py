"""Pin the CPU oracle against fixtures recorded from the reference (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

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
