"""Batch proposal extraction (native argmax rows, csrc/p2s_proposals.cpp, + the reference's NumPy calls) against
person_index_per_cam, the restatement of personAssociation.py:512-549 that the goldens pin: identical rows in
identical order on random thresholded matrices full of ties, empty cameras and shared detections.
Host-only code: runs without a GPU."""
import numpy as np
import pytest

import __graft_entry__ as entry
from pose2sim_amd import personAssociation as pa


@pytest.fixture(scope='module', autouse=True)
def built():
    entry.build_hip()


def _random_frames(rng, F, C, pmax, n_max):
    n_persons = rng.integers(0, pmax + 1, (F, C)).astype(np.int32)
    n_persons[rng.random(F) < 0.05] = 0
    aff = np.zeros((F, n_max, n_max))
    for f in range(F):
        N = int(n_persons[f].sum())
        if N == 0:
            continue
        m = rng.random((N, N))
        m = (m + m.T) / 2
        m[rng.random((N, N)) < 0.55] = 0.0                    # thresholded entries
        m = np.round(m, rng.choice([1, 2, 6]))                # many exact ties
        m = np.minimum(m, m.T)
        np.fill_diagonal(m, 1.0)
        aff[f, :N, :N] = m
    return n_persons, aff


@pytest.mark.parametrize('C,pmax,min_cams,seed', [(4, 3, 2, 1), (8, 4, 2, 2), (3, 6, 2, 3), (6, 2, 3, 4), (12, 3, 4, 5), (5, 8, 2, 6)])
def test_native_proposals_equal_numpy_form(C, pmax, min_cams, seed):
    rng = np.random.default_rng(seed)
    F = 400
    n_max = max(2, (C * pmax + 1) & ~1)
    n_persons, aff = _random_frames(rng, F, C, pmax, n_max)
    got = pa.proposals_batch(aff, n_persons, min_cams)
    handed_back = 0
    for f in range(F):
        cum = np.cumsum([0] + list(n_persons[f]))
        N = int(cum[-1])
        want = np.asarray(pa.person_index_per_cam(aff[f, :N, :N].copy(), cum, min_cams), dtype=float)
        g = np.asarray(got[f], dtype=float)
        assert g.size == want.size, (f, g, want)
        if want.size:
            assert np.array_equal(g.reshape(-1, C), want.reshape(-1, C), equal_nan=True), (f, g, want)
