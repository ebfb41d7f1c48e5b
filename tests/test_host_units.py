"""Sequential host steps against cases recorded from the reference (tests/golden/make_golden_host.py):
person tracking (common.py:1037-1136), gap interpolation (common.py:669-712), valid-section selection
(triangulation.py:93-148) and the skeleton order / swap map of every built-in model (:734-749)."""
import os
import warnings

import numpy as np
import pandas as pd
import pytest

from pose2sim_amd import postproc, skeletons


@pytest.fixture(scope='module')
def z(golden_dir):
    return np.load(os.path.join(golden_dir, 'host_units.npz'))


def test_person_tracking_matches_reference(z):
    for i in range(int(z['n_sort'])):
        max_dist = None if np.isnan(z[f'sort{i}_max']) else float(z[f'sort{i}_max'])
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            res = postproc.sort_people_sports2d(z[f'sort{i}_prev'].copy(), z[f'sort{i}_curr'].copy(), max_dist=max_dist)
        assert len(res) == int(z[f'sort{i}_n_out']), i
        for k, r in enumerate(res):
            want = z[f'sort{i}_out{k}']
            assert np.asarray(r).shape == want.shape, (i, k)
            assert np.array_equal(np.asarray(r, dtype=float), want, equal_nan=True), (i, k)


def test_interpolation_matches_reference(z):
    for i in range(int(z['n_interp'])):
        col = z[f'interp{i}_col']
        start = int(z[f'interp{i}_start'])
        s = pd.Series(col.copy(), index=range(start, start + len(col)))
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            got = postproc.interpolate_zeros_nans(s, int(z[f'interp{i}_N']), str(z[f'interp{i}_kind']))
        assert np.array_equal(np.asarray(got, dtype=float), z[f'interp{i}_out'], equal_nan=True), i


def test_valid_sections_match_reference(z):
    for i in range(int(z['n_chunk'])):
        mcs = int(z[f'chunk{i}_mcs'])
        got = postproc.indices_of_first_last_non_nan_chunks(pd.Series(z[f'chunk{i}_v']), min_chunk_size=None if mcs < 0 else mcs,
                                                            chunk_choice_method=str(z[f'chunk{i}_method']))
        assert tuple(int(v) for v in got) == tuple(int(v) for v in z[f'chunk{i}_out']), i


def test_skeleton_tables_match_reference(z):
    assert len(z['skel_models']) >= 7
    for m in [str(x) for x in z['skel_models']]:
        ids, names, swap = skeletons.keypoints(m)
        assert list(ids) == list(z[f'skel_{m}_ids']), m
        assert list(names) == [str(n) for n in z[f'skel_{m}_names']], m
        assert list(swap) == list(z[f'skel_{m}_swap']), m
