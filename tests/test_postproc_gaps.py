"""Gap interpolation and gap filling of the trial table (triangulation.py:889-894, 922-926) on arrays: the in-place,
gaps-only forms of pose2sim_amd/postproc.py against the table-wide pandas / scipy formulation of the same contract."""
import numpy as np
import pandas as pd
import pytest
from scipy import interpolate

from pose2sim_amd import postproc


def _table(F=400, cols=12, seed=3):
    rng = np.random.default_rng(seed)
    t = np.cumsum(rng.normal(size=(F, cols)), axis=0)
    t[rng.random((F, cols)) < 0.06] = np.nan
    t[rng.random((F, cols)) < 0.01] = 0.0                      # a 0 is a gap too (common.py:692)
    t[40:75, 2] = np.nan                                        # longer than max_gap
    t[:9, 3] = np.nan                                           # leading gap
    t[-6:, 4] = np.nan                                          # trailing gap
    t[:, 5] = np.nan                                            # never seen
    t[:, 6] = np.nan
    t[[3, 50, 120, 300], 6] = [1.0, 2.0, 3.0, 4.0]              # 4 good samples: left alone
    t[10, 7] = np.inf
    return t


def _whole_column(vals, labels, max_gap, kind):
    good = ~(np.isnan(vals) | (vals == 0))
    if good.sum() <= 4:
        return vals.copy()
    f = interpolate.interp1d(labels[good], vals[good], kind=kind, fill_value='extrapolate', bounds_error=False)
    out = np.where(good, vals, f(labels))
    bad = np.flatnonzero(~good)
    for run in np.split(bad, np.flatnonzero(np.diff(bad) > 1) + 1):
        if len(run) > max_gap:
            out[run] = np.nan
    return out


@pytest.mark.parametrize('kind', ['linear', 'slinear', 'quadratic', 'cubic'])
def test_interpolate_gaps_in_place_equals_the_whole_table_form(kind):
    t = _table()
    labels = np.arange(17, 17 + len(t))                         # frame numbers, not positions
    want = np.stack([_whole_column(t[:, c], labels, 10, kind) for c in range(t.shape[1])], axis=1)
    got = postproc.interpolate_gaps(t, labels, 10, kind)
    assert got is t                                             # in place
    assert np.array_equal(got, want, equal_nan=True)


def test_interpolate_zeros_nans_keeps_the_pandas_entry():
    t = _table()
    col = pd.Series(t[:, 0].copy(), index=np.arange(5, 5 + len(t)), name='x')
    out = postproc.interpolate_zeros_nans(col, 10, 'cubic')
    assert np.array_equal(out.to_numpy(), _whole_column(t[:, 0], np.asarray(col.index), 10, 'cubic'), equal_nan=True)
    assert out.name == 'x' and out.index.equals(col.index)
    assert np.array_equal(col.to_numpy(), t[:, 0], equal_nan=True)          # the caller's column is not written to


def test_a_failing_column_leaves_the_table_as_it_was(monkeypatch):
    t = _table()
    before = t.copy()
    calls = {'n': 0}
    real = postproc._interpolate_column

    def failing(vals, labels, max_gap, kind):
        calls['n'] += 1
        if calls['n'] == 5:
            raise ValueError('no interpolant')
        return real(vals, labels, max_gap, kind)
    monkeypatch.setattr(postproc, '_interpolate_column', failing)
    with pytest.raises(ValueError):
        postproc.interpolate_gaps(t, np.arange(len(t)), 10, 'linear')
    assert np.array_equal(t, before, equal_nan=True)            # triangulation.py:891-894: all columns or none


@pytest.mark.parametrize('how', ['last_value', 'zeros', 'nan'])
def test_fill_gaps_equals_the_dataframe_form(how):
    t = _table()
    df = pd.DataFrame(t.copy())
    if how == 'last_value':
        df = df.ffill(axis=0).bfill(axis=0)
    if how in ('last_value', 'zeros'):
        df = df.replace([np.nan, np.inf], 0)
    got = postproc.fill_gaps(t, how)
    assert got is t
    assert np.array_equal(got, df.to_numpy(), equal_nan=True)
