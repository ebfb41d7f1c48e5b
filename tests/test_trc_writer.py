"""Native .trc row writer (csrc/p2s_trc.cpp) against what the reference writes the rows with:
DataFrame.to_csv(sep='\\t', index=True, header=None, lineterminator='\\n') (triangulation.py:214), i.e. Python's
repr() of every float and an empty field for NaN.  Host-only code: runs without a GPU."""
import ctypes as C
import io
import random
import struct

import numpy as np
import pandas as pd
import pytest

import __graft_entry__ as entry
from pose2sim_amd import _lib, trc


@pytest.fixture(scope='module', autouse=True)
def built():
    entry.build_hip()


def test_float_text_is_python_repr():
    lib = _lib.load()
    buf = C.create_string_buffer(64)
    rng = random.Random(2)
    vals = [0.0, -0.0, 1.0, -1.5, 1e16, 1e15, 9999999999999998.0, 1e-4, 1e-5, 0.00012345, 123456789012345680.0, 5e-324,
            1.7976931348623157e308, float('inf'), float('-inf'), 2.0, 100.0, 1e22, 1.2345678901234567e-7, 0.1 + 0.2, 1 / 3]
    for _ in range(100000):
        k = rng.random()
        if k < 0.4:
            vals.append(rng.uniform(-5, 5))
        elif k < 0.6:
            vals.append(struct.unpack('d', struct.pack('Q', rng.getrandbits(64)))[0])
        elif k < 0.8:
            vals.append(rng.uniform(-1, 1) * 10 ** rng.randint(-30, 30))
        else:
            vals.append(round(rng.uniform(-100, 100), rng.randint(0, 6)))
    for v in vals:
        n = lib.p2s_format_float_repr(v, buf, 64)
        got = buf.value.decode()
        assert n == len(got)
        assert got == ('' if v != v else repr(v)), (repr(v), got)


@pytest.mark.parametrize('n_rows,n_cols', [(0, 6), (1, 3), (37, 78), (9000, 12)])
def test_rows_equal_pandas_to_csv(tmp_path, n_rows, n_cols):
    rng = np.random.default_rng(n_rows + n_cols)
    data = rng.normal(0, 2, (n_rows, n_cols))
    data[rng.random((n_rows, n_cols)) < 0.05] = np.nan
    data[rng.random((n_rows, n_cols)) < 0.02] = 0.0
    data[rng.random((n_rows, n_cols)) < 0.01] *= 1e-7
    if n_rows > 2:
        data[1, 0] = np.inf
        data[2, 1] = 3.0
    start = 17
    frames = np.arange(start, start + n_rows)
    df = pd.DataFrame(data, index=frames)
    df.insert(0, 't', df.index / 60)
    want = io.StringIO()
    df.to_csv(want, sep='\t', index=True, header=None, lineterminator='\n')
    path = tmp_path / 'rows.trc'
    path.write_text('header\n')
    trc.write_rows(str(path), np.asarray(df.index), df.iloc[:, 0].to_numpy(), df.iloc[:, 1:].to_numpy())
    assert path.read_text() == 'header\n' + want.getvalue()
