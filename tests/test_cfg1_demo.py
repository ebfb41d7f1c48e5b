"""BASELINE configs[0]: the Demo_SinglePerson cameras (Calib.qca.txt -> TOML) + synthetic HALPE_26 JSON,
100 and 300 frames -> .trc, against the reference's own converter and triangulate_all
(tests/golden/make_golden_cfg1.py).  CPU: host pipeline with the oracle-backed test double -> identical
bytes; GPU (-m gpu): HIP engine -> identical header, coordinates within 1e-7 of the file's unit RELATIVE to
the scene (the converter's translations are in km, so the scene is ~4e-3 units wide: 1e-10 absolute)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import e2e_common as ec  # noqa: E402
from test_e2e_trc import OracleEngine, _parse  # noqa: E402

from pose2sim_amd import calib_convert, skeletons, triangulation  # noqa: E402


@pytest.fixture(scope='module')
def z(golden_dir):
    return np.load(os.path.join(golden_dir, 'cfg1_demo.npz'))


@pytest.mark.parametrize('demo', ['Demo_SinglePerson', 'Demo_MultiPerson', 'Demo_Batch'])
def test_qca_converter_writes_the_reference_text(z, tmp_path, demo):
    qca = tmp_path / 'Calib.qca.txt'
    qca.write_text(str(z[f'{demo}_qca']))
    out = calib_convert.calib_qca_to_toml(str(qca))
    assert out == str(tmp_path / 'Calib.toml')
    assert open(out).read() == str(z[f'{demo}_toml'])


def _run(z, F, tmp_path, monkeypatch):
    ids, names, swap = skeletons.keypoints('HALPE_26')
    # calibration through this repo's converter, from the shipped .qca.txt data
    root = str(tmp_path / f'session_{F}')
    os.makedirs(os.path.join(root, 'calibration'))
    qca = os.path.join(root, 'calibration', 'Calib.qca.txt')
    with open(qca, 'w') as fh:
        fh.write(str(z['Demo_SinglePerson_qca']))
    toml_text = open(calib_convert.calib_qca_to_toml(qca)).read()
    people = ec.people_from_xyl(z[f'F{F}_xyl'], ids, 26)
    trial = ec.write_trial(root, f'trial_{F}', None, people, json_subdir='pose', calib_text=toml_text)
    monkeypatch.chdir(root)
    paths = triangulation.triangulate_all(ec.base_config(trial, False))
    assert [os.path.basename(p) for p in paths] == [str(z[f'F{F}_trc_name'])]
    return open(paths[0]).read(), str(z[f'F{F}_trc'])


@pytest.mark.parametrize('F', [100, 300])
def test_demo_trc_bytes_match_reference_with_oracle_backend(z, tmp_path, monkeypatch, F):
    monkeypatch.setattr(triangulation, '_make_engine', lambda: OracleEngine())
    got, want = _run(z, F, tmp_path, monkeypatch)
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize('F', [100, 300])
def test_demo_trc_matches_reference_on_gpu(z, tmp_path, monkeypatch, F):
    import __graft_entry__ as entry
    entry.build_hip()
    got, want = _run(z, F, tmp_path, monkeypatch)
    hg, dg = _parse(got)
    hw, dw = _parse(want)
    assert hg == hw
    assert dg.shape == dw.shape and np.array_equal(np.isnan(dg), np.isnan(dw))
    assert np.nanmax(np.abs(dg[:, 2:] - dw[:, 2:])) <= 1e-10
