"""End-to-end: pose JSON folders + calibration TOML -> .trc, against .trc files written by the
reference's own triangulate_all on the same trials (tests/golden/make_golden_e2e.py).

CPU run: the host pipeline (IO, tracking, interpolation, trimming, .trc writer) with the kernel
work delegated to the oracle through a TEST DOUBLE of the engine -> byte-identical .trc.
GPU run (-m gpu): the real HIP engine -> identical header / frame range, coordinates within 1e-7 m.
"""
import ast
import io
import os
import sys

import numpy as np
import pandas as pd
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import e2e_common as ec  # noqa: E402

from pose2sim_amd import skeletons, triangulation  # noqa: E402


class OracleEngine:
    """Test double with the Engine interface, backed by the C oracle (tests only)."""

    def __init__(self):
        self.P = None
        self.cal = None

    def set_calibration(self, P, cal=None):
        self.P, self.cal = P, cal

    @staticmethod
    def tri_params(thr, lik_thr, min_cams, undistort=False, lr_swap=False):
        return dict(thr=thr, lik=lik_thr, min_cams=min_cams, undistort=bool(undistort), lr_swap=bool(lr_swap))

    def triangulate(self, xyl, prm, swap_idx=None):
        from oracle import triangulation_ref as tr
        K = xyl.shape[-2]
        sw = list(swap_idx) if swap_idx is not None else list(range(K))
        lead = xyl.shape[:-3]
        x = np.asarray(xyl, dtype=np.float64).reshape((-1, 1) + xyl.shape[-3:])
        Q, e, n, m = tr.triangulate_batch(x, self.P, self.cal, sw, prm['lik'], prm['thr'], prm['min_cams'],
                                          prm['lr_swap'], prm['undistort'])
        return (Q.reshape(lead + (K, 3)), e.reshape(lead + (K,)).astype(np.float64), n.reshape(lead + (K,)),
                m.reshape(lead + (K,)))


def _cases(golden_dir):
    z = np.load(os.path.join(golden_dir, 'e2e_trc.npz'), allow_pickle=False)
    for name in z['cases']:
        name = str(name)
        tri = {str(k): ast.literal_eval(str(v)) for k, v in zip(z[f'{name}_tri_keys'], z[f'{name}_tri_vals'])}
        yield name, z, tri


def _run(name, z, tri, tmp_path, monkeypatch):
    ids, names, swap = skeletons.keypoints('HALPE_26')
    cams = ec.cams_from_arrays(z, f'{name}_')
    people = ec.people_from_xyl(z[f'{name}_xyl'], ids, 26)
    root = str(tmp_path / name)
    trial = ec.write_trial(root, 'trial_' + name, cams, people, json_subdir=str(z[f'{name}_json_subdir']))
    cfg = ec.base_config(trial, bool(z[f'{name}_multi']), **tri)
    monkeypatch.chdir(root)
    paths = triangulation.triangulate_all(cfg)
    got = {os.path.basename(p): open(p).read() for p in paths if p}
    want = {str(n): str(t) for n, t in zip(z[f'{name}_trc_names'], z[f'{name}_trc_texts'])}
    return got, want


def test_trc_bytes_match_reference_with_oracle_backend(golden_dir, tmp_path, monkeypatch):
    monkeypatch.setattr(triangulation, '_make_engine', lambda: OracleEngine())
    for name, z, tri in _cases(golden_dir):
        got, want = _run(name, z, tri, tmp_path, monkeypatch)
        assert sorted(got) == sorted(want), (name, sorted(got), sorted(want))
        for fn in want:
            assert got[fn] == want[fn], f'{name}/{fn} differs'


def _parse(text):
    lines = text.split('\n')
    return lines[:5], pd.read_csv(io.StringIO('\n'.join(lines[5:])), sep='\t', header=None).to_numpy(dtype=np.float64)


@pytest.mark.gpu
def test_trc_matches_reference_on_gpu(golden_dir, tmp_path, monkeypatch):
    import __graft_entry__ as entry
    entry.build_hip()
    for name, z, tri in _cases(golden_dir):
        got, want = _run(name, z, tri, tmp_path, monkeypatch)
        assert sorted(got) == sorted(want), (name, sorted(got), sorted(want))
        for fn in want:
            hg, dg = _parse(got[fn])
            hw, dw = _parse(want[fn])
            assert hg == hw, f'{name}/{fn}: header differs'
            assert dg.shape == dw.shape
            assert np.array_equal(np.isnan(dg), np.isnan(dw)), f'{name}/{fn}: NaN pattern'
            d = np.nanmax(np.abs(dg - dw))
            assert d <= 1e-7, f'{name}/{fn}: max |d| = {d:.3e} m'
