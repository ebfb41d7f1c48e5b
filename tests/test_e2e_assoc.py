"""End-to-end association: pose/<cam>_json -> pose-associated/<cam>_json, against the files the
reference's associate_all wrote for the same trial (tests/golden/make_golden_e2e_assoc.py).
CPU: host logic with an oracle-backed TEST DOUBLE of the engine; GPU (-m gpu): the HIP engine.
Both must reproduce every output file byte for byte (same people, same order, same {} gaps)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import e2e_common as ec  # noqa: E402

from pose2sim_amd import personAssociation as pa  # noqa: E402


class OracleAssocEngine:
    """Test double (tests only): Engine.associate through oracle/association_ref.py."""

    def set_calibration(self, P, cal=None):
        self.cal = cal

    @staticmethod
    def assoc_params(recon_thr, min_affinity, min_cams, **kw):
        return dict(recon_thr=recon_thr, min_aff=min_affinity, min_cams=min_cams)

    def associate(self, n_persons, kpts, prm):
        from oracle import association_ref as ar
        F, C = n_persons.shape
        n_max = max(2, (int(n_persons.sum(axis=1).max()) + 1) & ~1)
        out = np.zeros((F, n_max, n_max))
        row = 0
        for f in range(F):
            per_cam = []
            for c in range(C):
                per_cam.append([kpts[row + i].ravel() for i in range(n_persons[f, c])])
                row += n_persons[f, c]
            N = int(n_persons[f].sum())
            if N:
                _, res, _ = ar.associate_frame(per_cam, self.cal, prm['recon_thr'], prm['min_aff'], prm['min_cams'])
                out[f, :N, :N] = res
        return out


def _run(golden_dir, tmp_path, monkeypatch):
    z = np.load(os.path.join(golden_dir, 'e2e_assoc.npz'))
    cams = ec.cams_from_arrays(z)
    F, C = z['n_persons'].shape
    frames, row = [], 0
    for f in range(F):
        per_cam = []
        for c in range(C):
            if z['missing'][f, c]:
                per_cam.append(None)
                continue
            n = int(z['n_persons'][f, c])
            per_cam.append([z['kpts'][row + i].ravel() for i in range(n)])
            row += n
        frames.append(per_cam)
    root = str(tmp_path / 'assoc')
    trial = ec.write_trial(root, 'trial_assoc', cams, frames, json_subdir='pose')
    cfg = ec.base_config(trial, True)
    monkeypatch.chdir(root)
    pa.associate_all(cfg)
    got = {}
    d = os.path.join(trial, 'pose-associated')
    for cam in sorted(os.listdir(d)):
        for fn in sorted(os.listdir(os.path.join(d, cam))):
            got[f'{cam}/{fn}'] = open(os.path.join(d, cam, fn)).read()
    want = {str(n): str(t) for n, t in zip(z['names'], z['texts'])}
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k] == want[k], k


def test_associated_json_matches_reference_with_oracle_backend(golden_dir, tmp_path, monkeypatch):
    monkeypatch.setattr(pa, '_make_engine', lambda: OracleAssocEngine())
    _run(golden_dir, tmp_path, monkeypatch)


@pytest.mark.gpu
def test_associated_json_matches_reference_on_gpu(golden_dir, tmp_path, monkeypatch):
    import __graft_entry__ as entry
    entry.build_hip()
    _run(golden_dir, tmp_path, monkeypatch)


def test_single_person_mode_is_refused(golden_dir, tmp_path, monkeypatch):
    z = np.load(os.path.join(golden_dir, 'e2e_assoc.npz'))
    cams = ec.cams_from_arrays(z)
    root = str(tmp_path / 'single')
    trial = ec.write_trial(root, 'trial_s', cams, [[[], [], [], []]], json_subdir='pose')
    monkeypatch.chdir(root)
    with pytest.raises(NotImplementedError):
        pa.associate_all(ec.base_config(trial, False))
