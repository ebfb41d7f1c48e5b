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


# ---- single-person mode (personAssociation.py:67-257) ---------------------------------------------
class OracleSingleEngine:
    """Test double (tests only): Engine.associate_single through oracle/association_single_ref.py."""

    def set_calibration(self, P, cal=None):
        self.P = [np.asarray(p, dtype=np.float64) for p in P]

    def associate_single(self, n_persons, tracked, thr, lik_thr, min_cams):
        from oracle import association_single_ref as sr
        F, C = n_persons.shape
        comb = np.full((F, C), -1, dtype=np.int32)
        err = np.full(F, np.inf)
        Q = np.full((F, 3), np.nan)
        row = 0
        for f in range(F):
            per_cam = []
            for c in range(C):
                per_cam.append([list(tracked[row + i]) for i in range(n_persons[f, c])])
                row += n_persons[f, c]
            e, cb, q = sr.best_persons_and_cameras(per_cam, sr.persons_combinations(n_persons[f]), self.P, 0, thr,
                                                   min_cams, lik_thr)
            err[f], Q[f] = e, q
            comb[f] = np.where(np.isnan(cb), -1, cb).astype(np.int32)
        return comb, err, Q


def _run_single(golden_dir, tmp_path, monkeypatch, name):
    z = np.load(os.path.join(golden_dir, 'e2e_single.npz'))
    cams = ec.cams_from_arrays(z, prefix=name + '_')
    F, C = z[f'{name}_n_persons'].shape
    frames, row = [], 0
    for f in range(F):
        per_cam = []
        for c in range(C):
            n = int(z[f'{name}_n_persons'][f, c])
            per_cam.append([z[f'{name}_kpts'][row + i].ravel() for i in range(n)])
            row += n
        frames.append(per_cam)
    root = str(tmp_path / ('single_' + name))
    trial = ec.write_trial(root, 'trial_' + name, cams, frames, json_subdir='pose')
    cfg = ec.base_config(trial, False, min_cameras_for_triangulation=int(z[f'{name}_min_cams']))
    cfg['personAssociation']['single_person']['reproj_error_threshold_association'] = float(z[f'{name}_thr'])
    monkeypatch.chdir(root)
    pa.associate_all(cfg)
    got = {}
    d = os.path.join(trial, 'pose-associated')
    for cam in sorted(os.listdir(d)):
        for fn in sorted(os.listdir(os.path.join(d, cam))):
            got[f'{cam}/{fn}'] = open(os.path.join(d, cam, fn)).read()
    want = {str(n): str(t) for n, t in zip(z[f'{name}_names'], z[f'{name}_texts'])}
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k] == want[k], k


@pytest.mark.parametrize('name', ['s4', 's5'])
def test_single_person_json_matches_reference_with_oracle_backend(golden_dir, tmp_path, monkeypatch, name):
    monkeypatch.setattr(pa, '_make_engine', lambda: OracleSingleEngine())
    _run_single(golden_dir, tmp_path, monkeypatch, name)


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['s4', 's5'])
def test_single_person_json_matches_reference_on_gpu(golden_dir, tmp_path, monkeypatch, name):
    import __graft_entry__ as entry
    entry.build_hip()
    _run_single(golden_dir, tmp_path, monkeypatch, name)


def test_single_person_with_undistortion_is_refused(golden_dir, tmp_path, monkeypatch):
    """The reference's own code raises there (personAssociation.py:130); the message here says why."""
    z = np.load(os.path.join(golden_dir, 'e2e_assoc.npz'))
    cams = ec.cams_from_arrays(z)
    root = str(tmp_path / 'single')
    trial = ec.write_trial(root, 'trial_s', cams, [[[], [], [], []]], json_subdir='pose')
    monkeypatch.chdir(root)
    monkeypatch.setattr(pa, '_make_engine', lambda: OracleSingleEngine())
    with pytest.raises(NotImplementedError):
        pa.associate_all(ec.base_config(trial, False, undistort_points=True))
