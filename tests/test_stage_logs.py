"""The log lines of triangulate_all / associate_all (recap of reprojection errors, excluded cameras, interpolated
frames, thresholds, output paths) against the lines the reference logged on the same trials
(tests/golden/stage_logs.json <- make_golden_logs.py).  CPU: oracle-backed test doubles stand in for the HIP engine."""
import ast
import json
import logging
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import e2e_common as ec  # noqa: E402

from pose2sim_amd import personAssociation, skeletons, triangulation  # noqa: E402
from test_e2e_trc import OracleEngine  # noqa: E402


class Capture(logging.Handler):
    def __init__(self):
        super().__init__(level=logging.INFO)
        self.lines = []

    def emit(self, record):
        self.lines.append(f'{record.levelname}|{record.getMessage()}')


def _captured(fn, cfg, root, monkeypatch):
    cap = Capture()
    log = logging.getLogger()
    old_level, old_handlers = log.level, log.handlers[:]
    log.handlers = [cap]
    log.setLevel(logging.INFO)
    monkeypatch.chdir(root)
    try:
        with np.errstate(all='ignore'):
            fn(cfg)
    finally:
        log.handlers = old_handlers
        log.setLevel(old_level)
    real = os.path.realpath(root)
    return [ln.replace(real, '<ROOT>').replace(root, '<ROOT>') for ln in cap.lines]


@pytest.fixture(scope='module')
def gold(golden_dir):
    return json.load(open(os.path.join(golden_dir, 'stage_logs.json')))


def test_triangulation_log_lines(golden_dir, gold, tmp_path, monkeypatch):
    monkeypatch.setattr(triangulation, '_make_engine', lambda: OracleEngine())
    ids, names, swap = skeletons.keypoints('HALPE_26')
    z = np.load(os.path.join(golden_dir, 'e2e_trc.npz'), allow_pickle=False)
    for name in [str(n) for n in z['cases']]:
        tri = {str(k): ast.literal_eval(str(v)) for k, v in zip(z[f'{name}_tri_keys'], z[f'{name}_tri_vals'])}
        root = str(tmp_path / name)
        trial = ec.write_trial(root, 'trial_' + name, ec.cams_from_arrays(z, f'{name}_'), ec.people_from_xyl(z[f'{name}_xyl'], ids, 26),
                               json_subdir=str(z[f'{name}_json_subdir']))
        got = _captured(triangulation.triangulate_all, ec.base_config(trial, bool(z[f'{name}_multi']), **tri), root, monkeypatch)
        assert got == gold['tri_' + name], name


def _frames(z, pre, multi):
    n_persons = z[pre + 'n_persons']
    F, C = n_persons.shape
    frames, row = [], 0
    for f in range(F):
        per_cam = []
        for c in range(C):
            if multi and z['missing'][f, c]:
                per_cam.append(None)
                continue
            n = int(n_persons[f, c])
            per_cam.append([z[pre + 'kpts'][row + i].ravel() for i in range(n)])
            row += n
        frames.append(per_cam)
    return frames


def test_association_log_lines(golden_dir, gold, tmp_path, monkeypatch):
    from test_e2e_assoc import OracleAssocEngine, OracleSingleEngine
    za = np.load(os.path.join(golden_dir, 'e2e_assoc.npz'), allow_pickle=False)
    root = str(tmp_path / 'multi')
    trial = ec.write_trial(root, 'trial_assoc', ec.cams_from_arrays(za), _frames(za, '', True), json_subdir='pose')
    monkeypatch.setattr(personAssociation, '_make_engine', lambda: OracleAssocEngine())
    assert _captured(personAssociation.associate_all, ec.base_config(trial, True), root, monkeypatch) == gold['assoc_multi']
    zs = np.load(os.path.join(golden_dir, 'e2e_single.npz'), allow_pickle=False)
    root = str(tmp_path / 'single')
    trial = ec.write_trial(root, 'trial_s4', ec.cams_from_arrays(zs, prefix='s4_'), _frames(zs, 's4_', False), json_subdir='pose')
    cfg = ec.base_config(trial, False, min_cameras_for_triangulation=int(zs['s4_min_cams']))
    cfg['personAssociation']['single_person']['reproj_error_threshold_association'] = float(zs['s4_thr'])
    monkeypatch.setattr(personAssociation, '_make_engine', lambda: OracleSingleEngine())
    assert _captured(personAssociation.associate_all, cfg, root, monkeypatch) == gold['assoc_single']
