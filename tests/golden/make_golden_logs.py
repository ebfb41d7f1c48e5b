"""Goldens of the stages' log lines (recap_triangulate triangulation.py:255-360, recap_tracking
personAssociation.py:583-639 and everything else the stage functions log), recorded from the reference on the trials
of e2e_trc.npz / e2e_assoc.npz / e2e_single.npz.  The scratch directory of a run appears as <ROOT> -> stage_logs.json"""
import ast
import json
import logging
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))
from pose2sim_amd import skeletons  # noqa: E402
import e2e_common as ec  # noqa: E402


class Capture(logging.Handler):
    def __init__(self):
        super().__init__(level=logging.INFO)
        self.lines = []

    def emit(self, record):
        self.lines.append(f'{record.levelname}|{record.getMessage()}')


def run_captured(fn, cfg, root):
    cap = Capture()
    log = logging.getLogger()
    old_level, old_handlers = log.level, log.handlers[:]
    log.handlers = [cap]
    log.setLevel(logging.INFO)
    cwd = os.getcwd()
    os.chdir(root)
    try:
        with np.errstate(all='ignore'):
            fn(cfg)
    finally:
        os.chdir(cwd)
        log.handlers = old_handlers
        log.setLevel(old_level)
    real = os.path.realpath(root)
    return [ln.replace(real, '<ROOT>').replace(root, '<ROOT>') for ln in cap.lines]


def frames_from_arrays(z, pre, multi):
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


def gen():
    common, tri, pa, sk = ref_shim.load()
    ids, names, swap = skeletons.keypoints('HALPE_26')
    out = {}
    z = np.load(os.path.join(HERE, 'e2e_trc.npz'), allow_pickle=False)
    for name in [str(n) for n in z['cases']]:
        tri_over = {str(k): ast.literal_eval(str(v)) for k, v in zip(z[f'{name}_tri_keys'], z[f'{name}_tri_vals'])}
        root = tempfile.mkdtemp(prefix='p2s_logs_')
        try:
            trial = ec.write_trial(root, 'trial_' + name, ec.cams_from_arrays(z, f'{name}_'), ec.people_from_xyl(z[f'{name}_xyl'], ids, 26),
                                   json_subdir=str(z[f'{name}_json_subdir']))
            out['tri_' + name] = run_captured(tri.triangulate_all, ec.base_config(trial, bool(z[f'{name}_multi']), **tri_over), root)
        finally:
            shutil.rmtree(root, ignore_errors=True)
    za = np.load(os.path.join(HERE, 'e2e_assoc.npz'), allow_pickle=False)
    root = tempfile.mkdtemp(prefix='p2s_logs_')
    try:
        trial = ec.write_trial(root, 'trial_assoc', ec.cams_from_arrays(za), frames_from_arrays(za, '', True), json_subdir='pose')
        out['assoc_multi'] = run_captured(pa.associate_all, ec.base_config(trial, True), root)
    finally:
        shutil.rmtree(root, ignore_errors=True)
    zs = np.load(os.path.join(HERE, 'e2e_single.npz'), allow_pickle=False)
    root = tempfile.mkdtemp(prefix='p2s_logs_')
    try:
        trial = ec.write_trial(root, 'trial_s4', ec.cams_from_arrays(zs, prefix='s4_'), frames_from_arrays(zs, 's4_', False), json_subdir='pose')
        cfg = ec.base_config(trial, False, min_cameras_for_triangulation=int(zs['s4_min_cams']))
        cfg['personAssociation']['single_person']['reproj_error_threshold_association'] = float(zs['s4_thr'])
        out['assoc_single'] = run_captured(pa.associate_all, cfg, root)
    finally:
        shutil.rmtree(root, ignore_errors=True)
    with open(os.path.join(HERE, 'stage_logs.json'), 'w') as fh:
        json.dump(out, fh, indent=1)
    for k, v in out.items():
        print(k, len(v), 'lines')


if __name__ == '__main__':
    gen()
