"""End-to-end golden of the reference's SINGLE-person association (personAssociation.py:67-257,
769-780): associate_all on small trials with distractor persons -> the rewritten JSON files."""
import logging
import os
import shutil
import sys
import tempfile
import io
import contextlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))
import e2e_common as ec  # noqa: E402
from e2e_common import make_single_scene  # noqa: E402


def gen():
    common, tri, pa, sk = ref_shim.load()
    logging.getLogger().setLevel(logging.WARNING)
    out = {}
    cases = {'s4': (30, 4, 26, 81, 2, 20.0), 's5': (20, 5, 26, 82, 3, 12.0)}
    for name, (F, C, Kj, seed, min_cams, thr) in cases.items():
        cams, frames = make_single_scene(F, C, Kj, seed)
        cams['names'] = [f'cam_{c + 1:02d}' for c in range(C)]
        root = tempfile.mkdtemp(prefix='p2s_e2e_single_')
        try:
            trial = ec.write_trial(root, 'trial_' + name, cams, frames, json_subdir='pose')
            cfg = ec.base_config(trial, False, min_cameras_for_triangulation=min_cams)
            cfg['personAssociation']['single_person']['reproj_error_threshold_association'] = thr
            cwd = os.getcwd()
            os.chdir(root)
            try:
                with np.errstate(all='ignore'), contextlib.redirect_stdout(io.StringIO()):
                    pa.associate_all(cfg)
            finally:
                os.chdir(cwd)
            # per-frame internals straight from the reference functions (:67-99, :154-257)
            P_all = common.computeP(os.path.join(root, 'calibration', 'Calib.toml'), undistort=False)
            calib_params = common.retrieve_calib_params(os.path.join(root, 'calibration', 'Calib.toml'))
            kid = 18   # HALPE_26 'Neck'
            b_err, b_comb, b_Q = [], [], []
            for f in range(F):
                files = [os.path.join(trial, 'pose', f'cam_{c + 1:02d}_json', f'cam_{c + 1:02d}_{f:06d}.json') for c in range(C)]
                with np.errstate(all='ignore'), contextlib.redirect_stdout(io.StringIO()):
                    combs = pa.persons_combinations(files)
                    e, cb, q = pa.best_persons_and_cameras_combination(cfg, files, combs, P_all, kid, calib_params)
                b_err.append(float(e)); b_comb.append(np.asarray(cb[0], dtype=float)); b_Q.append(np.asarray(q[0], dtype=float)[:3])
            out[f'{name}_best_err'] = np.array(b_err); out[f'{name}_best_comb'] = np.array(b_comb); out[f'{name}_best_Q'] = np.array(b_Q)
            names, texts = [], []
            d = os.path.join(trial, 'pose-associated')
            for cam in sorted(os.listdir(d)):
                for fn in sorted(os.listdir(os.path.join(d, cam))):
                    names.append(f'{cam}/{fn}')
                    texts.append(open(os.path.join(d, cam, fn)).read())
            print(name, 'associated files:', len(names))
            n_persons = np.array([[len(p) for p in per_cam] for per_cam in frames], dtype=np.int32)
            rows = [np.asarray(p) for per_cam in frames for people in per_cam for p in people]
            for k, v in dict(n_persons=n_persons, kpts=np.array(rows).reshape(-1, Kj, 3), min_cams=min_cams, thr=thr,
                             S=np.array(cams['S']), K=np.array(cams['K']), dist=np.array(cams['dist']),
                             R=np.array(cams['R']), T=np.array(cams['T']), names=np.array(names, dtype='U64'),
                             texts=np.array(texts, dtype='U')).items():
                out[f'{name}_{k}'] = np.asarray(v)
        finally:
            shutil.rmtree(root, ignore_errors=True)
    out['cases'] = np.array(list(cases), dtype='U8')
    np.savez_compressed(os.path.join(HERE, 'e2e_single.npz'), **out)
    print('wrote e2e_single.npz')


if __name__ == '__main__':
    gen()
