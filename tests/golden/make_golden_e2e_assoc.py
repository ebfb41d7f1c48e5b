"""End-to-end association golden: the reference's associate_all on a small multi-person trial."""
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
import e2e_common as ec  # noqa: E402
from make_golden_assoc import make_scene  # noqa: E402


def gen():
    common, tri, pa, sk = ref_shim.load()
    logging.getLogger().setLevel(logging.WARNING)
    F, C, Pmax, Kj = 30, 4, 3, 26
    cams, frames = make_scene(F, C, Pmax, Kj, seed=71)
    cams['names'] = [f'cam_{c + 1:02d}' for c in range(C)]
    frames[7][2] = None            # a missing file
    root = tempfile.mkdtemp(prefix='p2s_e2e_assoc_')
    out = {}
    try:
        trial = ec.write_trial(root, 'trial_assoc', cams, frames, json_subdir='pose')
        cfg = ec.base_config(trial, True)
        cwd = os.getcwd()
        os.chdir(root)
        try:
            with np.errstate(all='ignore'):
                pa.associate_all(cfg)
        finally:
            os.chdir(cwd)
        names, texts = [], []
        d = os.path.join(trial, 'pose-associated')
        for cam in sorted(os.listdir(d)):
            for fn in sorted(os.listdir(os.path.join(d, cam))):
                names.append(f'{cam}/{fn}')
                texts.append(open(os.path.join(d, cam, fn)).read())
        print('associated files:', len(names))
        n_persons = np.array([[0 if p is None else len(p) for p in per_cam] for per_cam in frames], dtype=np.int32)
        missing = np.array([[p is None for p in per_cam] for per_cam in frames])
        rows = [np.asarray(p) for per_cam in frames for people in per_cam if people is not None for p in people]
        out = dict(n_persons=n_persons, missing=missing, kpts=np.array(rows).reshape(-1, Kj, 3),
                   S=np.array(cams['S']), K=np.array(cams['K']), dist=np.array(cams['dist']), R=np.array(cams['R']),
                   T=np.array(cams['T']), names=np.array(names, dtype='U64'), texts=np.array(texts, dtype='U'))
    finally:
        shutil.rmtree(root, ignore_errors=True)
    np.savez_compressed(os.path.join(HERE, 'e2e_assoc.npz'), **out)
    print('wrote e2e_assoc.npz')


if __name__ == '__main__':
    gen()
