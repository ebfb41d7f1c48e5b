"""BASELINE configs[0] / SURVEY 8c fixture 5: the Demo_SinglePerson cameras (Calib.qca.txt converted by the
reference's Utilities/calib_qca_to_toml.py), synthetic HALPE_26 JSON of 100 and 300 frames, through the
reference's triangulate_all -> .trc text.  Also the converter's TOML text for the three shipped .qca.txt
files.  The .qca.txt contents (calibration DATA shipped with the reference) are stored as inputs.
cv2.Rodrigues / lxml are stand-ins (ref_shim; xml.etree), so the rotation digits are those of
pose2sim_amd.cvmath.rodrigues_from_matrix: self-consistent, unpinned against OpenCV."""
import logging
import os
import shutil
import sys
import tempfile
import types
import xml.etree.ElementTree as ET

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))
from pose2sim_amd import calib as calib_mod, cvmath, skeletons, synth  # noqa: E402
import e2e_common as ec  # noqa: E402

DEMOS = ['Demo_SinglePerson', 'Demo_MultiPerson', 'Demo_Batch']


def demo_scene(calib_toml, F, K, seed):
    """One person moving in front of the demo cameras; the scene is scaled to the file's length unit
    (km, see calib_convert.py) so that the pixel geometry is that of a ~4 m capture volume."""
    P = calib_mod.computeP(calib_toml, undistort=False)
    cal = calib_mod.retrieve_calib_params(calib_toml)
    centers = np.array([-np.asarray(cal['R_mat'][c]).T @ np.asarray(cal['T'][c]) for c in range(len(P))])
    scale = np.linalg.norm(centers, axis=1).mean() / 4.0
    Q3d = synth.make_points3d(F, 1, K, seed=seed) * scale
    rng = np.random.default_rng(seed)
    C = len(P)
    xyl = np.full((F, 1, C, K, 3), np.nan, dtype=np.float32)
    Qh = np.concatenate([Q3d[:, 0], np.ones((F, K, 1))], axis=2)
    for c in range(C):
        uvw = Qh @ np.asarray(P[c]).T
        uv = uvw[..., :2] / uvw[..., 2:3] + rng.normal(0, 1.5, (F, K, 2))
        lik = rng.uniform(0.3, 1.0, (F, K))
        lik[rng.random((F, K)) < 0.05] = rng.uniform(0.0, 0.3)
        uv[rng.random((F, K)) < 0.03] += rng.normal(0, 60, 2)
        xyl[:, 0, c, :, :2] = uv.astype(np.float32)
        xyl[:, 0, c, :, 2] = lik.astype(np.float32)
    return xyl


def gen():
    common, tri, pa, sk = ref_shim.load()
    lx = types.ModuleType('lxml'); et = types.ModuleType('lxml.etree'); et.parse = ET.parse; lx.etree = et
    sys.modules['lxml'] = lx; sys.modules['lxml.etree'] = et
    import importlib
    conv = importlib.import_module('Pose2Sim.Utilities.calib_qca_to_toml')
    logging.getLogger().setLevel(logging.WARNING)
    out = {}
    tmp = tempfile.mkdtemp(prefix='p2s_cfg1_')
    try:
        for d in DEMOS:
            src = os.path.join(ref_shim.REFERENCE_ROOT, 'Pose2Sim', d, 'calibration', 'Calib.qca.txt')
            dst = os.path.join(tmp, d + '.qca.txt')
            shutil.copy(src, dst)
            conv.calib_qca_to_toml_func(dst)
            out[f'{d}_qca'] = np.array(open(dst).read())
            out[f'{d}_toml'] = np.array(open(dst.replace('.qca.txt', '.toml')).read())
        ids, names, swap = skeletons.keypoints('HALPE_26')
        K = len(ids)
        toml_path = os.path.join(tmp, 'Demo_SinglePerson.toml')
        for F in (100, 300):
            xyl = demo_scene(toml_path, F, K, seed=100 + F)
            root = os.path.join(tmp, f'session_{F}')
            frames = ec.people_from_xyl(xyl, ids, 26)
            trial = ec.write_trial(root, f'trial_{F}', None, frames, json_subdir='pose', calib_text=str(out['Demo_SinglePerson_toml']))
            cfg = ec.base_config(trial, False)
            cwd = os.getcwd()
            os.chdir(root)
            try:
                with np.errstate(all='ignore'):
                    tri.triangulate_all(cfg)
            finally:
                os.chdir(cwd)
            d3 = os.path.join(trial, 'pose-3d')
            trcs = sorted(f for f in os.listdir(d3) if f.endswith('.trc'))
            assert len(trcs) == 1, trcs
            out[f'F{F}_xyl'] = xyl
            out[f'F{F}_trc_name'] = np.array(trcs[0])
            out[f'F{F}_trc'] = np.array(open(os.path.join(d3, trcs[0])).read())
            print(F, trcs[0], len(str(out[f'F{F}_trc'])), 'chars')
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    np.savez_compressed(os.path.join(HERE, 'cfg1_demo.npz'), **out)
    print('wrote cfg1_demo.npz')


if __name__ == '__main__':
    gen()
