"""End-to-end goldens: the reference's triangulate_all on small synthetic trials -> .trc text."""
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
from pose2sim_amd import skeletons, synth  # noqa: E402
import e2e_common as ec  # noqa: E402


def _run_reference(tri_mod, cfg, root):
    cwd = os.getcwd()
    os.chdir(root)
    try:
        with np.errstate(all='ignore'):
            tri_mod.triangulate_all(cfg)
    finally:
        os.chdir(cwd)
    out = {}
    d = os.path.join(cfg['project']['project_dir'], 'pose-3d')
    for name in sorted(os.listdir(d)):
        if name.endswith('.trc'):
            out[name] = open(os.path.join(d, name)).read()
    return out


def gen_e2e():
    common, tri, pa, sk = ref_shim.load()
    logging.getLogger().setLevel(logging.WARNING)
    ids, names, swap = skeletons.keypoints('HALPE_26')
    K = len(ids)
    cases = {}

    # ---- single person, 4 cameras, 80 frames, a dropout section (trim + interpolate + fill) ----
    wl = synth.make_config(80, 4, K, 1, seed=21, p_lowlik=0.06, p_outlier=0.05, p_missing_cam=0.02)
    xyl = wl['xyl'].copy()
    xyl[30:34, :, :, 5] = np.nan         # a short gap for keypoint 5 (interpolated)
    xyl[50:75, :, :, 9] = np.nan         # a long gap for keypoint 9 (filled)
    xyl[0:3] = np.nan                    # leading frames without any detection (trimmed)
    cases['single'] = dict(xyl=xyl, cams=wl['cams'], multi=False, tri={})
    # ---- same data, other post-processing options ------------------------------------------------
    cases['single_opts'] = dict(xyl=xyl, cams=wl['cams'], multi=False,
                                tri={'interpolation': 'cubic', 'fill_large_gaps_with': 'nan', 'sections_to_keep': 'largest',
                                     'interp_if_gap_smaller_than': 10, 'handle_LR_swap': True})
    # ---- undistort + swap, 5 cameras ------------------------------------------------------------
    wl2 = synth.make_config(40, 5, K, 1, seed=22, undistort=True, lr_swap=True, swap_idx=swap, p_lowlik=0.06, p_outlier=0.05)
    cases['undistort'] = dict(xyl=wl2['xyl'], cams=wl2['cams'], multi=False,
                              tri={'undistort_points': True, 'handle_LR_swap': True, 'fill_large_gaps_with': 'zeros'})
    # ---- two persons, 4 cameras, 50 frames, through pose-associated ------------------------------
    wl3 = synth.make_config(50, 4, K, 2, seed=23, p_lowlik=0.05, p_outlier=0.04, p_missing_cam=0.0)
    x3 = wl3['xyl'].copy()
    x3[20:26, 1] = np.nan                # person 1 leaves for a few frames
    cases['multi'] = dict(xyl=x3, cams=wl3['cams'], multi=True, tri={}, json_subdir='pose-associated')

    out = {}
    for name, cs in cases.items():
        root = tempfile.mkdtemp(prefix='p2s_e2e_')
        try:
            people = ec.people_from_xyl(cs['xyl'], ids, 26)
            trial = ec.write_trial(root, 'trial_' + name, cs['cams'], people, json_subdir=cs.get('json_subdir', 'pose'))
            cfg = ec.base_config(trial, cs['multi'], **cs['tri'])
            trcs = _run_reference(tri, cfg, root)
            print(name, '->', list(trcs), [len(v) for v in trcs.values()], flush=True)
            out[f'{name}_xyl'] = cs['xyl']
            for k in ('S', 'K', 'dist', 'R', 'T'):
                out[f'{name}_{k}'] = np.array(cs['cams'][k])
            out[f'{name}_multi'] = np.array(cs['multi'])
            out[f'{name}_json_subdir'] = np.array(cs.get('json_subdir', 'pose'))
            out[f'{name}_tri_keys'] = np.array(list(cs['tri'].keys()), dtype='U64')
            out[f'{name}_tri_vals'] = np.array([repr(v) for v in cs['tri'].values()], dtype='U64')
            out[f'{name}_trc_names'] = np.array(list(trcs.keys()), dtype='U128')
            out[f'{name}_trc_texts'] = np.array(list(trcs.values()), dtype=object).astype('U')
        finally:
            shutil.rmtree(root, ignore_errors=True)
    out['cases'] = np.array(list(cases.keys()), dtype='U32')
    np.savez_compressed(os.path.join(HERE, 'e2e_trc.npz'), **out)
    print('wrote e2e_trc.npz')


if __name__ == '__main__':
    gen_e2e()
