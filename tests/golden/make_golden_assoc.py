"""Association goldens: the reference's compute_affinity / matchSVT / person_index_per_cam per frame."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))
from pose2sim_amd import synth  # noqa: E402


def make_scene(F, C, Pmax, Kj, seed, p_unseen=0.15, p_empty_cam=0.05, noise=2.0):
    """Per frame: every camera sees a random subset of the persons, in random order."""
    rng = np.random.default_rng(seed)
    cams = synth.make_cameras(C, seed=seed)
    Q3d = synth.make_points3d(F, Pmax, Kj, seed=seed)
    xyl = synth.make_observations(Q3d, cams, seed=seed, noise_px=noise, p_lowlik=0.05, p_outlier=0.02, p_missing_cam=0.0)
    frames = []
    for f in range(F):
        n_here = rng.integers(1, Pmax + 1)
        present = rng.permutation(Pmax)[:n_here]
        per_cam = []
        for c in range(C):
            if rng.random() < p_empty_cam:
                per_cam.append([])
                continue
            seen = [p for p in present if rng.random() > p_unseen]
            rng.shuffle(seen)
            people = []
            for p in seen:
                kp = xyl[f, p, c].astype(np.float64).copy()      # [Kj][3]
                if rng.random() < 0.1:                           # a few undetected joints
                    kp[rng.integers(0, Kj, 3)] = 0.0
                people.append(kp.ravel())
            per_cam.append(people)
        frames.append(per_cam)
    return cams, frames


def gen_association():
    common, tri, pa, sk = ref_shim.load()
    out = {}
    plan = [(40, 3, 2, 26, 61, 2), (60, 4, 3, 26, 62, 2), (80, 8, 4, 26, 63, 2), (40, 5, 4, 17, 64, 3), (12, 8, 6, 26, 65, 2)]
    for gi, (F, C, Pmax, Kj, seed, min_cams) in enumerate(plan):
        cams, frames = make_scene(F, C, Pmax, Kj, seed)
        cal = {'inv_K': cams['inv_K'], 'R_mat': cams['R_mat'], 'T': cams['T'], 'K': cams['K']}
        recon_thr, min_aff = 0.1, 0.2
        Nmax = max(sum(len(p) for p in fr) for fr in frames)
        n_persons = np.zeros((F, C), dtype=np.int32)
        rows = []
        aff_all = np.zeros((F, Nmax, Nmax))
        out_all = np.zeros((F, Nmax, Nmax))
        props_all = np.full((F, Nmax, C), np.nan)
        n_props = np.zeros(F, dtype=np.int32)
        for f, per_cam in enumerate(frames):
            n_persons[f] = [len(p) for p in per_cam]
            for people in per_cam:
                rows += [np.asarray(p) for p in people]
            cum = np.cumsum([0] + [len(p) for p in per_cam])
            N = cum[-1]
            with np.errstate(all='ignore'):
                aff = pa.compute_affinity([[list(p) for p in people] for people in per_cam], cal, cum,
                                          reconstruction_error_threshold=recon_thr)
                cc = pa.circular_constraint(cum)
                aff = aff * cc
                res = pa.matchSVT(aff, cum, cc, max_iter=20, w_rank=50, tol=1e-4, w_sparse=0.1)
                res[res < min_aff] = 0
                props = pa.person_index_per_cam(res, cum, min_cams)
            aff_all[f, :N, :N] = aff
            out_all[f, :N, :N] = res
            props = np.asarray(props, dtype=float).reshape(-1, C) if np.asarray(props).size else np.zeros((0, C))
            n_props[f] = props.shape[0]
            props_all[f, :props.shape[0]] = props
        g = dict(C=C, Kj=Kj, min_cams=min_cams, recon_thr=recon_thr, min_aff=min_aff, n_persons=n_persons,
                 kpts=np.array(rows).reshape(-1, Kj, 3), affinity=aff_all, result=out_all, proposals=props_all, n_props=n_props,
                 K=np.array(cams['K']), R=np.array(cams['R']), T=np.array(cams['T']), dist=np.array(cams['dist']),
                 S=np.array(cams['S']))
        for k, v in g.items():
            out[f'g{gi}_{k}'] = np.asarray(v)
        print(f'group {gi}: F={F} C={C} Nmax={Nmax} mean props={n_props.mean():.2f}', flush=True)
    out['n_groups'] = np.array(len(plan))
    np.savez_compressed(os.path.join(HERE, 'assoc_frames.npz'), **out)
    print('wrote assoc_frames.npz')


if __name__ == '__main__':
    gen_association()
