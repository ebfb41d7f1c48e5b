"""Generate the golden fixtures in this directory FROM THE REFERENCE (build container only).

    python -B tests/golden/make_golden.py [tri] [assoc] [host] [e2e]

Imports /root/reference/Pose2Sim through ref_shim.py (stand-ins for the packages this image lacks)
and records inputs -> outputs of the reference's own functions.  The fixtures (npz / json / trc)
are data; neither the reference nor this script's imports travel to the GPU box.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))
from pose2sim_amd import synth, skeletons  # noqa: E402


def _cfg(thr, min_cams, lr_swap, undistort):
    return {'triangulation': {'reproj_error_threshold_triangulation': thr,
                              'min_cameras_for_triangulation': min_cams,
                              'handle_LR_swap': lr_swap, 'undistort_points': undistort}}


def gen_triangulation_units():
    """Unit-level goldens of triangulation_from_best_cameras (triangulation.py:363-604)."""
    common, tri, pa, sk = ref_shim.load()
    import cv2   # the stand-in
    ids, names, swap_idx = skeletons.keypoints('HALPE_26')
    K = len(ids)
    groups = []
    gid = 0
    # (C, min_cams, lr_swap, undistort, lik_thr, thr, F, contamination)
    plan = [
        (2, 2, False, False, 0.3, 15.0, 6, dict(p_lowlik=0.10, p_outlier=0.05)),
        (3, 2, False, False, 0.3, 15.0, 6, dict(p_lowlik=0.10, p_outlier=0.08)),
        (3, 2, True, False, 0.3, 15.0, 6, dict(p_lowlik=0.10, p_outlier=0.08)),
        (4, 2, False, False, 0.3, 15.0, 8, dict(p_lowlik=0.10, p_outlier=0.08)),
        (4, 3, False, False, 0.3, 10.0, 6, dict(p_lowlik=0.15, p_outlier=0.10)),
        (4, 2, True, False, 0.3, 15.0, 8, dict(p_lowlik=0.10, p_outlier=0.08)),
        (4, 2, False, True, 0.3, 15.0, 6, dict(p_lowlik=0.10, p_outlier=0.08)),
        (4, 2, True, True, 0.3, 15.0, 6, dict(p_lowlik=0.10, p_outlier=0.08)),
        (5, 2, True, False, 0.3, 12.0, 6, dict(p_lowlik=0.10, p_outlier=0.10)),
        (6, 3, True, True, 0.3, 12.0, 5, dict(p_lowlik=0.10, p_outlier=0.10)),
        (8, 2, False, False, 0.3, 15.0, 8, dict(p_lowlik=0.05, p_outlier=0.03)),
        (8, 2, False, False, 0.3, 8.0, 5, dict(p_lowlik=0.15, p_outlier=0.12)),
        (8, 3, True, False, 0.3, 10.0, 4, dict(p_lowlik=0.10, p_outlier=0.10)),
        (8, 2, True, True, 0.3, 10.0, 4, dict(p_lowlik=0.10, p_outlier=0.10)),
        (8, 2, False, False, 0.0, 15.0, 3, dict(p_lowlik=0.10, p_outlier=0.05)),   # Q6: zero likelihoods
        (12, 4, False, False, 0.3, 10.0, 2, dict(p_lowlik=0.10, p_outlier=0.08)),
        (16, 3, False, False, 0.3, 12.0, 2, dict(p_lowlik=0.05, p_outlier=0.05)),
        (16, 12, True, True, 0.3, 8.0, 2, dict(p_lowlik=0.08, p_outlier=0.06)),
    ]
    for (C, min_cams, lr_swap, undistort, lik_thr, thr, F, cont) in plan:
        seed = 100 + gid
        wl = synth.make_config(F, C, K, 1, seed=seed, undistort=undistort, lr_swap=lr_swap,
                               swap_idx=swap_idx, p_missing_cam=0.03, **cont)
        xyl = wl['xyl'].astype(np.float64)
        cams = wl['cams']
        if lik_thr == 0.0:
            # exact zeros in the likelihood channel exercise the "== 0" rule (triangulation.py:436,446)
            rng = np.random.default_rng(seed)
            z = rng.random(xyl[..., 2].shape) < 0.08
            xyl[..., 2][z] = 0.0
        P = wl['P']
        cal = {'K': cams['K'], 'dist': cams['dist'], 'R': cams['R'], 'T': cams['T'],
               'optim_K': cams['optim_K']}
        cfg = _cfg(thr, min_cams, lr_swap, undistort)
        n = F * K
        coords = np.empty((n, 3, C))
        coords_sw = np.empty((n, 3, C))
        Qo = np.empty((n, 3)); eo = np.empty(n); no = np.empty(n, dtype=np.int32); mo = np.empty(n, dtype=np.uint32)
        u = 0
        for f in range(F):
            x = xyl[f, 0, :, :, 0].copy(); y = xyl[f, 0, :, :, 1].copy(); l = xyl[f, 0, :, :, 2].copy()
            if undistort:   # triangulation.py:808-813 through the cv2 stand-in
                pts = [np.array(tuple(zip(x[i], y[i]))).reshape(-1, 1, 2).astype('float32') for i in range(C)]
                und = [cv2.undistortPoints(pts[i], cal['K'][i], cal['dist'][i], None, cal['optim_K'][i]) for i in range(C)]
                x = np.array([[q[i][0][0] for i in range(len(q))] for q in und], dtype=np.float64)
                y = np.array([[q[i][0][1] for i in range(len(q))] for q in und], dtype=np.float64)
            with np.errstate(invalid='ignore'):   # :817-821
                x[l < lik_thr] = np.nan; y[l < lik_thr] = np.nan; l[l < lik_thr] = np.nan
            for k in range(K):
                ck = np.array((x[:, k], y[:, k], l[:, k]))
                cs = np.array((x[:, swap_idx[k]], y[:, swap_idx[k]], l[:, swap_idx[k]]))
                with np.errstate(all='ignore'):
                    Q, e, ne, idx = tri.triangulation_from_best_cameras(cfg, ck.copy(), cs.copy(), P, cal)
                coords[u] = ck; coords_sw[u] = cs
                Qo[u] = np.asarray(Q, dtype=np.float64); eo[u] = e; no[u] = ne
                m = 0
                for c in np.asarray(idx).ravel():
                    m |= 1 << int(c)
                mo[u] = m
                u += 1
        g = dict(C=C, min_cams=min_cams, lr_swap=lr_swap, undistort=undistort, lik_thr=lik_thr, thr=thr,
                 raw_xyl=xyl.astype(np.float32), coords=coords, coords_sw=coords_sw, P=np.array(P),
                 K=np.array(cams['K']), dist=np.array(cams['dist']), R=np.array(cams['R']),
                 T=np.array(cams['T']), optim_K=np.array(cams['optim_K']),
                 Q=Qo, err=eo, n_excl=no, mask=mo)
        groups.append(g)
        print(f'group {gid}: C={C} min={min_cams} swap={lr_swap} und={undistort} n={n} '
              f'ok={np.isfinite(eo).mean():.2f} mean_excl={no.mean():.2f}', flush=True)
        gid += 1
    flat = {}
    for i, g in enumerate(groups):
        for k, v in g.items():
            flat[f'g{i}_{k}'] = np.asarray(v)
    flat['n_groups'] = np.array(len(groups))
    np.savez_compressed(os.path.join(HERE, 'tri_units.npz'), **flat)
    print('wrote tri_units.npz', sum(len(g['err']) for g in groups), 'units')


if __name__ == '__main__':
    what = sys.argv[1:] or ['tri']
    if 'tri' in what:
        gen_triangulation_units()
    if 'assoc' in what:
        from make_golden_assoc import gen_association
        gen_association()
    if 'host' in what:
        from make_golden_host import gen_host
        gen_host()
    if 'e2e' in what:
        from make_golden_e2e import gen_e2e
        gen_e2e()
