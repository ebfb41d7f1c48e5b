"""Diagnostics: the units of the two-camera test whose |dQ| against the C oracle exceeds 1e-7 m, with their distance
from the origin, on both triangulation paths (is the excess a property of far, ill-posed points or of a kernel?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import tri_oracle
from pose2sim_amd import skeletons, synth
from pose2sim_amd.engine import Engine
_, _, swap = skeletons.keypoints('HALPE_26')
for C in (2, 3):
    wl = synth.make_config(60_000, C, 26, 1, seed=5 + C)
    Qr, er, nr, mr = tri_oracle.triangulate_batch(wl['xyl'].astype(np.float64), wl['P'], None, swap, 0.3, 15.0, 2, threads=32)
    for path in (0, 1):
        eng = Engine(0)
        eng.set_tuning(Engine.TUNE_TRI_PATH, path)
        eng.set_calibration(wl['P'])
        Q, err, nex, mask = eng.triangulate(wl['xyl'], eng.tri_params(15.0, 0.3, 2))
        ok = ~np.isnan(er.reshape(-1)) & ~np.isnan(err.reshape(-1))
        d = np.abs(Q.reshape(-1, 3) - Qr.reshape(-1, 3)).max(axis=1)
        d[~ok] = 0
        idx = np.argsort(-d)[:6]
        print(f'C={C} path={path} n>1e-7: {(d > 1e-7).sum()}  n>1e-8: {(d > 1e-8).sum()}')
        for i in idx:
            print(f'   unit {i} |dQ| {d[i]:.3e} |Q| {np.linalg.norm(Qr.reshape(-1, 3)[i]):.2f} m  err {er.reshape(-1)[i]:.3f} px')
        eng.close()
