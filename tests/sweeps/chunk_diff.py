import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import tri_oracle
from pose2sim_amd import skeletons, synth
from pose2sim_amd.engine import Engine
_, _, swap = skeletons.keypoints('HALPE_26')
F, C, K = int(sys.argv[1]), int(sys.argv[2]), 26
seed = int(sys.argv[3])
wl = synth.make_config(F, C, K, 1, seed=seed, **({'p_outlier': 0.04} if len(sys.argv) > 4 else {}))
eng = Engine(0); eng.set_calibration(wl['P'])
Q, err, nex, mask = eng.triangulate(wl['xyl'], eng.tri_params(15.0, 0.3, 2))
Qr, er, nr, mr = tri_oracle.triangulate_batch(wl['xyl'].astype(np.float64), wl['P'], None, swap, 0.3, 15.0, 2, threads=64)
Q = Q.reshape(-1, 3); Qr = Qr.reshape(-1, 3); err = err.reshape(-1); er = er.reshape(-1)
ok = ~np.isnan(er)
d = np.linalg.norm(Q - Qr, axis=1); d[~ok] = 0
idx = np.argsort(-d)[:12]
for i in idx:
    print(i, 'd=%.3e' % d[i], '|Q|=%.3e' % np.linalg.norm(Qr[i]), 'err', err[i], er[i], 'nex', nex.reshape(-1)[i], bin(mask.reshape(-1)[i]))
print('count d>1e-7:', (d > 1e-7).sum(), 'of', ok.sum(), ' count with |Q|<20 and d>1e-7:', ((d > 1e-7) & (np.linalg.norm(Qr, axis=1) < 20)).sum())
rel = d / np.maximum(1.0, np.linalg.norm(Qr, axis=1)) ** 2
print('max d/|Q|^2', rel.max())
