"""Details of the units of one case of tests/sweeps/fuzz_params.py (seed 1) that differ from the oracle."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import tri_oracle
from pose2sim_amd import synth
from pose2sim_amd.engine import Engine
rng = np.random.default_rng(1)
want = set(int(a) for a in sys.argv[1:])
eng = Engine(0)
for case in range(400):
    C = int(rng.integers(2, 17)); K = int(rng.choice([1, 3, 26, 33])); F = int(rng.choice([1, 2, 7, 64, 65, 200, 777]))
    min_cams = int(rng.integers(2, C + 1)); thr = float(rng.choice([1.0, 3.0, 15.0, 60.0])); lik = float(rng.choice([0.0, 0.1, 0.3, 0.9]))
    gen = dict(p_outlier=float(rng.choice([0.0, 0.03, 0.15, 0.4])), p_lowlik=float(rng.choice([0.0, 0.05, 0.3, 0.6])), p_missing_cam=float(rng.choice([0.0, 0.01, 0.2])))
    seed = int(rng.integers(1 << 30))
    zero = rng.random() < 0.3
    wl = synth.make_config(F, C, K, 1, seed=seed, **gen) if (zero or case in want) else None
    xyl = wl['xyl'] if wl else None
    if zero:
        xyl = xyl.copy(); z = rng.random(xyl.shape[:-1]) < 0.05; xyl[z] = 0.0
    if case not in want:
        continue
    if lik < 0.05:
        xyl = xyl.copy(); w = xyl[..., 2]; w[(w > 0) & (w < 0.05)] = 0.05      # as fuzz_params.py does
    Qr, er, nr, mr = tri_oracle.triangulate_batch(xyl.astype(np.float64), wl['P'], None, list(range(K)), lik, thr, min_cams, threads=8)
    eng.set_calibration(wl['P'])
    Q, err, nex, mask = eng.triangulate(xyl, eng.tri_params(thr, lik, min_cams))
    Q = Q.reshape(-1, 3); err = err.reshape(-1); nex = nex.reshape(-1); mask = mask.reshape(-1)
    Qo = np.asarray(Qr).reshape(-1, 3); eo = np.asarray(er).reshape(-1); no = np.asarray(nr).reshape(-1); mo = np.asarray(mr).reshape(-1)
    badu = np.flatnonzero((np.isnan(err) != np.isnan(eo)) | (nex != no) | (mask.astype(np.uint32) != mo.astype(np.uint32)) | (np.nan_to_num(np.abs(Q - Qo).max(axis=1)) > 1e-7))
    print('case', case, 'C', C, 'K', K, 'F', F, 'min_cams', min_cams, 'thr', thr, 'lik', lik, 'bad units', badu[:10])
    for u in badu[:4]:
        f, k = divmod(int(u), K)
        print('  unit', u, 'lik', np.round(xyl[f, 0, :, k, 2], 3))
        print('    got  err', err[u], 'nex', nex[u], 'mask', bin(int(mask[u])), 'Q', Q[u])
        print('    want err', eo[u], 'nex', no[u], 'mask', bin(int(mo[u])), 'Q', Qo[u])
