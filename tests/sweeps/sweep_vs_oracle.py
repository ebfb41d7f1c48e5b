"""Exhaustive parity sweep: every unit of several multi-million-unit workloads against the C oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import tri_oracle
from pose2sim_amd import skeletons, synth
from pose2sim_amd.engine import Engine

_, _, swap26 = skeletons.keypoints('HALPE_26')
eng = Engine(0)
# P2S_SWEEP_PATH=worklist: the streaming + work-list search pair everywhere (default: the one-launch kernel where it applies)
if os.environ.get('P2S_SWEEP_PATH') == 'worklist':
    eng.set_tuning(Engine.TUNE_TRI_PATH, Engine.TRI_PATH_WORKLIST)
if os.environ.get('P2S_SWEEP_PATH') == 'onetile':      # round 2's one-launch kernel with one tile per wave everywhere
    eng.set_tuning(Engine.TUNE_TRI_PATH, Engine.TRI_PATH_ONE_TILE)
if os.environ.get('P2S_SWEEP_PATH') == 'twotiles':     # round 2's one-launch kernel, two tiles per wave
    eng.set_tuning(Engine.TUNE_TRI_PATH, Engine.TRI_PATH_TWO_TILES)
if os.environ.get('P2S_SWEEP_PATH') == 'noscreen':     # the pooled kernel with every camera subset sent to the fp64 evaluation
    eng.set_tuning(Engine.TUNE_SCREEN, 0)
threads = min(128, len(os.sched_getaffinity(0)))
cases = [
    dict(name='cfg2', F=100_000, C=8, K=26, min_cams=2, seed=2),
    dict(name='c8 min4 heavy', F=60_000, C=8, K=26, min_cams=4, seed=3, gen=dict(p_outlier=0.10, p_lowlik=0.10)),
    dict(name='c6 swap', F=80_000, C=6, K=26, min_cams=2, seed=4, lr_swap=True),
    dict(name='c5 undistort', F=80_000, C=5, K=26, min_cams=2, seed=5, undistort=True),
    dict(name='c3', F=150_000, C=3, K=26, min_cams=2, seed=6),
    dict(name='c2', F=150_000, C=2, K=26, min_cams=2, seed=7),
    dict(name='c16 k131 min3', F=12_000, C=16, K=131, min_cams=3, seed=8),
    dict(name='c7 undistort swap', F=40_000, C=7, K=26, min_cams=3, seed=9, undistort=True, lr_swap=True),
    dict(name='c8 f64 inputs', F=40_000, C=8, K=26, min_cams=2, seed=10, f64=True),
    dict(name='c32 undistort swap', F=400, C=32, K=26, min_cams=2, seed=11, undistort=True, lr_swap=True, gen=dict(p_outlier=0.01, p_lowlik=0.03)),
    dict(name='c32 min28', F=2000, C=32, K=26, min_cams=28, seed=12, gen=dict(p_outlier=0.03)),
    dict(name='c24 min20 swap', F=2000, C=24, K=26, min_cams=20, seed=13, lr_swap=True),
    dict(name='c20 min3', F=1500, C=20, K=26, min_cams=3, seed=14, gen=dict(p_outlier=0.01, p_lowlik=0.02)),
    dict(name='c17 f64 undistort', F=3000, C=17, K=26, min_cams=12, seed=15, undistort=True, f64=True),
    dict(name='c9 min2', F=30_000, C=9, K=26, min_cams=2, seed=16),
    dict(name='c4', F=100_000, C=4, K=26, min_cams=2, seed=17),
    dict(name='c6 heavy', F=60_000, C=6, K=26, min_cams=2, seed=18, gen=dict(p_outlier=0.12, p_lowlik=0.08)),
    dict(name='c8 third outliers', F=20_000, C=8, K=26, min_cams=2, seed=19, gen=dict(p_outlier=0.33)),
    dict(name='c12 min3', F=30_000, C=12, K=26, min_cams=3, seed=20, gen=dict(p_outlier=0.05)),
    dict(name='c16 heavy', F=6_000, C=16, K=26, min_cams=4, seed=21, gen=dict(p_outlier=0.08, p_lowlik=0.08)),
    dict(name='c8 min7', F=60_000, C=8, K=26, min_cams=7, seed=22, gen=dict(p_outlier=0.05)),
]
import sys as _s
sel = _s.argv[1:]
for cs in cases:
    if sel and not any(x in cs['name'] for x in sel):
        continue
    und, sw = cs.get('undistort', False), cs.get('lr_swap', False)
    K = cs['K']
    swap = swap26 if K == 26 else list(range(K))
    wl = synth.make_config(cs['F'], cs['C'], K, 1, seed=cs['seed'], undistort=und, lr_swap=sw, swap_idx=swap, **cs.get('gen', {}))
    xyl = wl['xyl'].astype(np.float64) + (1e-9 if cs.get('f64') else 0.0)
    eng.set_calibration(wl['P'], wl['cams'] if und else None)
    prm = eng.tri_params(15.0, 0.3, cs['min_cams'], und, sw)
    eng.tri_stats(reset=True)
    t0 = time.time()
    Q, err, nex, mask = eng.triangulate(xyl if cs.get('f64') else wl['xyl'], prm, swap if sw else None)
    t1 = time.time()
    st = eng.tri_stats(reset=True)
    Qr, er, nr, mr = tri_oracle.triangulate_batch(xyl, wl['P'], wl['cams'] if und else None, swap, 0.3, 15.0, cs['min_cams'], sw, und, threads=threads)
    t2 = time.time()
    Q = Q.reshape(-1, 3); Qr = Qr.reshape(-1, 3); err = err.reshape(-1).astype(np.float64); er = er.reshape(-1)
    nan_mis = int((np.isnan(err) != np.isnan(er)).sum())
    nex_mis = int((nex.reshape(-1).astype(np.int64) != nr.reshape(-1)).sum())
    mask_mis = int((mask.reshape(-1).astype(np.uint32) != mr.reshape(-1).astype(np.uint32)).sum())
    ok = ~np.isnan(er) & ~np.isnan(err)
    d = np.linalg.norm(Q[ok] - Qr[ok], axis=1)
    de = np.abs(err[ok] - er[ok]) / np.maximum(1.0, np.abs(er[ok]))
    print(f"{cs['name']:20s} units {len(er):8d} valid {ok.sum():8d} nan_mis {nan_mis} nex_mis {nex_mis} mask_mis {mask_mis} "
          f"max dQ {d.max():.2e} (>1e-7: {(d > 1e-7).sum()}) max dErr {de.max():.2e}  subsets fp64 {st['subsets_evaluated']} screened {st['screened_subsets']}  gpu {t1 - t0:.1f}s oracle {t2 - t1:.1f}s", flush=True)
    if nan_mis or nex_mis or mask_mis or (d > 1e-7).any():
        bad = np.flatnonzero((np.isnan(err) != np.isnan(er)) | (nex.reshape(-1).astype(np.int64) != nr.reshape(-1)) | (mask.reshape(-1).astype(np.uint32) != mr.reshape(-1).astype(np.uint32)))
        for i in bad[:5]:
            print('   unit', i, 'gpu', err[i], nex.reshape(-1)[i], bin(mask.reshape(-1)[i]), 'ref', er[i], nr.reshape(-1)[i], bin(mr.reshape(-1)[i]))
        idx = np.flatnonzero(ok)[np.argsort(-d)[:3]]
        for i in idx:
            print('   far unit', i, 'd', np.linalg.norm(Q[i] - Qr[i]), '|Q|', np.linalg.norm(Qr[i]), 'err', err[i], er[i], bin(mask.reshape(-1)[i]))
