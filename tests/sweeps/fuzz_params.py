"""Randomised parameter sweep: many small workloads with random camera counts, keypoint counts, thresholds and
contamination, every unit of each against the C oracle, on the default path and on the work-list pair.
    python tests/sweeps/fuzz_params.py [n_cases] [seed] [plain|modes]
modes: undistortion, L/R swap and float64 observations switched on at random as well (then up to 24 cameras with
min_cameras close to the camera count, so that the search stays short), default path and round 2's one-launch kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import tri_oracle
from pose2sim_amd import synth
from pose2sim_amd.engine import Engine

from pose2sim_amd import skeletons
_, _, swap26 = skeletons.keypoints('HALPE_26')


def run(n_cases, seed, modes=False, verbose=True):
    """-> (cases with mismatches, worst |dQ| within 100 m).  Also called by tests/test_tri_gpu.py with a small n_cases."""
    rng = np.random.default_rng(seed)
    engines = {'auto': Engine(0), ('onetile' if modes else 'worklist'): Engine(0)}
    if modes:
        engines['onetile'].set_tuning(Engine.TUNE_TRI_PATH, Engine.TRI_PATH_ONE_TILE)
    else:
        engines['worklist'].set_tuning(Engine.TUNE_TRI_PATH, Engine.TRI_PATH_WORKLIST)
    threads = min(64, len(os.sched_getaffinity(0)))
    bad = 0
    worst = 0.0
    for case in range(n_cases):
        C = int(rng.integers(2, 17))
        K = int(rng.choice([1, 3, 26, 33]))
        F = int(rng.choice([1, 2, 7, 64, 65, 200, 777]))
        min_cams = int(rng.integers(2, C + 1))
        thr = float(rng.choice([1.0, 3.0, 15.0, 60.0]))
        lik = float(rng.choice([0.0, 0.1, 0.3, 0.9]))
        gen = dict(p_outlier=float(rng.choice([0.0, 0.03, 0.15, 0.4])), p_lowlik=float(rng.choice([0.0, 0.05, 0.3, 0.6])),
                   p_missing_cam=float(rng.choice([0.0, 0.01, 0.2])))
        und = sw = f64 = False
        swap = list(range(K))
        if modes:
            und, sw, f64 = bool(rng.random() < 0.4), bool(rng.random() < 0.4), bool(rng.random() < 0.3)
            if rng.random() < 0.3:
                C = int(rng.integers(17, 25)); min_cams = int(rng.integers(C - 3, C + 1)); F = min(F, 64)
            if sw:
                K = 26; swap = list(swap26)
        wl = synth.make_config(F, C, K, 1, seed=int(rng.integers(1 << 30)), undistort=und, lr_swap=sw, swap_idx=swap, **gen)
        xyl = wl['xyl']
        if rng.random() < 0.3:                                   # exact zero likelihoods (quirk Q6), as OpenPose writes them: (0, 0, 0)
            xyl = xyl.copy(); z = rng.random(xyl.shape[:-1]) < 0.05; xyl[z] = 0.0
        if lik < 0.05:
            # with a threshold of 0 a likelihood of 1e-3 passes: the camera then weighs 1e-6 of the others in the DLT and a
            # two-camera unit is a one-camera system to rounding (|dQ| of 1e-5 .. 1e-2 m between ANY two SVDs) -- not the subject
            xyl = xyl.copy(); w = xyl[..., 2]; w[(w > 0) & (w < 0.05)] = 0.05
        xin = xyl.astype(np.float64) + (1e-9 if f64 else 0.0)
        Qr, er, nr, mr = tri_oracle.triangulate_batch(xin, wl['P'], wl['cams'] if und else None, swap, lik, thr, min_cams, lr_swap=sw, undistort=und, threads=threads)
        for name, eng in engines.items():
            eng.set_calibration(wl['P'], wl['cams'] if und else None)
            Q, err, nex, mask = eng.triangulate(xin if f64 else xyl, eng.tri_params(thr, lik, min_cams, und, sw), swap if sw else None)
            Q = Q.reshape(-1, 3); err = err.reshape(-1); nex = nex.reshape(-1); mask = mask.reshape(-1)
            Qo = np.asarray(Qr).reshape(-1, 3); eo = np.asarray(er).reshape(-1)
            mis = int((np.isnan(err) != np.isnan(eo)).sum() + (nex.astype(np.int64) != np.asarray(nr).reshape(-1)).sum() +
                      (mask.astype(np.uint32) != np.asarray(mr).reshape(-1).astype(np.uint32)).sum())
            ok = ~np.isnan(eo) & ~np.isnan(err)
            dq = float(np.abs(Q[ok] - Qo[ok]).max()) if ok.any() else 0.0
            near = ok & (np.linalg.norm(Qo, axis=1) <= 100.0)
            dq_near = float(np.abs(Q[near] - Qo[near]).max()) if near.any() else 0.0
            worst = max(worst, dq_near)
            if mis or dq_near > 1e-7:
                bad += 1
                print(f'MISMATCH case {case} path {name}: C={C} K={K} F={F} min_cams={min_cams} thr={thr} lik={lik} und={und} swap={sw} f64={f64} {gen}: mismatches {mis} dQ(<=100 m) {dq_near:.2e} dQ {dq:.2e}')
    for eng in engines.values():
        eng.close()
    if verbose:
        print(f'{n_cases} cases x {len(engines)} paths: {bad} with mismatches, worst |dQ| within 100 m {worst:.2e} m')
    return bad, worst


if __name__ == '__main__':
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
        len(sys.argv) > 3 and sys.argv[3] == 'modes')
