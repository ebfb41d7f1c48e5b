"""Randomised association sweep: random camera / person counts, noise, missing detections and association parameters,
60 frames each, the kernel's thresholded affinity matrix and the proposals against the NumPy oracle.
    python tests/sweeps/fuzz_assoc.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from oracle import association_ref as ar
from pose2sim_amd import personAssociation as pa
from pose2sim_amd.engine import Engine
from multiprocessing import Pool


def ref_frame(args):
    per_cam, cal, thr, min_aff, min_cams = args
    with np.errstate(all='ignore'):
        _, res, props = ar.associate_frame(per_cam, cal, thr, min_aff, min_cams)
    return res, np.asarray(props, dtype=float)


if __name__ == '__main__':
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    eng = Engine(0)
    bad_cases, worst = 0, 0.0
    with Pool(min(32, os.cpu_count())) as pool:
        for case in range(n_cases):
            C = int(rng.integers(2, 13)); Pn = int(rng.integers(1, 7)); F = 60
            if C * Pn > 48:
                Pn = 48 // C
            noise = dict(noise_px=float(rng.choice([0.5, 2.0, 6.0])), p_missing_cam=float(rng.choice([0.0, 0.1, 0.3])))
            # a person a camera does not see is left out of its list; kept as an all-zero detection (affinity 1 with
            # everybody) the exact ties are broken by 1e-13 rounding noise, in the reference as well
            drop = True
            thr, min_aff, min_cams = float(rng.choice([0.05, 0.1, 0.3])), float(rng.choice([0.1, 0.2, 0.5])), int(rng.choice([2, 3]))
            cfg = dict(bench.CONFIGS['cfg3']); cfg.update(F=F, C=C, Pn=Pn, gen=noise, seed=int(rng.integers(1 << 20)))
            xyl, cams, P, swap, K = bench.make_workload(cfg, 0)
            n_persons, kpts = bench.make_association_inputs(xyl, 11, drop_missing=drop)
            eng.set_calibration(P, cams)
            aff = eng.associate(n_persons, kpts, Engine.assoc_params(thr, min_aff, min_cams))
            cal = {'inv_K': cams['inv_K'], 'R_mat': cams['R_mat'], 'T': cams['T']}
            jobs, row = [], 0
            for f in range(F):
                per_cam = []
                for c in range(C):
                    per_cam.append([kpts[row + i].astype(np.float64).ravel() for i in range(n_persons[f, c])])
                    row += n_persons[f, c]
                jobs.append((per_cam, cal, thr, min_aff, min_cams))
            refs = pool.map(ref_frame, jobs, chunksize=4)
            w, mis = 0.0, 0
            for f, (res, props) in enumerate(refs):
                N = int(n_persons[f].sum())
                if N == 0:
                    continue
                w = max(w, float(np.abs(aff[f, :N, :N] - res).max()))
                cum = np.cumsum([0] + list(n_persons[f]))
                got = np.asarray(pa.person_index_per_cam(aff[f, :N, :N].copy(), cum, min_cams), dtype=float)
                got = got.reshape(-1, C) if got.size else np.zeros((0, C))
                props = props.reshape(-1, C) if props.size else np.zeros((0, C))
                if got.shape != props.shape or not np.array_equal(got, props, equal_nan=True):
                    mis += 1
            worst = max(worst, w)
            if mis or w > 1e-7:
                bad_cases += 1
                print(f'MISMATCH case {case}: C={C} persons={Pn} {noise} drop={drop} thr={thr} min_aff={min_aff} min_cams={min_cams}: max |d affinity| {w:.2e}, frames with different proposals {mis}', flush=True)
    print(f'{n_cases} cases x 60 frames: {bad_cases} with mismatches, worst |d affinity| {worst:.2e}')
