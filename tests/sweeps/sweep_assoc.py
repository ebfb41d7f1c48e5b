"""Association parity sweep: GPU kernel against the NumPy oracle on many random frames (cfg3-like and harder)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from oracle import association_ref as ar
from pose2sim_amd import personAssociation as pa
from pose2sim_amd.engine import Engine
from multiprocessing import Pool

PARTIAL = 4      # ADMM passes of the second comparison: the iterate is still continuous there (the converged one is binary)


def ref_frame(args):
    per_cam, cal, thr, min_aff, min_cams = args
    with np.errstate(all='ignore'):
        _, res, props = ar.associate_frame(per_cam, cal, thr, min_aff, min_cams)
        cum = np.cumsum([0] + [len(p) for p in per_cam])
        part = ar.match_svt(ar.affinity_matrix(per_cam, cal, cum, thr), cum, max_iter=PARTIAL) if cum[-1] else None
    return res, np.asarray(props, dtype=float), part

if __name__ == '__main__':
    eng = Engine(0)
    if os.environ.get('P2S_SWEEP_FORM') == 'general':          # the kernel that assumes no symmetry, at every size
        eng.set_tuning(Engine.TUNE_ASSOC_FORM, Engine.ASSOC_FORM_GENERAL)
    for name, C, Pn, F, noise in (('cfg3', 8, 4, 1500, dict(p_missing_cam=0.0)), ('c4 p6 noisy', 4, 6, 800, dict(noise_px=6.0, p_missing_cam=0.0)), ('c12 p3', 12, 3, 600, dict(p_missing_cam=0.0)), ('cfg3 with all-zero duplicates', 8, 4, 600, {}),
                                  ('c4 p3 (<= 16)', 4, 3, 600, dict(p_missing_cam=0.0)), ('c5 p3 ragged', 5, 3, 600, dict(p_missing_cam=0.15, drop_missing=True))):
        noise = dict(noise)
        drop = noise.pop('drop_missing', False)      # ragged camera lists instead of all-zero detections
        cfg = dict(bench.CONFIGS['cfg3']); cfg.update(F=F, C=C, Pn=Pn, gen=noise)
        xyl, cams, P, swap, K = bench.make_workload(cfg, 0)
        n_persons, kpts = bench.make_association_inputs(xyl, 11, drop_missing=drop)
        eng.set_calibration(P, cams)
        prm = Engine.assoc_params(0.1, 0.2, 2)
        aff = eng.associate(n_persons, kpts, prm)
        aff_part = eng.associate(n_persons, kpts, Engine.assoc_params(0.1, -1.0, 2, max_iter=PARTIAL))
        cal = {'inv_K': cams['inv_K'], 'R_mat': cams['R_mat'], 'T': cams['T']}
        jobs, row = [], 0
        for f in range(F):
            per_cam = []
            for c in range(C):
                per_cam.append([kpts[row + i].astype(np.float64).ravel() for i in range(n_persons[f, c])])
                row += n_persons[f, c]
            jobs.append((per_cam, cal, 0.1, 0.2, 2))
        t0 = time.time()
        with Pool(min(32, os.cpu_count())) as pool:
            refs = pool.map(ref_frame, jobs, chunksize=8)
        worst, worst_part, prop_mis, n_props = 0.0, 0.0, 0, 0
        for f, (res, props, part) in enumerate(refs):
            N = int(n_persons[f].sum())
            if N == 0:
                continue
            worst = max(worst, float(np.abs(aff[f, :N, :N] - res).max()))
            worst_part = max(worst_part, float(np.abs(aff_part[f, :N, :N] - part).max()))
            cum = np.cumsum([0] + list(n_persons[f]))
            got = np.asarray(pa.person_index_per_cam(aff[f, :N, :N].copy(), cum, 2), dtype=float)
            got = got.reshape(-1, C) if got.size else np.zeros((0, C))
            props = props.reshape(-1, C) if props.size else np.zeros((0, C))
            n_props += len(props)
            if got.shape != props.shape or not np.array_equal(got, props, equal_nan=True):
                prop_mis += 1
                if prop_mis <= 3:
                    print('  frame', f, 'proposals differ:\n', got, '\n', props, '\n  max |d aff|', np.abs(aff[f, :N, :N] - res).max())
        print(f'{name:14s} frames {F} max |d affinity| {worst:.2e} (after {PARTIAL} passes {worst_part:.2e}) frames with different proposals {prop_mis} (of {n_props} proposals)  oracle {time.time() - t0:.0f}s', flush=True)
