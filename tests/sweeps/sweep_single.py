"""Single-person association parity sweep: GPU kernel against the NumPy oracle on many random frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'golden'))
import numpy as np
from multiprocessing import Pool
from e2e_common import make_single_scene
from oracle import association_single_ref as sr
from pose2sim_amd import synth
from pose2sim_amd.engine import Engine

def ref_frame(args):
    per_cam, P, thr, min_cams = args
    with np.errstate(all='ignore'):
        e, cb, q = sr.best_persons_and_cameras(per_cam, sr.persons_combinations([len(p) for p in per_cam]), P, 18, thr, min_cams, 0.3)
    return e, np.where(np.isnan(cb), -1, cb).astype(np.int32), q

if __name__ == '__main__':
    eng = Engine(0)
    for C, nd, F, thr, min_cams, seed in ((4, 2, 600, 20.0, 2, 1), (5, 2, 500, 6.0, 3, 2), (6, 1, 500, 3.0, 2, 3), (3, 3, 600, 10.0, 2, 4),
                                          (7, 1, 300, 1.0, 2, 5), (4, 3, 400, 0.2, 2, 6)):
        cams, frames = make_single_scene(F, C, 26, 1000 + seed, n_distract=nd)
        P = [np.asarray(p) for p in synth.projection_matrices(cams)]
        eng.set_calibration(np.array(P))
        n_persons = np.array([[len(p) for p in per_cam] for per_cam in frames], dtype=np.int32)
        tracked = np.array([np.asarray(p)[54:57] for per_cam in frames for people in per_cam for p in people], dtype=np.float64).reshape(-1, 3)
        comb, err, Q = eng.associate_single(n_persons, tracked, thr, 0.3, min_cams)
        t0 = time.time()
        with Pool(min(32, os.cpu_count())) as pool:
            refs = pool.map(ref_frame, [(per_cam, P, thr, min_cams) for per_cam in frames], chunksize=4)
        mis, worst_e, worst_q, none = 0, 0.0, 0.0, 0
        for f, (e, cb, q) in enumerate(refs):
            if not np.array_equal(comb[f], cb) or np.isinf(e) != np.isinf(err[f]):
                mis += 1
                if mis <= 3:
                    print('  frame', f, 'gpu', comb[f], err[f], 'ref', cb, e)
                continue
            if np.isinf(e):
                none += 1
                continue
            worst_e = max(worst_e, abs(err[f] - e) / max(1.0, abs(e)))
            worst_q = max(worst_q, float(np.abs(Q[f] - q).max()))
        print(f'C={C} distractors<={nd} thr={thr} min_cams={min_cams}: frames {F} mismatching choices {mis}, no solution {none}, max dErr {worst_e:.2e}, max dQ {worst_q:.2e}  oracle {time.time() - t0:.0f}s', flush=True)
