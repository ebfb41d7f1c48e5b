"""CPU oracle for the single-person association of one frame -- TEST INFRASTRUCTURE.

NumPy restatement of personAssociation.py:67-99 (persons_combinations), :102-151 (triangulate_comb,
pinhole branch) and :154-257 (best_persons_and_cameras_combination), on in-memory keypoints instead of
JSON paths.  Only tests/ may import it.  Pinned by tests/test_oracle_golden.py against per-frame
results recorded from the reference (tests/golden/make_golden_e2e_single.py).

Order-dependent behaviour reproduced: combinations are visited in itertools.product order; a
combination's cameras whose tracked-keypoint likelihood is below the threshold are switched off IN
PLACE (the row stays mutated for later levels, :212-213); `error_min` keeps the value of the last
combination evaluated (not the best one) and drives the while condition (:192, :233); the scan of a
level stops at the first combination below the threshold (:242-243); the best solution is kept
across levels with a strict '<' (:237).
"""
import itertools

import numpy as np

from oracle.triangulation_ref import pinhole_reproject, point_distance, weighted_dlt


def persons_combinations(n_persons_per_cam):
    """:67-99 from the per-camera person counts."""
    n = [int(x) for x in n_persons_per_cam]
    no_detect = [i for i, x in enumerate(n) if x == 0]
    n = [x if x != 0 else 1 for x in n]
    comb = np.array(list(itertools.product(*[range(x) for x in n])), float)
    comb[:, no_detect] = np.nan
    return comb


def triangulate_comb(comb, coords, P_all):
    """:102-151, pinhole branch."""
    keep = [i for i in range(len(comb)) if not np.isnan(comb[i])]
    Pk = [P_all[i] for i in keep]
    try:
        x, y, l = np.array([coords[i] for i in keep]).T
        Qh = weighted_dlt(Pk, x, y, l)
    except Exception:
        x = y = np.array([])
        Qh = np.array([np.nan, np.nan, np.nan, 1.0])
    u, v = pinhole_reproject(Pk, Qh)
    errs = [point_distance((x[c], y[c]), (u[c], v[c])) for c in range(len(u))]
    with np.errstate(invalid='ignore'):
        e = float(np.mean(errs)) if len(errs) else np.nan
    return e, comb, Qh


def best_persons_and_cameras(people_per_cam, combos, P_all, kid, thr, min_cams, lik_thr):
    """:154-257.  people_per_cam[c] = list of flat keypoint lists (read_json order).
    combos is modified in place like the reference's array.  -> (best_error, best_comb, best_Q[:3])."""
    C = len(people_per_cam)
    error_min = np.inf
    missing = int(np.all(np.isnan(combos), axis=0).sum())
    extra = 0
    best_error, best_comb, best_Q = np.inf, None, None
    while error_min > thr and C - (missing + extra) >= min_cams:
        for combination in combos:
            coords = []
            for c, pid in enumerate(combination):
                try:
                    coords.append(list(people_per_cam[c][int(pid)][kid * 3:kid * 3 + 3]))
                except Exception:
                    coords.append([np.nan, np.nan, np.nan])
            coords = np.array(coords, dtype=float)
            with np.errstate(invalid='ignore'):
                coords[:, 2][coords[:, 2] < lik_thr] = 0.0
            combination[coords[:, 2] == 0.0] = np.nan                      # in place (:213)
            active = np.where(~np.isnan(combination))[0]
            if len(active) < min_cams:
                continue
            errs, combs, Qs = [], [], []
            for off in itertools.combinations(active, extra):
                cb = combination.copy()
                cb[list(off)] = np.nan
                e, cb, Qh = triangulate_comb(cb, coords, P_all)
                errs.append(e); combs.append(cb); Qs.append(Qh)
            if np.all(np.isnan(errs)):
                continue
            error_min = np.nanmin(errs)
            i = int(np.argmin(errs))
            if error_min < best_error:
                best_error, best_comb, best_Q = error_min, combs[i], Qs[i]
            if error_min < thr:
                break
        extra += 1
    if best_comb is None:
        return np.inf, np.array([np.nan] * C), np.array([np.nan, np.nan, np.nan])
    return float(best_error), np.asarray(best_comb, dtype=float), np.asarray(best_Q, dtype=float)[:3]
