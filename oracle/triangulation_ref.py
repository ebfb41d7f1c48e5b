"""CPU oracle for the per-(frame x person x keypoint) robust triangulation -- TEST INFRASTRUCTURE.

This file is a loop-faithful NumPy restatement of the reference's algorithm.  It is the checker
for the HIP path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
it.  The product package (pose2sim_amd) never does and fails loudly without its HIP library.

Pinned: tests/test_oracle_golden.py checks every function here against fixtures generated in
the build container from the reference itself (tests/golden/make_golden.py, which imports
/root/reference with the stand-ins documented in tests/golden/ref_shim.py).

Third-party arithmetic: the reference calls cv2.SVDecomp (OpenCV one-sided Jacobi SVD, double).
OpenCV is absent from this image, so the oracle uses numpy.linalg.svd (LAPACK gesdd, double):
the same right singular vector up to sign, which cancels in Q = Vt[3,:3]/Vt[3,3].

All file:line citations are relative to /root/reference/Pose2Sim/.
"""
import itertools

import numpy as np


# --------------------------------------------------------------------------------------------
# common.py:327-354
def weighted_dlt(P_kept, x, y, w):
    """Weighted DLT: rows (P[0]-x*P[2])*w and (P[1]-y*P[2])*w per camera, smallest right singular
    vector, dehomogenised.  Fewer than 4 rows -> [nan, nan, nan, 1] (common.py:347-352).
    A non-finite system returns NaN (what a Jacobi SVD yields; LAPACK would raise)."""
    n = len(x)
    if 2 * n < 4:
        return np.array([np.nan, np.nan, np.nan, 1.0])
    A = np.empty((2 * n, 4))
    for c in range(n):
        Pc = P_kept[c]
        A[2 * c] = (Pc[0] - x[c] * Pc[2]) * w[c]
        A[2 * c + 1] = (Pc[1] - y[c] * Pc[2]) * w[c]
    if not np.isfinite(A).all():
        return np.array([np.nan, np.nan, np.nan, 1.0])
    _, _, Vt = np.linalg.svd(A, full_matrices=False)
    v = Vt[3]
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.array([v[0] / v[3], v[1] / v[3], v[2] / v[3], 1.0])


# common.py:357-375
def pinhole_reproject(P_kept, Qh):
    with np.errstate(divide='ignore', invalid='ignore'):
        u = [Pc[0] @ Qh / (Pc[2] @ Qh) for Pc in P_kept]
        v = [Pc[1] @ Qh / (Pc[2] @ Qh) for Pc in P_kept]
    return np.array(u), np.array(v)


# common.py:378-403 (1-D case: the only one the hot path uses)
def point_distance(a, b):
    """L2 distance with the reference's NaN rules: all-NaN difference -> inf, otherwise NaN
    components are dropped from the sum."""
    d = np.asarray(b, dtype=np.float64) - np.asarray(a, dtype=np.float64)
    if np.isnan(d).all():
        return np.inf
    with np.errstate(over='ignore'):
        return float(np.sqrt(np.nansum(d * d)))


def _mean_distance(xo, yo, u, v, count):
    """np.mean over the first `count` cameras of point_distance (triangulation.py:485-489, 559-561)."""
    if count == 0:
        return np.nan
    return float(np.mean([point_distance((xo[j], yo[j]), (u[j], v[j])) for j in range(count)]))


def _distorted_reproject(Q3, idx, cal, count):
    """cv2.projectPoints with the ORIGINAL K and distortion (triangulation.py:473, 535; quirk Q4)."""
    from pose2sim_amd import cvmath   # host-side camera model shared with the package (no GPU code)
    u = np.empty(count)
    v = np.empty(count)
    for j in range(count):
        c = idx[j]
        uv = cvmath.project_points(np.asarray(Q3, dtype=np.float64).reshape(1, 3),
                                   cal['R'][c], cal['T'][c], cal['K'][c], cal['dist'][c])
        u[j], v[j] = uv[0, 0], uv[0, 1]
    return u, v


# triangulation.py:363-604
def triangulate_unit(coords, coords_swapped, P, cal, thr, min_cams, lr_swap=False, undistort=False):
    """One unit.  coords, coords_swapped: (3, C) arrays (x, y, likelihood), already NaN where the
    likelihood was below the threshold (triangulation.py:817-821).  P: list of C 3x4 arrays.
    cal: dict with 'K','dist','R','T' lists (only read when undistort).

    Returns Q (3,), error (float, NaN when rejected), nb_cams_excluded (int),
    id_excluded_cams (int array).
    """
    x_all, y_all, l_all = (np.asarray(r, dtype=np.float64) for r in coords)
    xs_all, ys_all = (np.asarray(r, dtype=np.float64) for r in coords_swapped[:2])
    C = len(x_all)
    err_min = np.inf
    best = None
    ids_committed = None
    nb_excl = None
    Q = None

    level = 0
    while err_min > thr and C - level >= min_cams:                       # :408
        removal_sets = list(itertools.combinations(range(C), level))      # :411 (all C cameras)
        lik_sub = []
        for rs in removal_sets:                                          # :420-432
            l = l_all.copy()
            l[list(rs)] = np.nan
            lik_sub.append(l)
        ids_new = [np.flatnonzero(np.isnan(l)) for l in lik_sub]          # :435
        n_excl = [int(np.count_nonzero(np.nan_to_num(l) == 0)) for l in lik_sub]   # :436 NaN or 0
        n_off_tot = max(n_excl)                                          # :437 MAX over subsets (Q2)
        if n_off_tot > C - min_cams:                                     # :440
            break
        ids_committed = ids_new                                          # :442

        kept_idx, Qs, errs = [], [], []
        for l in lik_sub:                                                # :445-489
            keep = np.flatnonzero(~np.isnan(l) & (l != 0.0))
            kept_idx.append(keep)
            Pk = [P[c] for c in keep]
            Qh = weighted_dlt(Pk, x_all[keep], y_all[keep], l[keep])
            if undistort:
                u, v = _distorted_reproject(Qh[:3], keep, cal, len(keep))
            else:
                u, v = pinhole_reproject(Pk, Qh)
            Qs.append(Qh)
            errs.append(_mean_distance(x_all[keep], y_all[keep], u, v, len(keep)))
        errs = np.array(errs, dtype=np.float64)
        err_min = float(np.nanmin(errs))                                 # :500
        best = int(np.nanargmin(errs))                                   # :502 first index on ties
        nb_excl = n_excl[best]                                           # :503
        Q = Qs[best][:3]                                                 # :505

        if lr_swap and err_min > thr:                                    # :509-579 (quirk Q3)
            M = C - n_off_tot
            n_sw = 1
            e_sw_min = err_min
            sw_best = None
            Q_sw_best = None
            while e_sw_min > thr and n_sw < M / 2:                       # :513
                # The reference replicates REFERENCES to one array per off-configuration
                # (:518-519), so after the in-place writes (:525-526) every "swap subset" candidate
                # is the same vector: the first M kept cameras all carry the mirrored keypoint.
                e_sw, Q_sw = [], []
                for i, keep in enumerate(kept_idx):
                    xk, yk = x_all[keep].copy(), y_all[keep].copy()
                    xk[:M] = xs_all[keep][:M]
                    yk[:M] = ys_all[keep][:M]
                    Pk = [P[c] for c in keep]
                    Qh = weighted_dlt(Pk, xk, yk, lik_sub[i][keep])      # original likelihoods (:529)
                    if undistort:
                        u, v = _distorted_reproject(Qh[:3], keep, cal, M)   # :535-536 first M only
                    else:
                        u, v = pinhole_reproject(Pk, Qh)
                    Q_sw.append(Qh)
                    e_sw.append(_mean_distance(xk, yk, u, v, M))         # :559-561 first M only
                e_sw = np.array(e_sw, dtype=np.float64)
                e_sw_min = float(np.min(e_sw))                           # :567
                sw_best = int(np.argmin(e_sw))                           # :568
                Q_sw_best = Q_sw[sw_best][:3]
                n_sw += 1
            if e_sw_min < err_min:                                       # :576-579 nb_excl NOT updated
                err_min = e_sw_min
                best = sw_best
                Q = Q_sw_best
        level += 1                                                       # :583

    if best is not None:                                                 # :588-596
        id_excl = np.asarray(ids_committed[best], dtype=np.int64)
    else:
        id_excl = np.arange(C, dtype=np.int64)
        nb_excl = C
    if err_min > thr:                                                    # :600-602
        err_min = np.nan
        Q = np.array([np.nan, np.nan, np.nan])
    return np.asarray(Q, dtype=np.float64), err_min, int(nb_excl), id_excl


# --------------------------------------------------------------------------------------------
def excluded_mask(id_excl):
    m = 0
    for c in id_excl:
        m |= 1 << int(c)
    return m


def triangulate_batch(xyl, P, cal, swap_idx, lik_thr, thr, min_cams, lr_swap=False, undistort=False):
    """Frame loop of triangulate_all (triangulation.py:796-845) on a packed tensor.

    xyl: float array [F][Pn][C][K][3] (NaN = missing).  Returns
    Q [F][Pn][K][3] f64, err [F][Pn][K] f64, n_excl [F][Pn][K] i32, mask [F][Pn][K] u32.
    """
    from pose2sim_amd import cvmath
    xyl = np.asarray(xyl, dtype=np.float64)
    F, Pn, C, K, _ = xyl.shape
    Qo = np.full((F, Pn, K, 3), np.nan)
    eo = np.full((F, Pn, K), np.nan)
    no = np.zeros((F, Pn, K), dtype=np.int32)
    mo = np.zeros((F, Pn, K), dtype=np.uint32)
    for f in range(F):
        for n in range(Pn):
            x = xyl[f, n, :, :, 0].copy()      # [C][K]
            y = xyl[f, n, :, :, 1].copy()
            l = xyl[f, n, :, :, 2].copy()
            if undistort:                      # :808-813 (float32 round trip inside)
                for c in range(C):
                    pts = cvmath.undistort_points(np.stack([x[c], y[c]], axis=-1),
                                                  cal['K'][c], cal['dist'][c], cal['optim_K'][c])
                    x[c], y[c] = pts[:, 0], pts[:, 1]
            with np.errstate(invalid='ignore'):  # :817-821
                low = l < lik_thr
            x[low] = np.nan
            y[low] = np.nan
            l[low] = np.nan
            for k in range(K):
                ks = swap_idx[k]
                Q, e, ne, ids = triangulate_unit(
                    np.array([x[:, k], y[:, k], l[:, k]]),
                    np.array([x[:, ks], y[:, ks], l[:, ks]]),
                    P, cal, thr, min_cams, lr_swap, undistort)
                Qo[f, n, k] = Q
                eo[f, n, k] = e
                no[f, n, k] = ne
                mo[f, n, k] = excluded_mask(ids)
    return Qo, eo, no, mo
