"""CPU oracle for the multi-person association of one frame -- TEST INFRASTRUCTURE.

NumPy restatement of the reference's per-frame body of associate_all
(personAssociation.py:783-801): Pluecker rays (:277-316), pairwise epipolar affinity (:347-408),
circular constraint (:411-428), matchSVT (:431-509) and the proposal extraction (:512-549).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product
never does.  Pinned by tests/test_oracle_golden.py against fixtures recorded from the reference
(tests/golden/make_golden_assoc.py).  numpy.linalg.svd / inv are the reference's own calls here
(LAPACK, as in the reference): no third-party substitution on this path.

Citations are relative to /root/reference/Pose2Sim/personAssociation.py.
"""
import itertools

import numpy as np


def rays_of_person(kpts, inv_K, R_mat, T):
    """compute_rays (:277-316).  kpts: flat [x0, y0, l0, x1, ...].  -> [Kj][7] = direction (3),
    moment (3), likelihood; a joint with any NaN becomes a zero row."""
    kp = np.asarray(kpts, dtype=np.float64).reshape(-1, 3)
    cam_center = -R_mat.T @ T
    out = np.zeros((kp.shape[0], 7))
    for i in range(kp.shape[0]):
        q = np.array([kp[i, 0], kp[i, 1], 1.0])
        norm_Q = R_mat.T @ (inv_K @ q - T)
        line = norm_Q - cam_center
        with np.errstate(invalid='ignore', divide='ignore'):
            d = line / np.linalg.norm(line)
        m = np.cross(cam_center, d)
        row = np.concatenate([d, m, [kp[i, 2]]])
        if not np.isnan(row).any():
            out[i] = row
    return out


def affinity_matrix(people_per_cam, cal, cum, recon_thr):
    """compute_affinity (:347-408) followed by the circular constraint product (:794-795)."""
    N = int(cum[-1])
    rays = [np.array([rays_of_person(p, cal['inv_K'][c], cal['R_mat'][c], cal['T'][c]) for p in people])
            for c, people in enumerate(people_per_cam)]
    dist = np.zeros((N, N)) + 2 * recon_thr
    for c0, c1 in itertools.combinations(range(len(people_per_cam)), 2):
        if cum[c0] == cum[c0 + 1] or cum[c1] == cum[c1 + 1]:
            continue
        p0 = rays[c0][:, None]
        p1 = rays[c1][None, :]
        prod = np.sum(p0[..., :3] * p1[..., 3:6], axis=-1) + np.sum(p1[..., :3] * p0[..., 3:6], axis=-1)   # :341
        lik = np.sqrt(p0[..., -1] * p1[..., -1])
        mwd = np.sum(np.abs(prod) * lik, axis=-1) / (1e-5 + lik.sum(axis=-1))                               # :394
        dist[cum[c0]:cum[c0 + 1], cum[c1]:cum[c1 + 1]] = mwd
        dist[cum[c1]:cum[c1 + 1], cum[c0]:cum[c0 + 1]] = mwd.T
    dist[dist > recon_thr] = recon_thr
    aff = 1 - dist / recon_thr
    return aff * circular_constraint(cum)


def circular_constraint(cum):
    """:411-428: identity within a view, ones between different views."""
    N = int(cum[-1])
    cc = np.identity(N)
    for i in range(len(cum) - 1):
        cc[cum[i]:cum[i + 1], cum[i + 1]:N] = 1
        cc[cum[i + 1]:N, cum[i]:cum[i + 1]] = 1
    return cc


def singular_value_threshold(M, t):
    """SVT (:431-447)."""
    U, s, Vt = np.linalg.svd(M)
    return U @ np.diag(np.maximum(s - t, 0)) @ Vt


def match_svt(affinity, cum, max_iter=20, w_rank=50, tol=1e-4, w_sparse=0.1):
    """matchSVT (:450-509)."""
    X = affinity.copy()
    N = X.shape[0]
    if N == 0:
        return X
    cc = circular_constraint(cum)
    idx = np.arange(N)
    X[idx, idx] = 0.0
    Y = np.zeros_like(X)
    W = w_sparse - X
    mu = 64
    for _ in range(max_iter):
        X0 = X.copy()
        Q = singular_value_threshold(X + Y * 1.0 / mu, w_rank / mu)
        X = Q - (W + Y) / mu
        for i in range(len(cum) - 1):
            X[cum[i]:cum[i + 1], cum[i]:cum[i + 1]] = 0
        X[idx, idx] = 1.0
        X[X < 0] = 0
        X[X > 1] = 1
        X = X * cc
        X = (X + X.T) / 2
        Y = Y + mu * (X - Q)
        pRes = np.linalg.norm(X - Q) / N
        dRes = mu * np.linalg.norm(X - X0) / N
        if pRes < tol and dRes < tol:
            break
        if pRes > 10 * dRes:
            mu = 2 * mu
        elif dRes > 10 * pRes:
            mu = mu / 2
    return X


def proposals_from_affinity(affinity, cum, min_cams):
    """person_index_per_cam (:512-549): per row the best person of every camera, unique rows ordered
    by multiplicity, rows that reuse a person dropped, rows seen by too few cameras dropped."""
    n_views = len(cum) - 1
    rows = []
    for r in range(affinity.shape[0]):
        row = []
        for c in range(n_views):
            block = affinity[r, cum[c]:cum[c + 1]]
            row.append(float(np.argmax(block)) if (len(block) > 0 and max(block) > 0) else -1.0)
        rows.append(row)
    props = np.array(rows, dtype=float)
    if props.size == 0:
        return np.array([])
    props, counts = np.unique(props, axis=0, return_counts=True)
    props = props[np.argsort(counts)[::-1]]
    props[props == -1] = np.nan
    keep = np.ones(props.shape[0], dtype=bool)
    for i in range(1, len(props)):
        keep[i] = ~np.any(props[i] == props[:i], axis=0).any()
    props = props[keep]
    seen = [np.count_nonzero(~np.isnan(p)) for p in props]
    return np.array([p for (n, p) in zip(seen, props) if n >= min_cams])


def associate_frame(people_per_cam, cal, recon_thr, min_affinity, min_cams):
    """One frame: -> (affinity before matchSVT, thresholded matchSVT result, proposals)."""
    cum = np.cumsum([0] + [len(p) for p in people_per_cam])
    aff = affinity_matrix(people_per_cam, cal, cum, recon_thr)
    out = match_svt(aff, cum)
    out = out.copy()
    out[out < min_affinity] = 0
    return aff, out, proposals_from_affinity(out, cum, min_cams)
