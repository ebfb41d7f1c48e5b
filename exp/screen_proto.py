"""Prototype (CPU, NumPy): how well does an fp32 evaluation in coordinates centred on the level-0 point predict the
fp64 reprojection error of the camera-subset candidates?  Decides the guards and the margin of the fused kernel's screen.
usage: python exp/screen_proto.py [frames] [C] [n_iter] [level] [p_outlier]"""
import itertools
import sys
import numpy as np
sys.path.insert(0, '.')
from pose2sim_amd import synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n_iter = int(sys.argv[3]) if len(sys.argv) > 3 else 3
level = int(sys.argv[4]) if len(sys.argv) > 4 else 1
p_out = float(sys.argv[5]) if len(sys.argv) > 5 else 0.03
K = 26
thr, lik_thr = 15.0, 0.3
wl = synth.make_config(F, C, K, 1, seed=2, p_outlier=p_out)
P = np.stack(wl['P'])                                   # [C][3][4]
xyl = wl['xyl'].astype(np.float64)[:, 0]                # [F][C][K][3]
x = xyl[..., 0].transpose(0, 2, 1).reshape(-1, C)       # [U][C]
y = xyl[..., 1].transpose(0, 2, 1).reshape(-1, C)
w = xyl[..., 2].transpose(0, 2, 1).reshape(-1, C)
valid = ~(np.isnan(w) | (w < lik_thr))
w = np.where(valid, w, 0.0); x = np.where(valid, x, 0.0); y = np.where(valid, y, 0.0)
U = x.shape[0]


def rows(x, y, w):
    A = (P[None, :, 0, :] - x[..., None] * P[None, :, 2, :]) * w[..., None]     # [U][C][4]
    B = (P[None, :, 1, :] - y[..., None] * P[None, :, 2, :]) * w[..., None]
    return A, B


def solve(N):
    ev, V = np.linalg.eigh(N)
    v = V[..., 0]
    with np.errstate(all='ignore'):
        return v[..., :3] / v[..., 3:4]


def errors(Q, x, y, keep):
    Qh = np.concatenate([Q, np.ones(Q.shape[:-1] + (1,))], -1)                  # [U][4]
    pr = np.einsum('cij,uj->uci', P, Qh)
    with np.errstate(all='ignore'):
        d = np.hypot(pr[..., 0] / pr[..., 2] - x, pr[..., 1] / pr[..., 2] - y)
        return (d * keep).sum(-1) / keep.sum(-1)


A, B = rows(x, y, w)
Nc = np.einsum('uci,ucj->ucij', A, A) + np.einsum('uci,ucj->ucij', B, B)        # per-camera contribution
N = Nc.sum(1)
Q0 = solve(N)
e0 = errors(Q0, x, y, valid)
nv = valid.sum(1)
hard = (e0 > thr) & (nv >= 2 + level)
print(f'C={C} level={level} n_iter={n_iter} p_outlier={p_out}: units {U}, searching {hard.sum()} ({hard.mean() * 100:.1f} %)')
idx = np.flatnonzero(hard)
xh, yh, wh, vh, Q0h, Nh, Nch = x[idx], y[idx], w[idx], valid[idx], Q0[idx], N[idx], Nc[idx]
H = idx.size
subsets = list(itertools.combinations(range(C), level))
S = len(subsets)
e64 = np.full((H, S), np.inf)
for si, sub in enumerate(subsets):
    keep = vh.copy(); keep[:, list(sub)] = False
    ok = vh[:, list(sub)].all(1) & (keep.sum(1) >= 2)
    Qj = solve(Nh - Nch[:, list(sub)].sum(1))
    e = errors(Qj, xh, yh, keep)
    e64[:, si] = np.where(ok, e, np.inf)

# ---- fp32 screen in coordinates centred on Q0 ---------------------------------------------------------------------
f32 = np.float32
M64 = Nh[:, :3, :3]
g64 = np.einsum('uij,uj->ui', M64, Q0h) + Nh[:, :3, 3]
h64 = np.einsum('ui,uij,uj->u', Q0h, M64, Q0h) + 2 * np.einsum('ui,ui->u', Nh[:, :3, 3], Q0h) + Nh[:, 3, 3]
M32, g32, h32 = M64.astype(f32), g64.astype(f32), h64.astype(f32)
P32 = P.astype(f32)
x32, y32, w32 = xh.astype(f32), yh.astype(f32), wh.astype(f32)
Q032 = Q0h.astype(f32)
Ar = P32[None, :, 0, :] - x32[..., None] * P32[None, :, 2, :]                    # [H][C][4] unweighted rows, fp32
Br = P32[None, :, 1, :] - y32[..., None] * P32[None, :, 2, :]
Q0h32 = np.concatenate([Q032, np.ones((H, 1), dtype=f32)], -1)
u0 = np.einsum('uci,ui->uc', Ar, Q0h32).astype(f32)                             # residual numerators at Q0, fp32 (cancellation)
v0 = np.einsum('uci,ui->uc', Br, Q0h32).astype(f32)
w2 = w32 * w32


def inv3_apply(Mm, r):
    m00, m01, m02, m11, m12, m22 = Mm[:, 0, 0], Mm[:, 0, 1], Mm[:, 0, 2], Mm[:, 1, 1], Mm[:, 1, 2], Mm[:, 2, 2]
    c00 = m11 * m22 - m12 * m12; c01 = m02 * m12 - m01 * m22; c02 = m01 * m12 - m02 * m11
    c11 = m00 * m22 - m02 * m02; c12 = m01 * m02 - m00 * m12; c22 = m00 * m11 - m01 * m01
    det = m00 * c00 + m01 * c01 + m02 * c02
    with np.errstate(all='ignore'):
        idet = f32(1) / det
    o0 = (c00 * r[:, 0] + c01 * r[:, 1] + c02 * r[:, 2]) * idet
    o1 = (c01 * r[:, 0] + c11 * r[:, 1] + c12 * r[:, 2]) * idet
    o2 = (c02 * r[:, 0] + c12 * r[:, 1] + c22 * r[:, 2]) * idet
    tr = m00 + m11 + m22
    pd = (m00 > 0) & (c22 > 0) & (det > 0)
    with np.errstate(all='ignore'):
        return np.stack([o0, o1, o2], -1), np.where(pd, det / (tr * tr * tr), f32(0))


e32 = np.full((H, S), np.inf, dtype=f32)
cond = np.zeros((H, S), dtype=f32)
dlam = np.zeros((H, S), dtype=f32)
eye = np.eye(3, dtype=f32)[None]
with np.errstate(all='ignore'):
    for si, sub in enumerate(subsets):
        keep = vh.copy(); keep[:, list(sub)] = False
        ok = vh[:, list(sub)].all(1) & (keep.sum(1) >= 2)
        Mj, gj, hj = M32.copy(), g32.copy(), h32.copy()
        for j in sub:
            a, b = Ar[:, j, :3], Br[:, j, :3]
            Mj = Mj - w2[:, j, None, None] * (a[:, :, None] * a[:, None, :] + b[:, :, None] * b[:, None, :])
            gj = gj - w2[:, j, None] * (a * u0[:, j, None] + b * v0[:, j, None])
            hj = hj - w2[:, j] * (u0[:, j] * u0[:, j] + v0[:, j] * v0[:, j])
        lam = np.zeros(H, dtype=f32)
        lam_prev = lam
        for it in range(n_iter):
            q, cd = inv3_apply(Mj - lam[:, None, None] * eye, lam[:, None] * Q032 - gj)
            Mq = np.einsum('uij,uj->ui', Mj, q)
            num = (Mq * q).sum(-1) + f32(2) * (gj * q).sum(-1) + hj
            Qn = Q032 + q
            den = (Qn * Qn).sum(-1) + f32(1)
            lam_prev = lam
            lam = (num / den).astype(f32)
        q, cd = inv3_apply(Mj - lam[:, None, None] * eye, lam[:, None] * Q032 - gj)
        cond[:, si] = cd
        dlam[:, si] = np.abs(lam - lam_prev) / np.maximum(np.abs(lam), f32(1e-30))
        Qn = np.concatenate([Q032 + q, np.ones((H, 1), dtype=f32)], -1)
        pr = np.einsum('cij,uj->uci', P32, Qn).astype(f32)
        uu = pr[..., 0] - x32 * pr[..., 2]
        vv = pr[..., 1] - y32 * pr[..., 2]
        d = np.sqrt(uu * uu + vv * vv) / np.abs(pr[..., 2])
        e = (d * keep).sum(-1) / keep.sum(-1).astype(f32)
        e32[:, si] = np.where(ok, e, np.inf)

fin = np.isfinite(e64)
with np.errstate(all='ignore'):
    diff = np.abs(e32.astype(np.float64) - e64)
diff = np.where(fin, np.nan_to_num(diff, nan=np.inf), 0.0)
print(f'candidates {fin.sum()}: |e32 - e64| max {diff[fin].max():.3e} px, p99.9 {np.quantile(diff[fin], 0.999):.3e}, median {np.median(diff[fin]):.3e}')
for cmin, dmax in ((1e-3, 1e-2), (1e-3, 1e-3), (3e-3, 1e-3), (1e-2, 1e-3)):
    g = fin & (cond >= cmin) & (dlam <= dmax) & np.isfinite(e32)
    dd = diff[g]
    rel = dd / np.maximum(e64[g], 1.0)
    k = np.argmax(dd) if dd.size else 0
    print(f'  guards cond >= {cmin}, dlam <= {dmax}: pass {g.sum() / fin.sum():.4f}; max |d| {dd.max():.3e} px (e64 {e64[g][k]:.1f}), '
          f'max rel {rel.max():.3e}; max |d| for e64 <= 2 thr: {dd[e64[g] <= 2 * thr].max() if (e64[g] <= 2 * thr).any() else 0:.3e}')
best = e64.min(1)
g = (cond >= 1e-3) & (dlam <= 1e-3) & np.isfinite(e32)
for m_abs, m_rel in ((0.02, 1e-3), (0.05, 2e-3), (0.25, 1e-2)):
    m = m_abs + m_rel * np.minimum(e32, 1e6)
    e_lo = np.where(g, e32 - m, -np.inf)                  # lower bound of the candidate's fp64 error (unguarded: unknown)
    e_hi = np.where(g, e32 + m, np.inf)
    min_hi = e_hi.min(1)
    surv = fin & (e_lo <= thr) & (e_lo <= min_hi[:, None])
    win = e64.argmin(1)
    print(f'margin {m_abs}+{m_rel}e: survivors per searching unit {surv.sum() / H:.3f}; units with no survivor {np.mean(~surv.any(1)):.3f}; '
          f'wrongly pruned winners {np.sum((best <= thr) & ~surv[np.arange(H), win])}')
print('level success rate', np.mean(best <= thr))
# margin model with the convergence indicator: m = 0.02 + e32 (1e-3 + c dlam)
gg = fin & (cond >= 3e-3) & np.isfinite(e32) & (dlam <= 0.25)
for c in (0.5, 1.0, 2.0, 4.0):
    m = 0.02 + e32 * (1e-3 + c * dlam)
    ratio = np.where(gg, diff / m, 0.0)
    e_lo = np.where(gg, e32 - m, -np.inf); e_hi = np.where(gg, e32 + m, np.inf)
    surv = fin & (e_lo <= thr) & (e_lo <= e_hi.min(1)[:, None])
    print(f'model c={c}: guarded {gg.sum() / fin.sum():.4f}; max diff/m {ratio.max():.3f}; survivors/unit {surv.sum() / H:.3f}; '
          f'wrongly pruned {np.sum((best <= thr) & ~surv[np.arange(H), e64.argmin(1)])}')

# ---- variant: one solve fewer.  q from lam1 (= n_iter 1), dlam measured by one more Rayleigh quotient at that q ----------
if len(sys.argv) > 6 and sys.argv[6] == 'short':
    with np.errstate(all='ignore'):
        e32s = np.full((H, S), np.inf, dtype=f32); dls = np.zeros((H, S), dtype=f32); conds = np.zeros((H, S), dtype=f32)
        for si, sub in enumerate(subsets):
            keep = vh.copy(); keep[:, list(sub)] = False
            ok = vh[:, list(sub)].all(1) & (keep.sum(1) >= 2)
            Mj, gj, hj = M32.copy(), g32.copy(), h32.copy()
            for j in sub:
                a, b = Ar[:, j, :3], Br[:, j, :3]
                Mj = Mj - w2[:, j, None, None] * (a[:, :, None] * a[:, None, :] + b[:, :, None] * b[:, None, :])
                gj = gj - w2[:, j, None] * (a * u0[:, j, None] + b * v0[:, j, None])
                hj = hj - w2[:, j] * (u0[:, j] * u0[:, j] + v0[:, j] * v0[:, j])
            def ray(q):
                Mq = np.einsum('uij,uj->ui', Mj, q)
                num = (Mq * q).sum(-1) + f32(2) * (gj * q).sum(-1) + hj
                Qn = Q032 + q
                return (num / ((Qn * Qn).sum(-1) + f32(1))).astype(f32)
            q, cd = inv3_apply(Mj, -gj)
            lam1 = ray(q)
            q, cd = inv3_apply(Mj - lam1[:, None, None] * eye, lam1[:, None] * Q032 - gj)
            lam2 = ray(q)
            conds[:, si] = cd
            dls[:, si] = np.abs(lam2 - lam1) / np.maximum(np.abs(lam2), f32(1e-30))
            Qn = np.concatenate([Q032 + q, np.ones((H, 1), dtype=f32)], -1)
            pr = np.einsum('cij,uj->uci', P32, Qn).astype(f32)
            uu = pr[..., 0] - x32 * pr[..., 2]; vv = pr[..., 1] - y32 * pr[..., 2]
            d = np.sqrt(uu * uu + vv * vv) / np.abs(pr[..., 2])
            e = (d * keep).sum(-1) / keep.sum(-1).astype(f32)
            e32s[:, si] = np.where(ok, e, np.inf)
        diffs = np.abs(e32s.astype(np.float64) - e64)
    diffs = np.where(fin, np.nan_to_num(diffs, nan=np.inf), 0.0)
    gg = fin & (conds >= 3e-3) & np.isfinite(e32s) & (dls <= 0.25)
    print(f'SHORT variant: |e32 - e64| max {diffs[fin].max():.3e}, p99.9 {np.quantile(diffs[fin], 0.999):.3e}')
    for c in (1.0, 2.0, 4.0, 8.0):
        m = 0.02 + e32s * (1e-3 + c * dls)
        ratio = np.where(gg, diffs / m, 0.0)
        e_lo = np.where(gg, e32s - m, -np.inf); e_hi = np.where(gg, e32s + m, np.inf)
        surv = fin & (e_lo <= thr) & (e_lo <= e_hi.min(1)[:, None])
        print(f'  short c={c}: guarded {gg.sum() / fin.sum():.4f}; max diff/m {ratio.max():.3f}; survivors/unit {surv.sum() / H:.3f}; '
              f'wrongly pruned {np.sum((best <= thr) & ~surv[np.arange(H), e64.argmin(1)])}')
