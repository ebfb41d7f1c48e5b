/*
 * tri_oracle.c -- C restatement of the reference's robust triangulation: TEST INFRASTRUCTURE.
 *
 * Same algorithm as oracle/triangulation_ref.py (which is pinned to fixtures recorded from the
 * reference), compiled so that parity tests and bench.py's cpu_baseline leg can run 10^5..10^6
 * units in seconds.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use
 * it; the product (pose2sim_amd) never links or loads it.
 *
 * Follows /root/reference/Pose2Sim/triangulation.py:363-604 (subset search, every C(C,k)
 * subset in itertools.combinations order, duplicates included), :808-821 (undistort + mask) and
 * common.py:327-403 (weighted DLT by SVD, reprojection, euclidean_distance).  The SVD is a
 * one-sided Jacobi (Hestenes) SVD in double, the same family as OpenCV's cv2.SVDecomp.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXC 32
#define CAL_STRIDE 30 /* fx fy cx cy | k1 k2 p1 p2 k3 | R[9] | T[3] | newK[9] */

/* right singular vector of the smallest singular value of A (m x 4, row-major), m >= 4 */
static void smallest_right_singular_vector(const double *A, int m, double v[4]) {
    double W[2 * MAXC * 4];
    double V[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    memcpy(W, A, sizeof(double) * (size_t)m * 4);
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                double a = 0, b = 0, g = 0;
                for (int i = 0; i < m; ++i) {
                    a += W[i * 4 + p] * W[i * 4 + p];
                    b += W[i * 4 + q] * W[i * 4 + q];
                    g += W[i * 4 + p] * W[i * 4 + q];
                }
                if (fabs(g) <= 1e-300 || fabs(g) <= 2.2e-16 * sqrt(a * b)) continue;
                rotated = 1;
                const double zeta = (b - a) / (2.0 * g);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < m; ++i) {
                    const double wp = W[i * 4 + p], wq = W[i * 4 + q];
                    W[i * 4 + p] = c * wp - s * wq;
                    W[i * 4 + q] = s * wp + c * wq;
                }
                for (int i = 0; i < 4; ++i) {
                    const double vp = V[i * 4 + p], vq = V[i * 4 + q];
                    V[i * 4 + p] = c * vp - s * vq;
                    V[i * 4 + q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    int jmin = 0;
    double nmin = INFINITY;
    for (int j = 0; j < 4; ++j) {
        double n = 0;
        for (int i = 0; i < m; ++i) n += W[i * 4 + j] * W[i * 4 + j];
        if (n < nmin) { nmin = n; jmin = j; }
    }
    for (int i = 0; i < 4; ++i) v[i] = V[i * 4 + jmin];
}

/* common.py:327-354 */
static void weighted_dlt(const double *P, const int *cams, int n, const double *x, const double *y,
                         const double *w, double Q[3]) {
    Q[0] = Q[1] = Q[2] = NAN;
    if (2 * n < 4) return;
    double A[2 * MAXC * 4];
    int finite = 1;
    for (int j = 0; j < n; ++j) {
        const double *Pc = P + 12 * cams[j];
        for (int i = 0; i < 4; ++i) {
            A[(2 * j) * 4 + i] = (Pc[i] - x[j] * Pc[8 + i]) * w[j];
            A[(2 * j + 1) * 4 + i] = (Pc[4 + i] - y[j] * Pc[8 + i]) * w[j];
            if (!isfinite(A[(2 * j) * 4 + i]) || !isfinite(A[(2 * j + 1) * 4 + i])) finite = 0;
        }
    }
    if (!finite) return;
    double v[4];
    smallest_right_singular_vector(A, 2 * n, v);
    Q[0] = v[0] / v[3]; Q[1] = v[1] / v[3]; Q[2] = v[2] / v[3];
}

/* common.py:378-403 for one 2D pair */
static double pair_distance(double dx, double dy) {
    const int nx = isnan(dx), ny = isnan(dy);
    if (nx && ny) return INFINITY;
    return sqrt((nx ? 0.0 : dx * dx) + (ny ? 0.0 : dy * dy));
}

static void project(const double *P, const double *cal, int c, int undistort, const double Q[3], double *u,
                    double *v) {
    if (!undistort) { /* common.py:357-375 */
        const double *Pc = P + 12 * c;
        const double a = Pc[0] * Q[0] + Pc[1] * Q[1] + Pc[2] * Q[2] + Pc[3];
        const double b = Pc[4] * Q[0] + Pc[5] * Q[1] + Pc[6] * Q[2] + Pc[7];
        const double z = Pc[8] * Q[0] + Pc[9] * Q[1] + Pc[10] * Q[2] + Pc[11];
        *u = a / z; *v = b / z;
        return;
    }
    /* cv2.projectPoints with the original K / distortion (triangulation.py:473) */
    const double *k = cal + CAL_STRIDE * c;
    const double *R = k + 9, *T = k + 18;
    const double X = R[0] * Q[0] + R[1] * Q[1] + R[2] * Q[2] + T[0];
    const double Y = R[3] * Q[0] + R[4] * Q[1] + R[5] * Q[2] + T[1];
    double Z = R[6] * Q[0] + R[7] * Q[1] + R[8] * Q[2] + T[2];
    if (Z == 0.0) Z = 1.0;
    const double x = X / Z, y = Y / Z, r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    const double cd = 1 + k[4] * r2 + k[5] * r4 + k[8] * r6;
    *u = (x * cd + k[6] * a1 + k[7] * a2) * k[0] + k[2];
    *v = (y * cd + k[6] * a3 + k[7] * a1) * k[1] + k[3];
}

/* cv2.undistortPoints on float32 input, float32 output (triangulation.py:810-813) */
static void undistort(const double *cal, int c, double *px, double *py) {
    const double *k = cal + CAL_STRIDE * c;
    const double *nk = k + 21;
    const double u = (double)(float)*px, v = (double)(float)*py;
    const double ifx = 1.0 / k[0], ify = 1.0 / k[1];
    const double x0 = (u - k[2]) * ifx, y0 = (v - k[3]) * ify;
    double x = x0, y = y0;
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        const double ic = 1.0 / (1 + ((k[8] * r2 + k[5]) * r2 + k[4]) * r2);
        if (ic < 0) { x = x0; y = y0; break; }
        const double dx = ((2 * k[6]) * x) * y + k[7] * (r2 + (2 * x) * x);
        const double dy = k[6] * (r2 + (2 * y) * y) + ((2 * k[7]) * x) * y;
        x = (x0 - dx) * ic; y = (y0 - dy) * ic;
    }
    const double xx = nk[0] * x + nk[1] * y + nk[2], yy = nk[3] * x + nk[4] * y + nk[5];
    const double ww = 1.0 / (nk[6] * x + nk[7] * y + nk[8]);
    *px = (double)(float)(xx * ww); *py = (double)(float)(yy * ww);
}

static int next_combination(int *idx, int k, int n) {
    int i = k - 1;
    while (i >= 0 && idx[i] == n - k + i) --i;
    if (i < 0) return 0;
    ++idx[i];
    for (int j = i + 1; j < k; ++j) idx[j] = idx[j - 1] + 1;
    return 1;
}

/* triangulation.py:363-604 for one unit; x, y, l (and swapped xs, ys) already masked */
static void triangulate_unit(int C, const double *x, const double *y, const double *l, const double *xs,
                             const double *ys, const double *P, const double *cal, double thr, int min_cams,
                             int lr_swap, int undist, double Qout[3], double *err_out, int *nexcl_out,
                             uint32_t *mask_out) {
    double err_min = INFINITY, Q[3] = {NAN, NAN, NAN};
    int have_best = 0, nb_excl = C;
    uint32_t best_mask = 0;
    for (int level = 0; err_min > thr && C - level >= min_cams; ++level) {
        int idx[MAXC];
        for (int i = 0; i < level; ++i) idx[i] = i;
        /* pass 1: max exclusion count over the subsets (:436-441) */
        int n_off_tot = 0;
        do {
            uint32_t S = 0;
            for (int i = 0; i < level; ++i) S |= 1u << idx[i];
            int ne = 0;
            for (int c = 0; c < C; ++c) ne += ((S >> c) & 1u) || isnan(l[c]) || l[c] == 0.0;
            if (ne > n_off_tot) n_off_tot = ne;
        } while (level > 0 && next_combination(idx, level, C));
        if (n_off_tot > C - min_cams) break;
        const int M = C - n_off_tot;
        /* pass 2: evaluate every subset */
        for (int i = 0; i < level; ++i) idx[i] = i;
        double lvl_err = INFINITY, lvl_Q[3] = {NAN, NAN, NAN};
        double sw_err = INFINITY, sw_Q[3] = {NAN, NAN, NAN};
        int lvl_first = 1, sw_first = 1, lvl_ne = 0;
        uint32_t lvl_mask = 0, sw_mask = 0;
        do {
            uint32_t S = 0, nanm = 0;
            for (int i = 0; i < level; ++i) S |= 1u << idx[i];
            int cams[MAXC], n = 0, ne = 0;
            double xk[MAXC], yk[MAXC], wk[MAXC], xw[MAXC], yw[MAXC];
            for (int c = 0; c < C; ++c) {
                const int removed = (S >> c) & 1u;
                const double lc = removed ? NAN : l[c];
                if (isnan(lc)) nanm |= 1u << c;
                if (isnan(lc) || lc == 0.0) { ++ne; continue; }
                cams[n] = c; xk[n] = x[c]; yk[n] = y[c]; wk[n] = lc; xw[n] = xs[c]; yw[n] = ys[c];
                ++n;
            }
            double q[3];
            weighted_dlt(P, cams, n, xk, yk, wk, q);
            double sum = 0;
            for (int j = 0; j < n; ++j) {
                double u, v;
                project(P, cal, cams[j], undist, q, &u, &v);
                sum += pair_distance(u - xk[j], v - yk[j]);
            }
            const double e = sum / n;
            if (lvl_first || e < lvl_err) {
                lvl_err = e; memcpy(lvl_Q, q, sizeof q); lvl_ne = ne; lvl_mask = nanm; lvl_first = 0;
            }
            if (lr_swap && M > 2) { /* quirk Q3: first M kept cameras carry the mirrored keypoint */
                double x2[MAXC], y2[MAXC];
                for (int j = 0; j < n; ++j) { x2[j] = j < M ? xw[j] : xk[j]; y2[j] = j < M ? yw[j] : yk[j]; }
                double qs[3];
                weighted_dlt(P, cams, n, x2, y2, wk, qs);
                double s2 = 0;
                for (int j = 0; j < M; ++j) {
                    double u, v;
                    project(P, cal, cams[j], undist, qs, &u, &v);
                    s2 += pair_distance(u - x2[j], v - y2[j]);
                }
                const double es = s2 / M;
                if (sw_first || es < sw_err) { sw_err = es; memcpy(sw_Q, qs, sizeof qs); sw_mask = nanm; sw_first = 0; }
            }
        } while (level > 0 && next_combination(idx, level, C));
        err_min = lvl_err; memcpy(Q, lvl_Q, sizeof Q); nb_excl = lvl_ne; best_mask = lvl_mask; have_best = 1;
        if (lr_swap && err_min > thr && M > 2 && sw_err < err_min) {
            err_min = sw_err; memcpy(Q, sw_Q, sizeof Q); best_mask = sw_mask; /* nb_excl kept (:576-579) */
        }
    }
    if (!have_best) { best_mask = C == 32 ? 0xffffffffu : ((1u << C) - 1u); nb_excl = C; }
    if (!(err_min <= thr)) { err_min = NAN; Q[0] = Q[1] = Q[2] = NAN; }
    memcpy(Qout, Q, sizeof Q);
    *err_out = err_min; *nexcl_out = nb_excl; *mask_out = best_mask;
}

/* xyl [n_blocks][C][K][3] float64; outputs per unit (block*K + k). */
int tri_oracle_batch(int64_t n_blocks, int C, int K, const double *xyl, const int32_t *swap_idx, const double *P,
                     const double *cal, double thr, double lik_thr, int min_cams, int lr_swap, int undist,
                     double *Q, double *err, int32_t *n_excl, uint32_t *mask, int n_threads) {
    if (C < 1 || C > MAXC || K < 1 || (undist && !cal)) return -1;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t b = 0; b < n_blocks; ++b) {
        double *blk = (double *)malloc(sizeof(double) * (size_t)C * K * 3);
        memcpy(blk, xyl + b * C * K * 3, sizeof(double) * (size_t)C * K * 3);
        for (int c = 0; c < C; ++c)
            for (int k = 0; k < K; ++k) {
                double *p = blk + ((size_t)c * K + k) * 3;
                if (undist) undistort(cal, c, &p[0], &p[1]);
                if (p[2] < lik_thr) p[0] = p[1] = p[2] = NAN;
            }
        for (int k = 0; k < K; ++k) {
            double x[MAXC], y[MAXC], l[MAXC], xs[MAXC], ys[MAXC];
            const int ks = swap_idx ? swap_idx[k] : k;
            for (int c = 0; c < C; ++c) {
                const double *p = blk + ((size_t)c * K + k) * 3, *s = blk + ((size_t)c * K + ks) * 3;
                x[c] = p[0]; y[c] = p[1]; l[c] = p[2]; xs[c] = s[0]; ys[c] = s[1];
            }
            int ne;
            triangulate_unit(C, x, y, l, xs, ys, P, cal, thr, min_cams, lr_swap, undist, Q + (b * K + k) * 3,
                             err + b * K + k, &ne, mask + b * K + k);
            n_excl[b * K + k] = ne;
        }
        free(blk);
    }
    return 0;
}
