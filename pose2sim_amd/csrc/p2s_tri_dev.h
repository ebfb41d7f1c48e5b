// p2s_tri_dev.h -- device-side building blocks shared by the triangulation kernels (p2s_tri.hip: streaming,
// tiled and work-list search kernels; p2s_tri_fused.hip: the one-launch kernel with the in-wave subset search).
// Everything here is static / inline: each translation unit gets its own copy.
#ifndef P2S_TRI_DEV_H
#define P2S_TRI_DEV_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "p2s_internal.h"

namespace {


constexpr double kInf = __builtin_huge_val();

// np.nanargmin's order on the candidates (error, rank) of a level (triangulation.py:500-503): a number beats a NaN, then
// the lower error, then the lower rank (the first index); rank 0xffffffff = no candidate yet.  Among NaNs only the
// lowest rank stays (np.nanargmin raises when every error is NaN; the unit then ends as not triangulated either way).
__device__ __forceinline__ bool better_candidate(double e, uint32_t r, double be, uint32_t br) {
    const bool en = !(e == e), bn = !(be == be);
    return (r != 0xffffffffu) && ((br == 0xffffffffu) || (bn && !en) || (en == bn && (e < be || ((e == be || en) && r < br))));
}

// Calibration is read-only for the whole launch: going through the constant address space lets
// every uniform-index access become a scalar (SMEM) load and the value an SGPR operand.
typedef const __attribute__((address_space(4))) P2sCam *cam_cptr;


__device__ __forceinline__ double d_nan() { return __builtin_nan(""); }

// 1/d: v_rcp_f64 seed (4.6e-8 relative on gfx950, exp/seed_precision.hip) + one Newton step = 2e-15 -- a second step
// buys nothing the results can show.  0 -> inf, inf -> 0 and NaN pass through the seed unchanged.
__device__ __forceinline__ double fast_rcp(double d) {
    const double r0 = __builtin_amdgcn_rcp(d);
    const double e = fma(-d, r0, 1.0);
    const double r = fma(r0, e, r0);
    return (e == e) ? r : r0;
}

__device__ __forceinline__ double p2s_inline_rcp(double d) { return fast_rcp(d); }

// sqrt(s), s >= 0, to ~1 ulp: v_rsq_f64 seed, one Goldschmidt step, one residual correction.
__device__ __forceinline__ double fast_sqrt(double s) {
    const double y = __builtin_amdgcn_rsq(s);
    double g = s * y;
    double h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, s);
    g = fma(d, h, g);
    return (s == 0.0 || s == kInf) ? s : g;
}

// --------------------------------------------------------------------------------------------
// Normal-matrix contribution of one camera: rows (P0 - x P2) w and (P1 - y P2) w
// (common.py:344-345).  N is the upper triangle of the 4x4 A^T A: [00 01 02 03 11 12 13 22 23 33].
// PT is either an SGPR-backed constant pointer or an LDS pointer.  SIGN = -1 removes a camera.
template <int SIGN, typename PT>
__device__ __forceinline__ void accum_camera(double N[10], PT P, double x, double y, double w) {
    const double xw = x * w, yw = y * w;
    const double a0 = fma(-xw, P[8], P[0] * w), a1 = fma(-xw, P[9], P[1] * w), a2 = fma(-xw, P[10], P[2] * w),
                 a3 = fma(-xw, P[11], P[3] * w);
    const double b0 = fma(-yw, P[8], P[4] * w), b1 = fma(-yw, P[9], P[5] * w), b2 = fma(-yw, P[10], P[6] * w),
                 b3 = fma(-yw, P[11], P[7] * w);
    const double s0 = SIGN * a0, s1 = SIGN * a1, s2 = SIGN * a2, s3 = SIGN * a3;
    const double t0 = SIGN * b0, t1 = SIGN * b1, t2 = SIGN * b2, t3 = SIGN * b3;
    N[0] = fma(s0, a0, fma(t0, b0, N[0]));
    N[1] = fma(s0, a1, fma(t0, b1, N[1]));
    N[2] = fma(s0, a2, fma(t0, b2, N[2]));
    N[3] = fma(s0, a3, fma(t0, b3, N[3]));
    N[4] = fma(s1, a1, fma(t1, b1, N[4]));
    N[5] = fma(s1, a2, fma(t1, b2, N[5]));
    N[6] = fma(s1, a3, fma(t1, b3, N[6]));
    N[7] = fma(s2, a2, fma(t2, b2, N[7]));
    N[8] = fma(s2, a3, fma(t2, b3, N[8]));
    N[9] = fma(s3, a3, fma(t3, b3, N[9]));
}

// One step of iterative refinement for smallest_eigvec (see the end of that function): out of line and with
// scalar arguments (a pointer argument of a non-inlined function would force the matrix into scratch), so
// that the rare call leaves the register allocation of the hot loop alone.
__device__ __noinline__ double3 refine_eigvec(double n00, double m01, double m02, double b0, double n11, double m12,
                                              double b1, double n22, double b2, double c, double lam, double q0,
                                              double q1, double q2) {
    const double m00 = n00 - lam, m11 = n11 - lam, m22 = n22 - lam;
    const double c00 = fma(m11, m22, -m12 * m12);
    const double c01 = fma(m02, m12, -m01 * m22);
    const double c02 = fma(m01, m12, -m02 * m11);
    const double c11 = fma(m00, m22, -m02 * m02);
    const double c12 = fma(m01, m02, -m00 * m12);
    const double c22 = fma(m00, m11, -m01 * m01);
    const double det = fma(m00, c00, fma(m01, c01, m02 * c02));
    const double nid = -fast_rcp(det);
    const double r0 = fma(m00, q0, fma(m01, q1, fma(m02, q2, b0)));     // residual of (M - lambda) q = -b
    const double r1 = fma(m01, q0, fma(m11, q1, fma(m12, q2, b1)));
    const double r2 = fma(m02, q0, fma(m12, q1, fma(m22, q2, b2)));
    double x0 = fma((c00 * r0 + c01 * r1 + c02 * r2), nid, q0);
    double x1 = fma((c01 * r0 + c11 * r1 + c12 * r2), nid, q1);
    double x2 = fma((c02 * r0 + c12 * r1 + c22 * r2), nid, q2);
    const double p0 = -(c00 * x0 + c01 * x1 + c02 * x2) * nid;          // dq/dlambda at the refined point
    const double p1 = -(c01 * x0 + c11 * x1 + c12 * x2) * nid;
    const double p2 = -(c02 * x0 + c12 * x1 + c22 * x2) * nid;
    const double f = c + (b0 * x0 + b1 * x1 + b2 * x2) - lam;
    const double dl = f * fast_rcp(1.0 + (x0 * x0 + x1 * x1 + x2 * x2));
    const bool ok = det > 0.0;
    double3 r;
    r.x = ok ? fma(dl, p0, x0) : q0;
    r.y = ok ? fma(dl, p1, x1) : q1;
    r.z = ok ? fma(dl, p2, x2) : q2;
    return r;
}

// Smallest eigenvector of the 4x4 SPD matrix N, dehomogenised: v = (q, 1), N v = lambda v, i.e.
// q = V[0:3,3]/V[3,3] of the reference's SVD of A (common.py:348-350), N = A^T A.
//
// With N = [[M, b], [b^T, c]]: (M - lambda I) q = -b and f(lambda) = c - lambda + b.q(lambda) = 0.
// On (-inf, mu_1) (mu_1 = smallest eigenvalue of M) f is decreasing and concave and its only root
// there is the smallest eigenvalue of N (interlacing).  Halley's iteration from lambda = 0 -- the
// inhomogeneous least-squares point -- converges cubically; an iterate that jumps over the pole
// mu_1 (M - lambda I no longer positive definite) is pulled back by bisection, so the SMALLEST
// root is the one found.  The 3x3 systems go through the adjugate (one reciprocal).  The loop
// stops on a first-order bound of the error of q and returns q + dlambda * dq/dlambda; agreement
// with the SVD is ~1e-10 relative (tests/test_tri_gpu.py).
// One evaluation of the secular function at lambda: q(lambda) = -(M - lambda)^-1 b through the adjugate, its
// derivative p = dq/dlambda = (M - lambda)^-1 q, f = c - lambda + b.q and the Halley step of f (Newton's when the
// Halley denominator is not positive).
struct EigPass {
    double y0, y1, y2, p0, p1, p2;     // q(lambda), dq/dlambda
    double f, qq, qp, pp, nid, tr, step;
    bool pd;                           // M - lambda I positive definite
};

__device__ __forceinline__ void eig_pass(const double N[10], double lam, EigPass &o) {
    const double b0 = N[3], b1 = N[6], b2 = N[8], c = N[9];
    const double m01 = N[1], m02 = N[2], m12 = N[5];
    const double m00 = N[0] - lam, m11 = N[4] - lam, m22 = N[7] - lam;
    const double c00 = fma(m11, m22, -m12 * m12);
    const double c01 = fma(m02, m12, -m01 * m22);
    const double c02 = fma(m01, m12, -m02 * m11);
    const double c11 = fma(m00, m22, -m02 * m02);
    const double c12 = fma(m01, m02, -m00 * m12);
    const double c22 = fma(m00, m11, -m01 * m01);
    const double det = fma(m00, c00, fma(m01, c01, m02 * c02));
    o.pd = (m00 > 0.0) && (c22 > 0.0) && (det > 0.0);
    o.nid = -fast_rcp(det);
    o.y0 = (c00 * b0 + c01 * b1 + c02 * b2) * o.nid;
    o.y1 = (c01 * b0 + c11 * b1 + c12 * b2) * o.nid;
    o.y2 = (c02 * b0 + c12 * b1 + c22 * b2) * o.nid;
    o.p0 = -(c00 * o.y0 + c01 * o.y1 + c02 * o.y2) * o.nid;
    o.p1 = -(c01 * o.y0 + c11 * o.y1 + c12 * o.y2) * o.nid;
    o.p2 = -(c02 * o.y0 + c12 * o.y1 + c22 * o.y2) * o.nid;
    o.f = c + (b0 * o.y0 + b1 * o.y1 + b2 * o.y2) - lam;
    o.qq = 1.0 + (o.y0 * o.y0 + o.y1 * o.y1 + o.y2 * o.y2);          // -f'
    o.qp = o.y0 * o.p0 + o.y1 * o.p1 + o.y2 * o.p2;                   // -f''/2
    o.pp = o.p0 * o.p0 + o.p1 * o.p1 + o.p2 * o.p2;
    o.tr = m00 + m11 + m22;
    const double qq2 = o.qq * o.qq;
    const double den = fma(o.f, o.qp, qq2);
    o.step = o.f * o.qq * fast_rcp(den > 0.0 ? den : qq2);            // Halley, else Newton (f / qq)
}

__device__ __forceinline__ void smallest_eigvec(const double N[10], double q[3]) {
    const double c = N[9];
    double q0 = d_nan(), q1 = d_nan(), q2 = d_nan();
    bool weak = false;                      // M has a nearly free direction: the result gets a refinement step
    double lam, lo = 0.0, hi = kInf;
    bool done;
    EigPass e;
    // First pass, at lambda = 0 (the inhomogeneous least-squares point): never the last one, so it carries no
    // stopping test.  M itself not positive definite = rank-deficient system: give up at once (q stays NaN).
    eig_pass(N, 0.0, e);
    if (e.pd) {
        q0 = fma(e.step, e.p0, e.y0); q1 = fma(e.step, e.p1, e.y1); q2 = fma(e.step, e.p2, e.y2);
        weak = e.pp * c * c > 1e8;
    }
    lam = e.step;
    done = !e.pd || !(fabs(lam) < kInf);
    lam = done ? 0.0 : lam;
    const double tol_abs = 2e-15 * fabs(c);
#pragma unroll 1
    for (int it = 1; it < 64; ++it) {       // one more pass as a rule; a root next to the pole mu_1 needs the bisection below
        if (__all(done)) break;
        eig_pass(N, lam, e);
        // Error of the corrected q = q(lambda) + step dq/dlambda, two terms:
        //  * |error of lambda + step| <~ |f''/(2f')| step^2 (Newton's bound; Halley's is smaller) times |dq/dlambda|
        //  * the second-order term of q itself: |(M - lambda)^-1 dq/dlambda| step^2 <= |dq/dlambda| step^2 / (mu_1 - lambda),
        //    with 1 / (mu_1 - lambda) = (mu_2 - lambda)(mu_3 - lambda) / det <= (tr / 2)^2 / det.  It only bites when M
        //    has a weak direction (two cameras left), where it was worth up to 3e-7 m after a single pass.
        const double st2 = e.step * e.step;
        const double t = e.qp * st2;                                   // (t / qq)^2 pp <= 1e-21 qq
        const double s2 = 0.25 * e.tr * e.tr * e.nid * st2;
        const double lim = 1e-21 * e.qq;
        const bool conv = e.pd && (((t * t * e.pp <= lim * e.qq * e.qq) && (s2 * s2 * e.pp <= lim)) || (fabs(e.step) <= tol_abs));
        if (!done && e.pd) {
            q0 = fma(e.step, e.p0, e.y0); q1 = fma(e.step, e.p1, e.y1); q2 = fma(e.step, e.p2, e.y2);
            weak = e.pp * c * c > 1e8;      // rounding noise of lambda (~1e-13 |c| in the worst cases seen) times |dq/dlambda|
        }
        double lam_new = lam + e.step;
        lo = (e.pd && e.f >= 0.0) ? lam : lo;
        hi = e.pd ? hi : fmin(hi, lam);
        lam_new = e.pd ? lam_new : 0.5 * (lo + hi);
        lam_new = (lam_new >= hi) ? 0.5 * (lam + hi) : lam_new;
        // NaN system, or no positive-definite point at all: give up at once
        const bool bad = !(lam_new == lam_new) || (!e.pd && !(hi > lo));
        if (!done) lam = lam_new;
        done = done || conv || bad;
    }
    // With a nearly free direction in M (two cameras facing each other: depth along their common line) the
    // rounding noise of f(lambda) -- b.q cancels against c to ~1e-10 relative -- leaves lambda off by ~1e-7 and
    // q off by |dq/dlambda| times that, up to ~1e-7 m.  One step of iterative refinement of (M - lambda) q = -b
    // with the residual formed in the original data, then lambda and q corrected together, brings it back to
    // ~1e-11 (checked against the SVD of A on 5.2 M units with 4 ring cameras).  Rare: a whole-wave branch.
    weak = weak && (q0 == q0);
    if (__any(weak)) {
        const double3 x = refine_eigvec(N[0], N[1], N[2], N[3], N[4], N[5], N[6], N[7], N[8], N[9], lam, q0, q1, q2);
        const bool use = weak && (x.x == x.x) && (x.y == x.y) && (x.z == x.z);
        q0 = use ? x.x : q0; q1 = use ? x.y : q1; q2 = use ? x.z : q2;
    }
    q[0] = q0; q[1] = q1; q[2] = q2;
}

// euclidean_distance (common.py:378-403) of one 2D point pair: all-NaN difference -> inf,
// otherwise NaN components are skipped.
__device__ __forceinline__ double pair_distance(double dx, double dy) {
    const bool nx = !(dx == dx), ny = !(dy == dy);
    const double sx = nx ? 0.0 : dx * dx;
    const double sy = ny ? 0.0 : dy * dy;
    return (nx && ny) ? kInf : fast_sqrt(sx + sy);
}

// cv2.projectPoints with the ORIGINAL intrinsics and distortion (triangulation.py:473, quirk Q4).
__device__ __forceinline__ void project_distorted(cam_cptr cam, const double q[3], double &u, double &v) {
    const double X = fma(cam->R[0], q[0], fma(cam->R[1], q[1], fma(cam->R[2], q[2], cam->T[0])));
    const double Y = fma(cam->R[3], q[0], fma(cam->R[4], q[1], fma(cam->R[5], q[2], cam->T[1])));
    double Z = fma(cam->R[6], q[0], fma(cam->R[7], q[1], fma(cam->R[8], q[2], cam->T[2])));
    Z = (Z == 0.0) ? 1.0 : Z;
    const double rZ = fast_rcp(Z);
    const double x = X * rZ, y = Y * rZ;
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    const double cdist = 1 + cam->k[0] * r2 + cam->k[1] * r4 + cam->k[4] * r6;
    const double xd = x * cdist + cam->k[2] * a1 + cam->k[3] * a2;
    const double yd = y * cdist + cam->k[2] * a3 + cam->k[3] * a1;
    u = xd * cam->fx + cam->cx;
    v = yd * cam->fy + cam->cy;
}

// cv2.undistortPoints(float32 pts, K, dist, None, optim_K) (triangulation.py:810-813): 5 fixed-point
// iterations in double, result rounded to float32.  Contraction is off and the operation order is
// that of pose2sim_amd/cvmath.py so that the float32 rounding is bit-identical to the oracle's.
__device__ __noinline__ void undistort_point(cam_cptr cam, double &px, double &py) {
#pragma clang fp contract(off)
    const double u = (double)(float)px, v = (double)(float)py;
    const double x0 = (u - cam->cx) * cam->ifx;
    const double y0 = (v - cam->cy) * cam->ify;
    double x = x0, y = y0;
    const double k0 = cam->k[0], k1 = cam->k[1], k2 = cam->k[2], k3 = cam->k[3], k4 = cam->k[4];
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        const double icdist = 1.0 / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        if (icdist < 0) { x = x0; y = y0; break; }
        const double dx = ((2 * k2) * x) * y + k3 * (r2 + (2 * x) * x);
        const double dy = k2 * (r2 + (2 * y) * y) + ((2 * k3) * x) * y;
        x = (x0 - dx) * icdist;
        y = (y0 - dy) * icdist;
    }
    const double xx = cam->nk[0] * x + cam->nk[1] * y + cam->nk[2];
    const double yy = cam->nk[3] * x + cam->nk[4] * y + cam->nk[5];
    const double ww = 1.0 / (cam->nk[6] * x + cam->nk[7] * y + cam->nk[8]);
    px = (double)(float)(xx * ww);
    py = (double)(float)(yy * ww);
}

// Lexicographic unranking of the r-th k-subset of {0..n-1} (itertools.combinations order).
__device__ __forceinline__ uint32_t unrank_subset(uint32_t r, int n, int k, const uint32_t *__restrict__ binom) {
    uint32_t S = 0;
    int x = 0;
    for (int left = k; left > 0; --left) {
        // number of subsets that start with element x: C(n-1-x, left-1)
        uint32_t cnt = binom[(n - 1 - x) * 33 + (left - 1)];
        while (r >= cnt) {
            r -= cnt;
            ++x;
            cnt = binom[(n - 1 - x) * 33 + (left - 1)];
        }
        S |= 1u << x;
        ++x;
    }
    return S;
}

// Rank of the k-subset S of {0..n-1} in itertools.combinations order (the inverse of unrank_subset): the subsets
// before it either start lower at some position, sum over x = prev+1 .. c-1 of C(n-1-x, left-1) = C(n-prev-1, left) -
// C(n-c, left) by the hockey-stick identity.
__device__ __forceinline__ uint32_t rank_subset(uint32_t S, int n, int k, const uint32_t *__restrict__ binom) {
    uint32_t r = 0;
    int prev = -1, left = k;
    for (uint32_t b = S; b != 0u; b &= b - 1, --left) {
        const int c = __builtin_ctz(b);
        r += binom[(n - prev - 1) * 33 + left] - binom[(n - c) * 33 + left];
        prev = c;
    }
    return r;
}

__device__ __forceinline__ double wave_min_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}

// Index of the n-th (0-based) set bit of a 64-bit mask, or -1.
__device__ __forceinline__ int nth_set_bit(unsigned long long m, int n) {
    for (int i = 0; i < n; ++i) m &= m - 1;
    return m ? __builtin_ctzll(m) : -1;
}

__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }

// One unit's observations in the LDS tile: camera c at p[c * stride + {0,1,2}].
// The likelihood mask (triangulation.py:817-821) is applied while reading: a likelihood below the
// threshold turns x, y and the likelihood into NaN (a NaN likelihood compares false and stays).
template <typename T>
struct UnitObs {
    const T *p;
    int stride;
    double lik_thr;
    __device__ __forceinline__ void raw(int c, double &x, double &y, double &w) const {
        const T *q = p + c * stride;
        x = (double)q[0]; y = (double)q[1]; w = (double)q[2];
    }
    __device__ __forceinline__ void rawT(int c, T &x, T &y, T &w) const {
        const T *q = p + c * stride;
        x = q[0]; y = q[1]; w = q[2];
    }
    __device__ __forceinline__ void masked_xy(int c, double &x, double &y) const {
        double w;
        raw(c, x, y, w);
        const bool low = w < lik_thr;
        x = low ? d_nan() : x;
        y = low ? d_nan() : y;
    }
};

// One unit's observations held in registers (direct kernel, C <= CT): indices must be compile-time
// constants, which the fully unrolled camera loops provide.
template <typename T, int CT>
struct RegObs {
    T x[CT], y[CT], w[CT];
    double lik_thr;
    __device__ __forceinline__ void raw(int c, double &xo, double &yo, double &wo) const {
        xo = (double)x[c]; yo = (double)y[c]; wo = (double)w[c];
    }
    __device__ __forceinline__ void rawT(int c, T &xo, T &yo, T &wo) const { xo = x[c]; yo = y[c]; wo = w[c]; }
    __device__ __forceinline__ void masked_xy(int c, double &xo, double &yo) const {
        const bool low = (double)w[c] < lik_thr;
        xo = low ? d_nan() : (double)x[c];
        yo = low ? d_nan() : (double)y[c];
    }
};

// Camera loop: CT > 0 -> fully unrolled for up to CT cameras (constant indices, and the scheduler sees
// every load of the pass at once); CT == 0 -> run-time camera count.
template <int CT, typename F>
__device__ __forceinline__ void for_each_cam(int C, F &&f) {
    if constexpr (CT > 0) {
#pragma unroll
        for (int c = 0; c < CT; ++c)
            if (c < C) f(c);
    } else {
#pragma unroll 2
        for (int c = 0; c < C; ++c) f(c);
    }
}

// Coordinates that are numbers (not inf, not NaN; a pair whose magnitudes sum to inf counts as not finite either).
template <typename T>
__device__ __forceinline__ bool finite_xy(T x, T y) { return (__builtin_fabs(x) + __builtin_fabs(y)) < (T)__builtin_huge_val(); }
__device__ __forceinline__ bool finite_xy(float x, float y) { return (__builtin_fabsf(x) + __builtin_fabsf(y)) < __builtin_huge_valf(); }

// Level-0 pass over all cameras: classify each camera (NaN / zero likelihood) and accumulate the
// normal matrix of the valid ones.  Branch-free: an invalid camera enters with weight 0.
// A camera whose coordinates are not finite although its likelihood passes is taken as a missing detection (the NaN
// class): the reference keeps it among the valid cameras, finds every subset that contains it at error inf and arrives at
// the same point, error, exclusion count and excluded-camera set one search level later.
template <typename T, int CT, typename OBS>
__device__ __forceinline__ void classify_and_accumulate(cam_cptr cams, int C, const OBS &o, double N[10],
                                                         uint32_t &nanmask, uint32_t &zeromask) {
    for_each_cam<CT>(C, [&](int c) {
        T x, y, w;
        o.rawT(c, x, y, w);
        const bool isn = !(w == w) || ((double)w < o.lik_thr) || !finite_xy(x, y);
        const bool isz = (w == (T)0) && !isn;
        nanmask |= isn ? (1u << c) : 0u;
        zeromask |= isz ? (1u << c) : 0u;
        const bool ok = !(isn || isz);
        accum_camera<1>(N, cams[c].P, (double)(ok ? x : (T)0), (double)(ok ? y : (T)0), (double)(ok ? w : (T)0));
    });
}

// 1/sqrt(t) for 0 < t < inf: v_rsq_f64 seed (5.2e-8 relative on gfx950) + one Newton step = 4e-15.
__device__ __forceinline__ double fast_rsqrt(double t) {
    const double r = __builtin_amdgcn_rsq(t);
    const double e = fma(-t * r, r, 1.0);
    return fma(0.5 * r, e, r);
}

// Reprojection distance of one camera (common.py:357-403).  With a = P0.Q, b = P1.Q, z = P2.Q:
// |(a/z - x, b/z - y)| = s / sqrt(s z^2), s = (a - x z)^2 + (b - y z)^2: one reciprocal square
// root instead of a division and a square root.  `regular` is false for degenerate or NaN operands
// (s z^2 not in (0, inf)); those are redone by camera_distance_exact.
template <bool UNDISTORT>
__device__ __forceinline__ double camera_distance(cam_cptr cam, const double q[3], double x, double y, bool &regular) {
    if (UNDISTORT) {
        double u, v;
        project_distorted(cam, q, u, v);
        regular = true;
        return pair_distance(u - x, v - y);
    }
    // an fp64 FMA reads one SGPR: with the constant term as the addend of the innermost FMA the compiler first moves it
    // to a VGPR pair (2 v_mov per row); multiplying first and adding the constant last is one instruction fewer per row
    const double a = fma(cam->P[0], q[0], fma(cam->P[1], q[1], cam->P[2] * q[2])) + cam->P[3];
    const double b = fma(cam->P[4], q[0], fma(cam->P[5], q[1], cam->P[6] * q[2])) + cam->P[7];
    const double z = fma(cam->P[8], q[0], fma(cam->P[9], q[1], cam->P[10] * q[2])) + cam->P[11];
    const double dxz = fma(-x, z, a), dyz = fma(-y, z, b);
    const double s = fma(dxz, dxz, dyz * dyz);
    const double t = s * z * z;
    regular = (t > 0.0) && (t < kInf);
    return s * fast_rsqrt(t);
}

// The literal formula with the reference's NaN rules (all-NaN difference -> inf, nansum).
// (q by value: a pointer argument of a non-inlined function would force q into scratch memory)
__device__ __noinline__ double camera_distance_exact(cam_cptr cam, double q0, double q1, double q2, double x, double y) {
    const double a = fma(cam->P[0], q0, fma(cam->P[1], q1, fma(cam->P[2], q2, cam->P[3])));
    const double b = fma(cam->P[4], q0, fma(cam->P[5], q1, fma(cam->P[6], q2, cam->P[7])));
    const double z = fma(cam->P[8], q0, fma(cam->P[9], q1, fma(cam->P[10], q2, cam->P[11])));
    const double rz = fast_rcp(z);
    return pair_distance(a * rz - x, b * rz - y);
}

// Mean reprojection error over the cameras of `kept` (triangulation.py:472-489).
template <typename T, bool UNDISTORT, int CT = 0, typename OBS>
__device__ __forceinline__ double mean_error(cam_cptr cams, int C, const OBS &o, uint32_t kept, const double q[3]) {
    double sum = 0.0;
    bool irregular = false;
    for_each_cam<CT>(C, [&](int c) {
        double x, y, w;
        o.raw(c, x, y, w);
        const bool k = (kept >> c) & 1u;
        bool reg;
        const double d = camera_distance<UNDISTORT>(cams + c, q, x, y, reg);
        irregular = irregular || (k && !reg);
        sum += k ? d : 0.0;
    });
    if (__any(irregular)) {                        // rare: some wanted camera is degenerate / NaN
        double sum2 = 0.0;
        for_each_cam<CT>(C, [&](int c) {
            double x, y, w;
            o.raw(c, x, y, w);
            const double d = camera_distance_exact(cams + c, q[0], q[1], q[2], x, y);
            sum2 += ((kept >> c) & 1u) ? d : 0.0;
        });
        sum = irregular ? sum2 : sum;
    }
    return sum * fast_rcp((double)__popc(kept));   // no camera kept -> NaN, as np.mean([])
}

// L/R-swap candidate (triangulation.py:509-561, quirk Q3): the first M kept cameras carry the
// mirrored keypoint's (x, y), still weighted by the unit's own likelihoods; the error is the mean
// over those first M cameras only.  In two parts, so that a caller may look at a few cameras first.
template <typename T, int CT = 0, typename OBS>
__device__ __forceinline__ void swap_solve(cam_cptr cams, int C, const OBS &o, const OBS &osw, uint32_t kept, int M, double qs[3]) {
    double Nw[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) Nw[i] = 0.0;
    int taken = 0;
    for_each_cam<CT>(C, [&](int c) {
        double x, y, w, xs, ys;
        o.raw(c, x, y, w);
        osw.masked_xy(c, xs, ys);
        const bool k = (kept >> c) & 1u;
        const bool sw = taken < M;
        accum_camera<1>(Nw, cams[c].P, k ? (sw ? xs : x) : 0.0, k ? (sw ? ys : y) : 0.0, k ? w : 0.0);
        taken += k ? 1 : 0;
    });
    smallest_eigvec(Nw, qs);
}

// The same solve for the many candidates of one unit's level: the normal matrix of ALL valid cameras with the mirrored
// coordinates is built once (swap_base; cameras whose mirrored point is masked are left out and remembered), and a
// candidate subtracts the cameras it removes -- and, where the subset "removes" cameras that were out already (quirk Q1
// padding), the last kept cameras, which then keep their own coordinates (only the first M kept ones are mirrored), and
// adds those back with their own coordinates.  A mirrored camera with a masked point makes the whole candidate NaN, as
// its row does in the accumulation above.  P comes from LDS (per-lane camera indices).
template <typename T, typename OBS>
__device__ __forceinline__ void swap_base(cam_cptr cams, int C, const OBS &o, const OBS &osw, uint32_t valid, double Nsw[10],
                                          uint32_t &nan_sw) {
#pragma unroll
    for (int i = 0; i < 10; ++i) Nsw[i] = 0.0;
    nan_sw = 0;
    for_each_cam<0>(C, [&](int c) {
        double x, y, w, xs, ys;
        o.raw(c, x, y, w);
        osw.masked_xy(c, xs, ys);
        const bool v = (valid >> c) & 1u;
        const bool bad = v && (!(xs == xs) || !(ys == ys));
        nan_sw |= bad ? (1u << c) : 0u;
        const bool use = v && !bad;
        accum_camera<1>(Nsw, cams[c].P, use ? xs : 0.0, use ? ys : 0.0, use ? w : 0.0);
    });
}

template <typename T, typename OBS>
__device__ __forceinline__ void swap_solve_from_base(const double Nsw[10], uint32_t nan_sw, const double *sP, const OBS &o,
                                                     const OBS &osw, uint32_t valid, uint32_t kept, int M, bool go, double qs[3]) {
    double Ns[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) Ns[i] = Nsw[i];
    uint32_t tail = 0;                                          // kept cameras beyond the first M: their own coordinates
    {
        uint32_t k = kept;
        for (int extra = go ? __popc(kept) - M : 0; __any(extra > 0); --extra) {
            const bool on = extra > 0 && k != 0u;
            const int c = on ? 31 - __builtin_clz(k) : 0;
            tail |= on ? (1u << c) : 0u;
            k &= on ? ~(1u << c) : ~0u;
        }
    }
    for (uint32_t rr = go ? (((valid & ~kept) | tail) & ~nan_sw) : 0u; __any(rr != 0u); rr &= rr - 1) {
        const bool on = rr != 0u;
        const int c = on ? __builtin_ctz(rr) : 0;
        double x, y, w, xs, ys;
        o.raw(c, x, y, w);
        osw.masked_xy(c, xs, ys);
        accum_camera<-1>(Ns, sP + c * 12, on ? xs : 0.0, on ? ys : 0.0, on ? w : 0.0);
    }
    for (uint32_t rr = go ? tail : 0u; __any(rr != 0u); rr &= rr - 1) {
        const bool on = rr != 0u;
        const int c = on ? __builtin_ctz(rr) : 0;
        double x, y, w;
        o.raw(c, x, y, w);
        accum_camera<1>(Ns, sP + c * 12, on ? x : 0.0, on ? y : 0.0, on ? w : 0.0);
    }
    smallest_eigvec(Ns, qs);
    if ((kept & ~tail & nan_sw) != 0u) { qs[0] = d_nan(); qs[1] = d_nan(); qs[2] = d_nan(); }
}

template <typename T, bool UNDISTORT, int CT = 0, typename OBS>
__device__ __forceinline__ double swap_error(cam_cptr cams, int C, const OBS &osw, uint32_t kept, int M, const double qs[3]) {
    double sum = 0.0;
    bool irregular = false;
    int taken = 0;
    for_each_cam<CT>(C, [&](int c) {
        double xs, ys;
        osw.masked_xy(c, xs, ys);
        const bool k = ((kept >> c) & 1u) && taken < M;
        bool reg;
        const double d = camera_distance<UNDISTORT>(cams + c, qs, xs, ys, reg);
        irregular = irregular || (k && !reg);
        sum += k ? d : 0.0;
        taken += ((kept >> c) & 1u) ? 1 : 0;
    });
    if (__any(irregular)) {
        double sum2 = 0.0;
        taken = 0;
        for_each_cam<CT>(C, [&](int c) {
            double xs, ys;
            osw.masked_xy(c, xs, ys);
            const bool k = ((kept >> c) & 1u) && taken < M;
            const double d = camera_distance_exact(cams + c, qs[0], qs[1], qs[2], xs, ys);
            sum2 += k ? d : 0.0;
            taken += ((kept >> c) & 1u) ? 1 : 0;
        });
        sum = irregular ? sum2 : sum;
    }
    return sum * fast_rcp((double)M);
}

template <typename T, bool UNDISTORT, int CT = 0, typename OBS>
__device__ __forceinline__ double swap_candidate(cam_cptr cams, int C, const OBS &o, const OBS &osw, uint32_t kept,
                                                 int M, double qs[3]) {
    swap_solve<T, CT>(cams, C, o, osw, kept, M, qs);
    return swap_error<T, UNDISTORT, CT>(cams, C, osw, kept, M, qs);
}

// A swap candidate whose error is certainly above `bmean` is worth nothing (bmean: what can still matter; the sum of the
// first-M kept cameras' distances only grows): the distances of the cameras in `order` (most suspicious first, wave
// uniform) are added until every lane of the wave is out -- then inf comes back -- or all are through, in which case
// the candidate is evaluated as swap_error does it.  NaN terms add nothing here and bound nothing.
template <typename T, bool UNDISTORT, typename OBS>
__device__ __forceinline__ double swap_error_pruned(cam_cptr cams, int C, const OBS &osw, uint32_t kept, int M, const double qs[3],
                                                    const uint8_t *order, bool go, double bmean) {
    const double bnd = bmean * (double)M * (1.0 + 1e-9);
    double psum = 0.0;
    for (int idx = 0; idx < C; ++idx) {
        const int c = __builtin_amdgcn_readfirstlane((int)order[idx]);
        double xs, ys;
        osw.masked_xy(c, xs, ys);
        bool reg;
        const double d = camera_distance<UNDISTORT>(cams + c, qs, xs, ys, reg);
        const bool k = ((kept >> c) & 1u) && (__popc(kept & ((1u << c) - 1u)) < M);     // among the first M kept cameras
        psum += (k && reg && d == d) ? d : 0.0;
        if ((idx & 3) == 3 && __all(!go || psum > bnd)) break;
    }
    const bool alive = go && !(psum > bnd);
    if (!__any(alive)) return kInf;
    const double es = swap_error<T, UNDISTORT, 0>(cams, C, osw, kept, M, qs);
    return alive ? es : kInf;
}

}  // namespace

#endif
