// p2s_tri.hip -- fused robust multi-view triangulation for gfx950 (MI355X, CDNA4).
//
// One kernel does, per (block = frame x person, keypoint) unit, everything the reference does in
// triangulate_all's frame/person/keypoint loops (triangulation.py:796-845) and in
// triangulation_from_best_cameras (triangulation.py:363-604):
//
//   stage    a tile of FB consecutive blocks ([FB][C][K][3], contiguous in HBM) is copied into LDS
//            with 16-byte-per-lane coalesced loads; the C projection matrices sit beside it
//   prepare  each lane undistorts (triangulation.py:808-813) and likelihood-masks (:817-821) the
//            observations of its unit in place in LDS
//   level 0  one lane per unit: weighted-DLT normal matrix (common.py:327-354 restated as the
//            smallest eigenpair of A^T A, fp64), reprojection error (common.py:357-403)
//   search   units whose error exceeds the threshold are handed to groups of G lanes of the same
//            wavefront; lane j of a group evaluates camera subset #(round*G + j) of the level
//            (lexicographic itertools.combinations order, triangulation.py:411), the group's
//            argmin (first index on ties, :502) is taken with wave shuffles; levels proceed in
//            lock step across the wave (ballot of units still above threshold)
//
// Everything is data-parallel fp64 VALU work on an HBM-streamed tensor: no MFMA (4x4 systems).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "p2s_internal.h"

namespace {

constexpr double kInf = __builtin_huge_val();

__device__ __forceinline__ double d_nan() { return __builtin_nan(""); }

// --------------------------------------------------------------------------------------------
// Normal-matrix contribution of one camera: rows (P0 - x P2) w and (P1 - y P2) w.
// N is the upper triangle of the 4x4 A^T A: [00 01 02 03 11 12 13 22 23 33].
template <bool SUBTRACT>
__device__ __forceinline__ void accum_camera(double N[10], const double *__restrict__ P, double x, double y,
                                             double w) {
    const double a0 = (P[0] - x * P[8]) * w, a1 = (P[1] - x * P[9]) * w, a2 = (P[2] - x * P[10]) * w,
                 a3 = (P[3] - x * P[11]) * w;
    const double b0 = (P[4] - y * P[8]) * w, b1 = (P[5] - y * P[9]) * w, b2 = (P[6] - y * P[10]) * w,
                 b3 = (P[7] - y * P[11]) * w;
    const double s = SUBTRACT ? -1.0 : 1.0;
    N[0] += s * (a0 * a0 + b0 * b0);
    N[1] += s * (a0 * a1 + b0 * b1);
    N[2] += s * (a0 * a2 + b0 * b2);
    N[3] += s * (a0 * a3 + b0 * b3);
    N[4] += s * (a1 * a1 + b1 * b1);
    N[5] += s * (a1 * a2 + b1 * b2);
    N[6] += s * (a1 * a3 + b1 * b3);
    N[7] += s * (a2 * a2 + b2 * b2);
    N[8] += s * (a2 * a3 + b2 * b3);
    N[9] += s * (a3 * a3 + b3 * b3);
}

// Smallest eigenvector of the 4x4 SPD matrix N, dehomogenised: v = (q, 1), N v = lambda v.
// With N = [[M, b], [b^T, c]]: (M - lambda I) q = -b and lambda = c + b.q.  Newton on that secular
// equation (== Rayleigh-quotient update) from lambda = 0, i.e. from the inhomogeneous least-squares
// point; monotone and quadratically convergent below the smallest eigenvalue of M.  The 3x3
// systems are solved by LDL^T.  Returns q = V[0:3,3]/V[3,3] of the reference's SVD
// (common.py:348-350) to ~1e-12 relative.
__device__ __forceinline__ void smallest_eigvec(const double N[10], double q[3]) {
    const double b0 = N[3], b1 = N[6], b2 = N[8], c = N[9];
    const double tol_abs = 2e-15 * fabs(c);
    double lam = 0.0;
    double q0 = 0, q1 = 0, q2 = 0;
    bool done = false;
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        const double d0 = N[0] - lam;
        const double i0 = 1.0 / d0;
        const double l10 = N[1] * i0, l20 = N[2] * i0;
        const double d1 = (N[4] - lam) - l10 * N[1];
        const double i1 = 1.0 / d1;
        const double t21 = N[5] - l20 * N[1];
        const double l21 = t21 * i1;
        const double d2 = (N[7] - lam) - l20 * N[2] - l21 * t21;
        const double i2 = 1.0 / d2;
        const double z0 = -b0;
        const double z1 = -b1 - l10 * z0;
        const double z2 = -b2 - l20 * z0 - l21 * z1;
        const double y2 = z2 * i2;
        const double y1 = z1 * i1 - l21 * y2;
        const double y0 = z0 * i0 - l10 * y1 - l20 * y2;
        if (!done) { q0 = y0; q1 = y1; q2 = y2; }
        const double g = c + (b0 * y0 + b1 * y1 + b2 * y2) - lam;
        const double qq = 1.0 + (y0 * y0 + y1 * y1 + y2 * y2);
        const double lam_new = lam + g / qq;
        const bool conv = fabs(lam_new - lam) <= 1e-9 * fabs(lam_new) + tol_abs;
        if (!done) lam = lam_new;
        // a NaN system never converges; give up on it at once
        done = done || conv || !(lam_new == lam_new);
        if (__all(done)) break;
    }
    q[0] = q0; q[1] = q1; q[2] = q2;
}

// euclidean_distance (common.py:378-403) of one 2D point pair: all-NaN difference -> inf,
// otherwise NaN components are skipped.
__device__ __forceinline__ double pair_distance(double dx, double dy) {
    const bool nx = !(dx == dx), ny = !(dy == dy);
    const double sx = nx ? 0.0 : dx * dx;
    const double sy = ny ? 0.0 : dy * dy;
    return (nx && ny) ? kInf : sqrt(sx + sy);
}

// reprojection (common.py:357-375)
__device__ __forceinline__ void project_pinhole(const double *__restrict__ P, const double q[3], double &u,
                                                double &v) {
    const double a = P[0] * q[0] + P[1] * q[1] + P[2] * q[2] + P[3];
    const double b = P[4] * q[0] + P[5] * q[1] + P[6] * q[2] + P[7];
    const double z = P[8] * q[0] + P[9] * q[1] + P[10] * q[2] + P[11];
    u = a / z;
    v = b / z;
}

// cv2.projectPoints with the ORIGINAL intrinsics and distortion (triangulation.py:473, quirk Q4).
__device__ __forceinline__ void project_distorted(const P2sCam &cam, const double q[3], double &u, double &v) {
    const double X = cam.R[0] * q[0] + cam.R[1] * q[1] + cam.R[2] * q[2] + cam.T[0];
    const double Y = cam.R[3] * q[0] + cam.R[4] * q[1] + cam.R[5] * q[2] + cam.T[1];
    double Z = cam.R[6] * q[0] + cam.R[7] * q[1] + cam.R[8] * q[2] + cam.T[2];
    Z = (Z == 0.0) ? 1.0 : Z;
    const double x = X / Z, y = Y / Z;
    const double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
    const double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
    const double cdist = 1 + cam.k[0] * r2 + cam.k[1] * r4 + cam.k[4] * r6;
    const double xd = x * cdist + cam.k[2] * a1 + cam.k[3] * a2;
    const double yd = y * cdist + cam.k[2] * a3 + cam.k[3] * a1;
    u = xd * cam.fx + cam.cx;
    v = yd * cam.fy + cam.cy;
}

// cv2.undistortPoints(float32 pts, K, dist, None, optim_K) (triangulation.py:810-813): 5 fixed-point
// iterations in double, result rounded to float32.  Contraction is off and the operation order is
// that of pose2sim_amd/cvmath.py so that the float32 rounding is bit-identical to the oracle's.
__device__ __noinline__ void undistort_point(const P2sCam &cam, double &px, double &py) {
#pragma clang fp contract(off)
    const double u = (double)(float)px, v = (double)(float)py;
    const double x0 = (u - cam.cx) * cam.ifx;
    const double y0 = (v - cam.cy) * cam.ify;
    double x = x0, y = y0;
    const double k0 = cam.k[0], k1 = cam.k[1], k2 = cam.k[2], k3 = cam.k[3], k4 = cam.k[4];
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        const double icdist = 1.0 / (1 + ((k4 * r2 + k1) * r2 + k0) * r2);
        if (icdist < 0) { x = x0; y = y0; break; }
        const double dx = ((2 * k2) * x) * y + k3 * (r2 + (2 * x) * x);
        const double dy = k2 * (r2 + (2 * y) * y) + ((2 * k3) * x) * y;
        x = (x0 - dx) * icdist;
        y = (y0 - dy) * icdist;
    }
    const double xx = cam.nk[0] * x + cam.nk[1] * y + cam.nk[2];
    const double yy = cam.nk[3] * x + cam.nk[4] * y + cam.nk[5];
    const double ww = 1.0 / (cam.nk[6] * x + cam.nk[7] * y + cam.nk[8]);
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

// Index of the n-th (0-based) set bit of a wave-uniform 64-bit mask, or -1.
__device__ __forceinline__ int nth_set_bit(unsigned long long m, int n) {
    for (int i = 0; i < n; ++i) m &= m - 1;
    return m ? __builtin_ctzll(m) : -1;
}

__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }

struct Cand {   // one evaluated camera configuration
    double err;
    double q[3];
};

// Mean reprojection error of point q over the cameras in `kept` (bit mask), at most `limit`
// of them in camera order (limit = all for the plain candidate, M for the swap candidate).
template <typename T, bool UNDISTORT>
__device__ __forceinline__ double mean_reproj_error(const P2sTriArgs &a, const double *__restrict__ sP,
                                                    const T *__restrict__ obs, int strideC, uint32_t kept,
                                                    int limit, const double q[3], const T *__restrict__ obs_sw,
                                                    int n_swapped) {
    double sum = 0.0;
    int taken = 0;
    for (int c = 0; c < a.C; ++c) {
        if (!((kept >> c) & 1u) || taken >= limit) continue;
        double x = (double)obs[c * strideC + 0], y = (double)obs[c * strideC + 1];
        if (taken < n_swapped) { x = (double)obs_sw[c * strideC + 0]; y = (double)obs_sw[c * strideC + 1]; }
        double u, v;
        if (UNDISTORT) project_distorted(a.cams[c], q, u, v);
        else project_pinhole(sP + c * 12, q, u, v);
        sum += pair_distance(u - x, v - y);
        ++taken;
    }
    return sum / (double)taken;   // taken == 0 -> NaN, as np.mean of an empty list
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// LDS layout: [tile: FB*C*K*3 of T][P: C*12 doubles][binom: 33*33 u32]
template <typename T, bool UNDISTORT, bool LRSWAP>
__global__ void p2s_tri_kernel(const P2sTriArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int C = a.C, K = a.K, FB = a.FB;
    const int blk_elems = C * K * 3;
    T *tile = reinterpret_cast<T *>(smem);
    double *sP = reinterpret_cast<double *>(smem + a.lds_P_off);
    uint32_t *sBinom = reinterpret_cast<uint32_t *>(smem + a.lds_binom_off);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int64_t tile0 = (int64_t)blockIdx.x * FB;               // first block of this tile
    const int nb = (int)min((int64_t)FB, a.n_blocks - tile0);      // blocks in this tile
    const int n_units = nb * K;

    // ---- stage ---------------------------------------------------------------------------
    {
        const T *src = reinterpret_cast<const T *>(a.xyl) + tile0 * blk_elems;
        const int n_elems = nb * blk_elems;
        constexpr int VEC = 16 / sizeof(T);
        const int n_vec = n_elems / VEC;                           // tile base is 16-B aligned (host)
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        const v4u *src4 = reinterpret_cast<const v4u *>(src);
        v4u *dst4 = reinterpret_cast<v4u *>(tile);
        for (int i = tid; i < n_vec; i += blockDim.x) dst4[i] = __builtin_nontemporal_load(src4 + i);
        for (int i = n_vec * VEC + tid; i < n_elems; i += blockDim.x) tile[i] = src[i];
        for (int i = tid; i < C * 12; i += blockDim.x) sP[i] = a.cams[i / 12].P[i % 12];
        for (int i = tid; i < 33 * 33; i += blockDim.x) sBinom[i] = a.binom[i];
    }
    __syncthreads();

    // ---- prepare: undistort + likelihood mask, in place ------------------------------------
    for (int u = tid; u < n_units; u += blockDim.x) {
        const int b = u / K, k = u - b * K;
        T *o = tile + (size_t)b * blk_elems + k * 3;
        for (int c = 0; c < C; ++c) {
            T *p = o + c * K * 3;
            double x = (double)p[0], y = (double)p[1];
            const double l = (double)p[2];
            if (UNDISTORT) {
                undistort_point(a.cams[c], x, y);
                p[0] = (T)x; p[1] = (T)y;
            }
            if (l < a.lik_thr) {           // NaN likelihood compares false and stays NaN
                p[0] = (T)d_nan(); p[1] = (T)d_nan(); p[2] = (T)d_nan();
            }
        }
    }
    __syncthreads();

    const double thr = a.thr;
    const int G = a.G;                      // lanes per search group (power of two >= C, <= 64)
    const int groups = 64 / G;
    const int grp = lane / G, lig = lane - grp * G;

    for (int base = 0; base < n_units; base += blockDim.x) {
        const int u = base + tid;
        const bool active = u < n_units;
        const int b = active ? u / K : 0, k = active ? u - b * K : 0;
        const T *obs = tile + (size_t)b * blk_elems + k * 3;            // camera stride K*3
        const int strideC = K * 3;
        const T *obs_sw = obs;
        if (LRSWAP) obs_sw = tile + (size_t)b * blk_elems + a.swap_idx[k] * 3;

        // ---- level 0 ---------------------------------------------------------------------
        double N[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) N[i] = 0.0;
        uint32_t nanmask = 0, zeromask = 0;
        if (active) {
            for (int c = 0; c < C; ++c) {
                const double x = (double)obs[c * strideC + 0], y = (double)obs[c * strideC + 1],
                             w = (double)obs[c * strideC + 2];
                if (!(w == w)) { nanmask |= 1u << c; continue; }
                if (w == 0.0) { zeromask |= 1u << c; continue; }
                accum_camera<false>(N, sP + c * 12, x, y, w);
            }
        }
        const uint32_t allmask = (C == 32) ? 0xffffffffu : ((1u << C) - 1u);
        const uint32_t dmask = nanmask | zeromask;                 // cameras already out (NaN or zero likelihood)
        const uint32_t valid = allmask & ~dmask;
        const int V = __popc(dmask);
        const int nvalid = C - V;
        const int Lmax = C - a.min_cams - V;                       // last level that runs (triangulation.py:408, 437-441)

        double err_min = kInf;
        double Qb[3] = {d_nan(), d_nan(), d_nan()};
        int n_excl = C;                                            // :595-596 when no level completes
        uint32_t mask = allmask;
        bool need = false;

        if (active && Lmax >= 0) {
            double q[3] = {d_nan(), d_nan(), d_nan()};
            if (nvalid >= 2) smallest_eigvec(N, q);                // common.py:347 (fewer than 4 rows -> NaN)
            double e = mean_reproj_error<T, UNDISTORT>(a, sP, obs, strideC, valid, C, q, obs, 0);
            err_min = e; Qb[0] = q[0]; Qb[1] = q[1]; Qb[2] = q[2];
            n_excl = V; mask = nanmask;
            if (LRSWAP && err_min > thr && nvalid > 2) {           // :509-579 at level 0: M = nvalid
                double Ns[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) Ns[i] = 0.0;
                for (int c = 0; c < C; ++c) {
                    if (!((valid >> c) & 1u)) continue;
                    accum_camera<false>(Ns, sP + c * 12, (double)obs_sw[c * strideC + 0],
                                        (double)obs_sw[c * strideC + 1], (double)obs[c * strideC + 2]);
                }
                double qs[3];
                smallest_eigvec(Ns, qs);
                const double es = mean_reproj_error<T, UNDISTORT>(a, sP, obs, strideC, valid, nvalid, qs, obs_sw, nvalid);
                if (es < err_min) { err_min = es; Qb[0] = qs[0]; Qb[1] = qs[1]; Qb[2] = qs[2]; }
            }
            need = (err_min > thr) && (Lmax >= 1);
        }

        // ---- subset search, levels in lock step across the wave -----------------------------
        unsigned long long pend_level = __ballot(need);
        for (int level = 1; pend_level != 0ull; ++level) {
            unsigned long long pending = pend_level;
            bool cont = false;                                      // owner lanes: continue to level+1
            const uint32_t nsub = sBinom[C * 33 + level];
            while (pending != 0ull) {
                const unsigned long long before = pending;
                for (int i = 0; i < groups && pending; ++i) pending &= pending - 1;
                const unsigned long long batch = before & ~pending;   // owners served in this pass
                const int owner = nth_set_bit(batch, grp);            // unit this group works on (lane id) or -1

                // gather the owner's state
                const int src = owner < 0 ? lane : owner;
                const uint32_t o_nan = __shfl(nanmask, src, 64), o_zero = __shfl(zeromask, src, 64);
                const int o_unit = __shfl(u, src, 64);
                double No[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) No[i] = shfl_d(N[i], src);
                const uint32_t o_d = o_nan | o_zero, o_valid = allmask & ~o_d;
                const int oV = __popc(o_d);
                const int ob = o_unit / K, ok = o_unit - ob * K;
                const T *oobs = tile + (size_t)ob * blk_elems + ok * 3;
                const T *oobs_sw = oobs;
                if (LRSWAP) oobs_sw = tile + (size_t)ob * blk_elems + a.swap_idx[ok] * 3;
                const int M = C - oV - level;                       // cameras left when `level` valid ones go (:437, 513)

                // best candidates seen by this lane (plain / swap), lowest rank first
                double be = kInf, bq0 = d_nan(), bq1 = d_nan(), bq2 = d_nan();
                uint32_t brank = 0xffffffffu, bS = 0;
                double se = kInf, sq0 = d_nan(), sq1 = d_nan(), sq2 = d_nan();
                uint32_t srank = 0xffffffffu, sS = 0;

                for (uint32_t r0 = 0; r0 < nsub; r0 += G) {
                    const uint32_t r = r0 + lig;
                    bool go = (owner >= 0) && (r < nsub);
                    uint32_t S = 0;
                    if (go) {
                        S = unrank_subset(r, C, level, sBinom);
                        // duplicates of one effective configuration (quirk Q1) carry identical numbers;
                        // only the lexicographically first one -- padding = lowest cameras of D -- can win
                        const uint32_t pad = S & o_d;
                        const int np = __popc(pad);
                        uint32_t low = 0, dd = o_d;
                        for (int i = 0; i < np; ++i) { low |= dd & (0u - dd); dd &= dd - 1; }
                        go = (pad == low);
                    }
                    if (!go) continue;
                    const uint32_t Rreal = S & o_valid;
                    const uint32_t kept = o_valid & ~Rreal;
                    const int nkept = __popc(kept);
                    double Ns[10];
#pragma unroll
                    for (int i = 0; i < 10; ++i) Ns[i] = No[i];
                    for (uint32_t rr = Rreal; rr; rr &= rr - 1) {
                        const int c = __builtin_ctz(rr);
                        accum_camera<true>(Ns, sP + c * 12, (double)oobs[c * strideC + 0],
                                           (double)oobs[c * strideC + 1], (double)oobs[c * strideC + 2]);
                    }
                    double q[3] = {d_nan(), d_nan(), d_nan()};
                    if (nkept >= 2) smallest_eigvec(Ns, q);
                    const double e = mean_reproj_error<T, UNDISTORT>(a, sP, oobs, strideC, kept, C, q, oobs, 0);
                    if (e < be || (brank == 0xffffffffu)) { be = e; bq0 = q[0]; bq1 = q[1]; bq2 = q[2]; brank = r; bS = S; }
                    if (LRSWAP && M > 2) {
                        // the first M kept cameras carry the mirrored keypoint (quirk Q3)
                        double Nw[10];
#pragma unroll
                        for (int i = 0; i < 10; ++i) Nw[i] = 0.0;
                        int taken = 0;
                        for (int c = 0; c < C; ++c) {
                            if (!((kept >> c) & 1u)) continue;
                            const T *p = (taken < M) ? oobs_sw : oobs;
                            accum_camera<false>(Nw, sP + c * 12, (double)p[c * strideC + 0],
                                                (double)p[c * strideC + 1], (double)oobs[c * strideC + 2]);
                            ++taken;
                        }
                        double qs[3];
                        smallest_eigvec(Nw, qs);
                        const double es = mean_reproj_error<T, UNDISTORT>(a, sP, oobs, strideC, kept, M, qs, oobs_sw, M);
                        if (es < se || (srank == 0xffffffffu)) { se = es; sq0 = qs[0]; sq1 = qs[1]; sq2 = qs[2]; srank = r; sS = S; }
                    }
                }

                // group argmin, first (lowest-rank) index on ties (np.nanargmin, :502; np.argmin, :568)
                for (int off = G >> 1; off > 0; off >>= 1) {
                    const double oe = shfl_d(be, lane ^ off);
                    const uint32_t orank = __shfl(brank, lane ^ off, 64);
                    const bool take = (orank != 0xffffffffu) && (brank == 0xffffffffu || oe < be || (oe == be && orank < brank));
                    const double t0 = shfl_d(bq0, lane ^ off), t1 = shfl_d(bq1, lane ^ off), t2 = shfl_d(bq2, lane ^ off);
                    const uint32_t tS = __shfl(bS, lane ^ off, 64);
                    if (take) { be = oe; brank = orank; bq0 = t0; bq1 = t1; bq2 = t2; bS = tS; }
                    if (LRSWAP) {
                        const double xe = shfl_d(se, lane ^ off);
                        const uint32_t xrank = __shfl(srank, lane ^ off, 64);
                        const bool tk = (xrank != 0xffffffffu) && (srank == 0xffffffffu || xe < se || (xe == se && xrank < srank));
                        const double s0 = shfl_d(sq0, lane ^ off), s1 = shfl_d(sq1, lane ^ off), s2 = shfl_d(sq2, lane ^ off);
                        const uint32_t xS = __shfl(sS, lane ^ off, 64);
                        if (tk) { se = xe; srank = xrank; sq0 = s0; sq1 = s1; sq2 = s2; sS = xS; }
                    }
                }
                // level result, identical in every lane of the group
                double l_err = be, l_q0 = bq0, l_q1 = bq1, l_q2 = bq2;
                uint32_t l_mask = o_nan | bS;
                const int l_nexcl = oV + __popc(bS & o_valid);      // :436 counts NaN or zero
                if (LRSWAP && l_err > thr && M > 2 && se < l_err) { // :576-579, nb_cams_excluded NOT updated
                    l_err = se; l_q0 = sq0; l_q1 = sq1; l_q2 = sq2; l_mask = o_nan | sS;
                }

                // hand the result to the owner lane: it reads it from the first lane of its group
                const bool is_owner_now = (batch >> lane) & 1ull;
                const int from = __popcll(batch & ((1ull << lane) - 1ull)) * G;
                const double r_err = shfl_d(l_err, from), r_q0 = shfl_d(l_q0, from), r_q1 = shfl_d(l_q1, from),
                             r_q2 = shfl_d(l_q2, from);
                const uint32_t r_mask = __shfl(l_mask, from, 64);
                const int r_nexcl = __shfl(l_nexcl, from, 64);
                if (is_owner_now) {
                    err_min = r_err; Qb[0] = r_q0; Qb[1] = r_q1; Qb[2] = r_q2;
                    mask = r_mask; n_excl = r_nexcl;
                    cont = (err_min > thr) && (level + 1 <= Lmax);
                }
            }
            pend_level = __ballot(cont);
        }

        // ---- finalise (triangulation.py:588-604) ------------------------------------------
        if (active) {
            const int64_t gu = tile0 * K + u;
            const bool fail = !(err_min <= thr);
            double *Qo = a.Q + gu * 3;
            Qo[0] = fail ? d_nan() : Qb[0];
            Qo[1] = fail ? d_nan() : Qb[1];
            Qo[2] = fail ? d_nan() : Qb[2];
            a.err[gu] = fail ? __builtin_nanf("") : (float)err_min;
            a.n_excl[gu] = (uint8_t)n_excl;
            a.mask[gu] = mask;
        }
    }
}

// ---------------------------------------------------------------------------------------------
template <typename T>
static hipError_t launch_t(const P2sTriArgs &a, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
#define P2S_LAUNCH(U, L)                                                                                     \
    do {                                                                                                     \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&p2s_tri_kernel<T, U, L>),          \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
        if (e != hipSuccess) return e;                                                                       \
        hipLaunchKernelGGL((p2s_tri_kernel<T, U, L>), grid, block, lds, s, a);                                \
        return hipGetLastError();                                                                            \
    } while (0)
    if (a.undistort) {
        if (a.lr_swap) P2S_LAUNCH(true, true);
        else P2S_LAUNCH(true, false);
    } else {
        if (a.lr_swap) P2S_LAUNCH(false, true);
        else P2S_LAUNCH(false, false);
    }
#undef P2S_LAUNCH
}

hipError_t p2s_launch_tri(const P2sTriArgs &a, int dtype, int grid, int threads, size_t lds, hipStream_t s) {
    if (dtype == 0) return launch_t<float>(a, dim3(grid), dim3(threads), lds, s);
    return launch_t<double>(a, dim3(grid), dim3(threads), lds, s);
}
