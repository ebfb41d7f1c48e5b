// p2s_assoc.hip -- multi-person association of every frame on gfx950 (MI355X, CDNA4).
//
// One wavefront per frame does the per-frame body of associate_all (personAssociation.py:783-800):
//
//   rays      Pluecker coordinates of the camera->keypoint lines (compute_rays, :277-316), staged
//             in LDS one chunk of joints at a time
//   affinity  likelihood-weighted mean reciprocal product of every cross-view person pair
//             (compute_affinity, :347-408) with the circular constraint (:411-428, :794-795)
//   matchSVT  the ADMM loop of :450-509 entirely in LDS: each iteration thresholds the singular
//             values of an N x N matrix (SVT, :431-447).  The SVD is a one-sided (Hestenes) Jacobi
//             SVD in fp64 -- all n/2 column pairs of a round-robin step rotate in parallel, L lanes
//             per pair, dot products reduced by wave shuffles -- and U diag(max(s-t,0)) V^T is
//             rebuilt fused with the element-wise update, so the thresholded matrix is never stored
//   cut       entries below min_affinity are zeroed (:800) and the N x N result goes to HBM; the
//             order-sensitive proposal extraction (person_index_per_cam, :512-549) stays on the host
//
// N = detections of the frame over all cameras (<= P2S_MAX_PERSONS_TOTAL).  Per frame the work is
// ~10^7 fp64 flops on ~10 KB of input: compute/latency bound, LDS resident, no MFMA (the matrices
// are 32 x 32 and every step is data dependent).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include <math.h>

#include "p2s.h"
#include "p2s_internal.h"
#include "p2s_math.h"

namespace {

// one Newton step on the v_rcp_f64 / v_rsq_f64 seeds (5e-8 -> 4e-15, exp/seed_precision.hip)
__device__ __forceinline__ double rcp_1(double d) {
    const double r = __builtin_amdgcn_rcp(d);
    return fma(r, fma(-d, r, 1.0), r);
}
__device__ __forceinline__ double rsqrt_1(double t) {
    const double r = __builtin_amdgcn_rsq(t);
    return fma(0.5 * r, fma(-t * r, r, 1.0), r);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// NW waves work on one frame: NW = 1 needs only wave-local ordering, NW = 2 a workgroup barrier.
template <int NW>
__device__ __forceinline__ void lds_fence() {
    if constexpr (NW == 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}

// OR / sum over the whole workgroup (scratch: 2 * NW doubles of LDS, reused)
template <int NW>
__device__ __forceinline__ bool block_any(bool v, double *scratch, int tid) {
    const bool w = __any(v);
    if constexpr (NW == 1) return w;
    if ((tid & 63) == 0) scratch[tid >> 6] = w ? 1.0 : 0.0;
    __syncthreads();
    bool r = false;
#pragma unroll
    for (int i = 0; i < NW; ++i) r = r || (scratch[i] != 0.0);
    __syncthreads();
    return r;
}
template <int NW>
__device__ __forceinline__ double block_sum(double v, double *scratch, int tid) {
    v = wave_sum(v);
    if constexpr (NW == 1) return v;
    if ((tid & 63) == 0) scratch[tid >> 6] = v;
    __syncthreads();
    double r = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) r += scratch[i];
    __syncthreads();
    return r;
}

// Sum over the L consecutive lanes of a column pair with DPP moves (L = 2, 4, 8, 16): quad permutes for the first
// two stages, then row_half_mirror / row_mirror, which pair each lane with one that holds the other half's sum.
// ~10 cycles per stage against a ~120-cycle LDS round trip for ds_bpermute, and the Jacobi step is one long
// dependent chain.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    // every lane has a source lane under these controls: bound_ctrl spares the compiler the initialisation of "old"
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
template <int L>
__device__ __forceinline__ void group_sum3(double &a, double &b, double &c) {
    if constexpr (L >= 2) { a += dpp_move<0xB1>(a); b += dpp_move<0xB1>(b); c += dpp_move<0xB1>(c); }      // quad_perm [1,0,3,2]
    if constexpr (L >= 4) { a += dpp_move<0x4E>(a); b += dpp_move<0x4E>(b); c += dpp_move<0x4E>(c); }      // quad_perm [2,3,0,1]
    if constexpr (L >= 8) { a += dpp_move<0x141>(a); b += dpp_move<0x141>(b); c += dpp_move<0x141>(c); }   // row_half_mirror
    if constexpr (L >= 16) { a += dpp_move<0x140>(a); b += dpp_move<0x140>(b); c += dpp_move<0x140>(c); }  // row_mirror
    if constexpr (L >= 32) {
        a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64); c += __shfl_xor(c, 16, 64);
    }
}

// p2s_get_assoc_stats: one lane per frame adds to the context's sharded counters
__device__ __forceinline__ void assoc_count(const P2sAssocArgs &a, unsigned long long frames, unsigned long long passes,
                                            unsigned long long sweeps, unsigned long long flops) {
    unsigned long long *st = a.stats + (size_t)(blockIdx.x % P2S_STAT_SHARDS) * P2S_STAT_STRIDE;
    atomicAdd(st + 0, frames); atomicAdd(st + 1, passes); atomicAdd(st + 2, sweeps); atomicAdd(st + 3, flops);
}

// round-robin (circle method) pairing: step s of n-1, pair k of n/2 -> columns (p, q), n even
__device__ __forceinline__ void rr_pair(int n, int s, int k, int &p, int &q) {
    const int m = n - 1;
    if (k == 0) { p = m; q = s; return; }
    p = s + k; p -= (p >= m) ? m : 0;          // (s + k) mod m, operands < m
    q = s - k; q += (q < 0) ? m : 0;           // (s - k) mod m
}

// One-sided Jacobi SVD of the n x n matrix stored column-major in A (column j at A + j*n): on return
// the columns of A are sigma_j u_j and V (column-major) holds the right singular vectors (V must hold
// an orthogonal matrix on entry: the identity, or the previous iteration's V for a warm start, in
// which case A must already be (input matrix) . V).
// RPL = rows per lane (compile time): the RPL column elements a lane owns are read with one batch of
// LDS loads and held in registers, so a round-robin step costs two LDS round trips instead of 4*RPL.
template <int RPL, int L, int NW>
__device__ void jacobi_svd_t(double *A, double *V, int n, int ld, int lane, int &n_sweeps, double *scratch) {
    const int npairs = n >> 1;
    const int rpl = (n + L - 1) / L;                    // rows per lane (<= RPL)
    const int k = lane / L, sub = lane - k * L;
    const bool on = k < npairs;
    const int r0 = sub * rpl, cnt = on ? max(0, min(n, r0 + rpl) - r0) : 0;
    for (int sweep = 0; sweep < 40; ++sweep) {
        ++n_sweeps;
        bool rotated = false;
        for (int s = 0; s < n - 1; ++s) {
            int p = 0, q = 1;
            if (on) rr_pair(n, s, k, p, q);
            double *ap = A + p * ld + r0, *aq = A + q * ld + r0, *vp = V + p * ld + r0, *vq = V + q * ld + r0;
            // the V columns are fetched with the A columns: their LDS latency then hides under the dot products and
            // the rotation math instead of starting a second round trip once the rotation is known
            double x[RPL], y[RPL], vx[RPL], vy[RPL];
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const bool in = i < cnt;
                x[i] = in ? ap[i] : 0.0;
                y[i] = in ? aq[i] : 0.0;
                vx[i] = in ? vp[i] : 0.0;
                vy[i] = in ? vq[i] : 0.0;
            }
            double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
            for (int i = 0; i < RPL; ++i) { al = fma(x[i], x[i], al); be = fma(y[i], y[i], be); ga = fma(x[i], y[i], ga); }
            group_sum3<L>(al, be, ga);
            // rotate when |ga| > 1e-15 sqrt(al be)  (compared squared: no square root on the test)
            const bool rot = on && (ga * ga > 1e-30 * (al * be)) && (ga != 0.0);
            // a pair already orthogonal to 1e-7 is orthogonal to ~1e-14 after its rotation (quadratic convergence)
            rotated = rotated || (rot && (ga * ga > 1e-14 * (al * be)));
            if (__any(rot)) {
                // Rutishauser's stable formulas; reciprocal / square roots by Newton from the hardware
                // seeds: the rotation only has to be orthogonal to working precision
                // the angle only has to be good enough to annihilate the off-diagonal to ~1e-14 (one Newton step on the
                // hardware seeds: 4e-15); the rotation's orthogonality hangs on c below, which keeps its two steps
                const double zeta = (be - al) * rcp_1(2.0 * ga);
                const double zz = fma(zeta, zeta, 1.0);
                const double tt = rcp_1(fabs(zeta) + zz * rsqrt_1(zz));
                const double t = (zeta >= 0.0) ? tt : -tt;
                const double c = p2s_rsqrt(fma(t, t, 1.0)), sn = c * t;
#pragma unroll
                for (int i = 0; i < RPL; ++i) {
                    if (rot && i < cnt) {
                        ap[i] = c * x[i] - sn * y[i]; aq[i] = sn * x[i] + c * y[i];
                        vp[i] = c * vx[i] - sn * vy[i]; vq[i] = sn * vx[i] + c * vy[i];
                    }
                }
            }
            lds_fence<NW>();
        }
        if (!block_any<NW>(rotated, scratch, lane)) break;   // no pair was further than 1e-7 from orthogonal in this sweep: done
    }
}

// Streaming form for wide columns (n > 32: more than 8 rows per lane would not fit the register file).
template <int NW>
__device__ void jacobi_svd_wide(double *A, double *V, int n, int ld, int lane, int L, int &n_sweeps, double *scratch) {
    const int npairs = n >> 1;
    const int rpl = (n + L - 1) / L;
    const int k = lane / L, sub = lane - k * L;
    const bool on = k < npairs;
    const int r0 = sub * rpl, r1 = min(n, r0 + rpl);
    for (int sweep = 0; sweep < 40; ++sweep) {
        ++n_sweeps;
        bool rotated = false;
        for (int s = 0; s < n - 1; ++s) {
            int p = 0, q = 1;
            if (on) rr_pair(n, s, k, p, q);
            double *ap = A + p * ld, *aq = A + q * ld, *vp = V + p * ld, *vq = V + q * ld;
            double al = 0.0, be = 0.0, ga = 0.0;
            if (on) {
#pragma unroll 4
                for (int r = r0; r < r1; ++r) {
                    const double x = ap[r], y = aq[r];
                    al = fma(x, x, al); be = fma(y, y, be); ga = fma(x, y, ga);
                }
            }
            for (int off = L >> 1; off > 0; off >>= 1) {
                al += __shfl_xor(al, off, 64); be += __shfl_xor(be, off, 64); ga += __shfl_xor(ga, off, 64);
            }
            const bool rot = on && (fabs(ga) > 1e-15 * sqrt(al * be)) && (ga != 0.0);
            rotated = rotated || (rot && (fabs(ga) > 1e-7 * sqrt(al * be)));
            if (rot) {
                const double zeta = (be - al) / (2.0 * ga);
                const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
#pragma unroll 4
                for (int r = r0; r < r1; ++r) {
                    const double x = ap[r], y = aq[r];
                    ap[r] = c * x - sn * y; aq[r] = sn * x + c * y;
                    const double vx = vp[r], vy = vq[r];
                    vp[r] = c * vx - sn * vy; vq[r] = sn * vx + c * vy;
                }
            }
            lds_fence<NW>();
        }
        if (!block_any<NW>(rotated, scratch, lane)) break;
    }
}

template <int NW>
__device__ void jacobi_svd(double *A, double *V, int n, int ld, int lane, int &n_sweeps, double *scratch) {
    const int npairs = n >> 1;
    int L = 1;
    while ((L << 1) * npairs <= 64 * NW && L < 32) L <<= 1;   // lanes per column pair (power of two, within a wave)
    // one wave:  n <= 4 -> 32, n <= 8 -> 16, n <= 16 -> 8, n <= 32 -> 4, else 2
    // two waves: n <= 8 -> 32, n <= 16 -> 16, n <= 32 -> 8, n <= 64 -> 4
    const int rpl = (n + L - 1) / L;
    if (L == 32) jacobi_svd_t<1, 32, NW>(A, V, n, ld, lane, n_sweeps, scratch);
    else if (L == 16) jacobi_svd_t<1, 16, NW>(A, V, n, ld, lane, n_sweeps, scratch);
    else if (L == 8 && rpl <= 2) jacobi_svd_t<2, 8, NW>(A, V, n, ld, lane, n_sweeps, scratch);
    else if (L == 8 && rpl <= 4) jacobi_svd_t<4, 8, NW>(A, V, n, ld, lane, n_sweeps, scratch);
    else if (L == 4 && rpl <= 8) jacobi_svd_t<8, 4, NW>(A, V, n, ld, lane, n_sweeps, scratch);
    else jacobi_svd_wide<NW>(A, V, n, ld, lane, L, n_sweeps, scratch);
}

}  // namespace

// LDS (doubles): A, V, Y, X of n x (n+1) each, wts[n] ; ints: view[n].  The constant matrix W of matchSVT
// (read twice per element per iteration) lives in the frame's slab of the output buffer in HBM/L2 until
// the result overwrites it: one matrix less in LDS lets a fourth frame share the CU.
// the ray chunk aliases A, V, Y while the affinity is being accumulated
template <typename T, int NW>
__global__ void __launch_bounds__(64 * NW) p2s_assoc_kernel(const P2sAssocArgs a) {
    constexpr int NT = 64 * NW;                     // threads working on this frame
    extern __shared__ __align__(16) unsigned char smem[];
    const int n_max = a.Nmax;                       // even
    // every LDS matrix has an odd leading dimension (n + 1): with 32 doubles per column the 16 column pairs of a
    // Jacobi step (and the 32 rows read side by side in the products) all started in the same bank -- a 16-way
    // conflict on every access, which is where most of the step's ~3 400 cycles went
    const int mat = n_max * (n_max + 1);
    double *A = reinterpret_cast<double *>(smem);
    double *V = A + mat;
    double *Y = V + mat;
    double *X = Y + mat;
    double *wts = X + mat;
    double *scratch = wts + n_max;                  // 4 doubles for the workgroup reductions
    int *view = reinterpret_cast<int *>(scratch + 4);
    double *rays = A;                               // [person][joint in chunk][7]
    const int lane = threadIdx.x;                   // 0 .. NT-1
    const int64_t f = blockIdx.x;
    const int C = a.C, Kj = a.Kj;
    double *out = a.affinity + f * (int64_t)n_max * n_max;
    double *W = out;                                // stride n (n*n <= n_max*n_max)

    // ---- frame layout ---------------------------------------------------------------------
    int N = 0;
    for (int c = 0; c < C; ++c) {
        const int pc = a.n_persons[f * C + c];
        for (int i = lane; i < pc; i += NT)
            if (N + i < n_max) view[N + i] = c;
        N += pc;
    }
    N = min(N, n_max);
    const int n = max(2, (N + 1) & ~1);             // even working size (zero padding)
    const int ld = n + 1;                           // leading dimension of the LDS matrices
    for (int i = lane; i < n * ld; i += NT) X[i] = 0.0;
    for (int i = N + lane; i < n; i += NT) view[i] = -1 - i;   // padding rows: each its own "view"
    for (int i = lane; i < n_max * n_max; i += NT) out[i] = 0.0;
    lds_fence<NW>();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");   // W (global) is read back by other lanes of this wave
    if (N == 0) return;

    const T *kp = reinterpret_cast<const T *>(a.kpts) + a.offsets[f] * (int64_t)Kj * 3;
    const int n_pairs = N * (N - 1) / 2;
    const bool trace = P2S_DEBUG_MODE(a) == 7;
    uint64_t t_start = 0, t_aff = 0, t_prod = 0, t_svd = 0, t_upd = 0;
    int n_sweeps = 0, n_iter = 0, n_iter_all = 0;
    if (trace) t_start = __builtin_amdgcn_s_memtime();

    // ---- rays + affinity accumulation, a chunk of joints at a time ------------------------------
    int Kc = (3 * n_max * n_max) / (7 * N);
    Kc = max(1, min(Kc, Kj));
    for (int j0 = 0; j0 < Kj; j0 += Kc) {
        const int kc = min(Kc, Kj - j0);
        for (int it = lane; it < N * kc; it += NT) {          // compute_rays (:293-314)
            const int i = it / kc, j = it - i * kc;
            const P2sCam &cam = a.cams[view[i]];
            const T *o = kp + ((int64_t)i * Kj + j0 + j) * 3;
            const double x = (double)o[0], y = (double)o[1], lik = (double)o[2];
            const double v0 = cam.iK[0] * x + cam.iK[1] * y + cam.iK[2] - cam.T[0];
            const double v1 = cam.iK[3] * x + cam.iK[4] * y + cam.iK[5] - cam.T[1];
            const double v2 = cam.iK[6] * x + cam.iK[7] * y + cam.iK[8] - cam.T[2];
            const double l0 = cam.R[0] * v0 + cam.R[3] * v1 + cam.R[6] * v2 - cam.center[0];
            const double l1 = cam.R[1] * v0 + cam.R[4] * v1 + cam.R[7] * v2 - cam.center[1];
            const double l2 = cam.R[2] * v0 + cam.R[5] * v1 + cam.R[8] * v2 - cam.center[2];
            const double nrm = sqrt(l0 * l0 + l1 * l1 + l2 * l2);
            double d0 = l0 / nrm, d1 = l1 / nrm, d2 = l2 / nrm;
            double m0 = cam.center[1] * d2 - cam.center[2] * d1;
            double m1 = cam.center[2] * d0 - cam.center[0] * d2;
            double m2 = cam.center[0] * d1 - cam.center[1] * d0;
            double lk = lik;
            const bool anynan = !(d0 == d0) || !(d1 == d1) || !(d2 == d2) || !(m0 == m0) || !(m1 == m1) ||
                                !(m2 == m2) || !(lk == lk);
            if (anynan) { d0 = d1 = d2 = m0 = m1 = m2 = lk = 0.0; }
            double *r = rays + ((size_t)i * kc + j) * 7;
            r[0] = d0; r[1] = d1; r[2] = d2; r[3] = m0; r[4] = m1; r[5] = m2; r[6] = lk;
        }
        lds_fence<NW>();
        for (int pr = lane; pr < n_pairs; pr += NT) {         // compute_affinity (:383-394)
            // pair index -> (i, l), i < l
            int i = (int)((1.0 + sqrt(1.0 + 8.0 * (double)pr)) * 0.5);
            while (i * (i - 1) / 2 > pr) --i;
            while ((i + 1) * i / 2 <= pr) ++i;
            const int l = i, ii = pr - i * (i - 1) / 2;       // ii < l
            if (view[ii] == view[l]) continue;
            const double *r0 = rays + (size_t)ii * kc * 7, *r1 = rays + (size_t)l * kc * 7;
            double num = 0.0, den = 0.0;
            for (int j = 0; j < kc; ++j) {
                const double *p0 = r0 + j * 7, *p1 = r1 + j * 7;
                const double prod = (p0[0] * p1[3] + p0[1] * p1[4] + p0[2] * p1[5]) +
                                    (p1[0] * p0[3] + p1[1] * p0[4] + p1[2] * p0[5]);
                const double lk = sqrt(p0[6] * p1[6]);
                num = fma(fabs(prod), lk, num);
                den += lk;
            }
            X[ii * ld + l] += num;
            W[ii * n + l] += den;
        }
        lds_fence<NW>();
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
    // distance -> affinity (:397-406), circular constraint (:794-795), matchSVT initialisation (:467-475)
    const double thr = a.recon_thr;
    for (int pr = lane; pr < n * n; pr += NT) {
        const int i = pr / n, l = pr - i * n;
        if (i >= l) continue;
        double aff = 0.0;
        if (i < N && l < N && view[i] != view[l]) {
            double d = X[i * ld + l] / (1e-5 + W[i * n + l]);
            d = d > thr ? thr : d;                           // NaN stays NaN like the reference's comparison
            aff = 1.0 - d / thr;
        }
        X[i * ld + l] = aff; X[l * ld + i] = aff;
        W[i * n + l] = a.w_sparse - aff; W[l * n + i] = a.w_sparse - aff;
    }
    for (int i = lane; i < n; i += NT) { X[i * ld + i] = 0.0; W[i * n + i] = a.w_sparse; }
    for (int i = lane; i < n * ld; i += NT) Y[i] = 0.0;
    lds_fence<NW>();
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");

    if (trace) t_aff = __builtin_amdgcn_s_memtime() - t_start;
    // ---- matchSVT (:477-505) --------------------------------------------------------------
    double mu = 64.0;
    for (int iter = 0; iter < a.max_iter; ++iter) {
        uint64_t tt0 = 0;
        ++n_iter_all;
        if (trace) { tt0 = __builtin_amdgcn_s_memtime(); ++n_iter; }
        // SVT input B = X + Y/mu (:480), column-major for the column rotations.
        // First iteration: A = B, V = I.  Later iterations warm-start from the previous right singular
        // vectors: B changes little from one ADMM iteration to the next, so A = B . V_prev already has
        // nearly orthogonal columns and the Jacobi iteration needs ~2 sweeps instead of ~8.
        if (iter == 0) {
            for (int i = lane; i < n * n; i += NT) {
                const int r = i / n, c = i - r * n;
                A[c * ld + r] = X[r * ld + c] + Y[r * ld + c] * 1.0 / mu;
                V[c * ld + r] = (r == c) ? 1.0 : 0.0;
            }
        } else {
            const int lpr = max(1, NT / n);                   // lanes per matrix row
            const int row = lane % n, part = lane / n;
            const int cpl = (n + lpr - 1) / lpr;              // columns per lane
            const bool on = part < lpr;
            for (int cb = 0; cb < cpl; cb += 16) {
                double acc[16];
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) acc[jj] = 0.0;
                const int j0 = part * cpl + cb;
                const int nj = on ? max(0, min(min(16, cpl - cb), n - j0)) : 0;
                for (int kk = 0; kk < n; ++kk) {
                    const double bk = X[row * ld + kk] + Y[row * ld + kk] * 1.0 / mu;   // B[row][kk]
#pragma unroll
                    for (int jj = 0; jj < 16; ++jj)
                        if (jj < nj) acc[jj] = fma(bk, V[(j0 + jj) * ld + kk], acc[jj]);
                }
#pragma unroll
                for (int jj = 0; jj < 16; ++jj)
                    if (jj < nj) A[(j0 + jj) * ld + row] = acc[jj];
            }
        }
        lds_fence<NW>();
        if (trace) { const uint64_t t = __builtin_amdgcn_s_memtime(); t_prod += t - tt0; tt0 = t; }
        jacobi_svd<NW>(A, V, n, ld, lane, n_sweeps, scratch);
        if (trace) { const uint64_t t = __builtin_amdgcn_s_memtime(); t_svd += t - tt0; tt0 = t; }
        const double tsv = a.w_rank / mu;
        for (int j = lane; j < n; j += NT) {
            double s2 = 0.0;
            for (int r = 0; r < n; ++r) s2 = fma(A[j * ld + r], A[j * ld + r], s2);
            const double sg = sqrt(s2);
            wts[j] = (sg > tsv) ? (sg - tsv) / sg : 0.0;
        }
        lds_fence<NW>();
        double pres2 = 0.0, dres2 = 0.0;
        for (int pr = lane; pr < n * (n + 1) / 2; pr += NT) {
            int l = (int)((sqrt(1.0 + 8.0 * (double)pr) - 1.0) * 0.5);
            while (l * (l + 1) / 2 > pr) --l;
            while ((l + 1) * (l + 2) / 2 <= pr) ++l;
            const int i = pr - l * (l + 1) / 2;               // i <= l
            if (l >= N) continue;                             // zero padding is not part of the problem
            double q_il = 0.0, q_li = 0.0;                    // SVT (:443-445): U diag(max(s-t,0)) Vt
            for (int j = 0; j < n; ++j) {
                const double w = wts[j];
                q_il = fma(w * A[j * ld + i], V[j * ld + l], q_il);
                q_li = fma(w * A[j * ld + l], V[j * ld + i], q_li);
            }
            const bool same_view = view[i] == view[l];
            double x_il = q_il - (W[i * n + l] + Y[i * ld + l]) / mu;   // :482
            double x_li = q_li - (W[l * n + i] + Y[l * ld + i]) / mu;
            if (same_view) { x_il = 0.0; x_li = 0.0; }        // :485-487
            if (i == l) { x_il = 1.0; x_li = 1.0; }           // :490
            x_il = x_il < 0.0 ? 0.0 : x_il; x_il = x_il > 1.0 ? 1.0 : x_il;   // :491-492
            x_li = x_li < 0.0 ? 0.0 : x_li; x_li = x_li > 1.0 ? 1.0 : x_li;
            const double cc = (same_view && i != l) ? 0.0 : 1.0;              // :495
            x_il *= cc; x_li *= cc;
            const double sym = (x_il + x_li) / 2;             // :496
            const double old_il = X[i * ld + l], old_li = X[l * ld + i];
            Y[i * ld + l] = Y[i * ld + l] + mu * (sym - q_il);  // :497
            pres2 += (sym - q_il) * (sym - q_il);
            dres2 += (sym - old_il) * (sym - old_il);
            if (i != l) {
                Y[l * ld + i] = Y[l * ld + i] + mu * (sym - q_li);
                pres2 += (sym - q_li) * (sym - q_li);
                dres2 += (sym - old_li) * (sym - old_li);
            }
            X[i * ld + l] = sym; X[l * ld + i] = sym;
        }
        lds_fence<NW>();
        if (trace) t_upd += __builtin_amdgcn_s_memtime() - tt0;
        const double pRes = sqrt(block_sum<NW>(pres2, scratch, lane)) / (double)N;          // :500
        const double dRes = mu * sqrt(block_sum<NW>(dres2, scratch, lane)) / (double)N;     // :501
        if (pRes < a.tol && dRes < a.tol) break;                        // :502
        if (pRes > 10 * dRes) mu = 2 * mu;                              // :504
        else if (dRes > 10 * pRes) mu = mu / 2;                         // :505
    }
    // ---- min_affinity cut (:800) and store (the slab held W until now) ---------------------------
    for (int i = lane; i < n_max * n_max; i += NT) out[i] = 0.0;
    lds_fence<NW>();                                 // the zeros of one wave must not land on the values of the other
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    for (int pr = lane; pr < N * N; pr += NT) {
        const int i = pr / N, l = pr - i * N;
        const double v = X[i * ld + l];
        out[i * n_max + l] = (v < a.min_affinity) ? 0.0 : v;
    }
    if (a.stats && lane == 0) {
        // fp64 operations of this frame (p2s_get_assoc_stats): a column-pair rotation is 3 dot products, 4 columns of
        // n rows updated (A and V) and ~50 operations of rotation arithmetic; a pass also has the warm-start product,
        // the rebuilt U diag Vt for both triangles and the element-wise update
        const unsigned long long nn = (unsigned long long)n;
        const unsigned long long flops = (unsigned long long)n_sweeps * (nn - 1) * (nn / 2) * (18 * nn + 50) +
                                         (unsigned long long)n_iter_all * (2 * nn * nn * nn + 3 * nn * nn * nn + 30 * nn * nn) +
                                         (unsigned long long)n_pairs * Kj * 17;
        assoc_count(a, 1, (unsigned long long)n_iter_all, (unsigned long long)n_sweeps, flops);
    }
    if (trace) {
        lds_fence<NW>();
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (lane == 0) {
            out[0] = (double)(__builtin_amdgcn_s_memtime() - t_start); out[1] = (double)t_aff; out[2] = (double)t_prod;
            out[3] = (double)t_svd; out[4] = (double)t_upd; out[5] = (double)n_sweeps; out[6] = (double)n_iter; out[7] = (double)N;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Symmetric form for up to 32 detections: ONE wave per frame, no workgroup barrier anywhere.
//
// The SVT input B = X + Y/mu of matchSVT (:480) is symmetric to rounding (X is symmetrised at :496, Y only ever
// receives X - Q with Q = U diag(.) Vt of a symmetric matrix), so its SVD is its eigendecomposition with the signs
// moved into U.  With a shift c >= ||B||_F the matrix M = B + cI is positive definite: the one-sided Jacobi iteration
// on M then needs NO accumulated V -- once the columns of G = M V are orthogonal, v_j = g_j / |g_j| and the
// eigenvalue of B is lambda_j = |g_j| - c -- and Q = sum_j sgn(lambda_j) max(|lambda_j| - t, 0) v_j v_j^T.  That halves
// the rotation work of a step and frees one LDS matrix.  What else is different from p2s_assoc_kernel:
//   * odd-even ordering with one column of every pair resident in registers (jacobi_oe below): one column read from
//     LDS and one written per step;
//   * the squared column norms travel with the columns (alpha' = alpha - t gamma, beta' = beta + t gamma; the first
//     padding element of the LDS slot and a register, refreshed from the data in the first step of every sweep), so a
//     step reduces one dot product over the L lanes of a column pair instead of three;
//   * a step is issue bound, not latency bound (every fp64 or 32-bit VALU instruction costs a wave ~4 cycles of its
//     SIMD), so the frame gets one wave (L = 4 lanes x 8 rows per column pair at 32 detections: the rotation's ~25
//     scalar-like instructions are issued once per frame instead of once per wave of the frame) and the CU hides the
//     chain latency with 12 frames = 3 waves per SIMD: G (32 x 34), the packed triangle of B (its last row in G's
//     second padding element) and the view list are 12 800 B of LDS, 10 allocation granules of 1 280 B;
//   * X, Y and W of matchSVT are symmetric element-wise state in the registers of the lane that owns the pair
//     (lane + 64 k is the pair's index in the packed triangle), 9 pairs per lane;
//   * the next pass starts from V of this one (G <- (B' + c'I) V: B changes little between ADMM passes).
// The rotation, thresholds, update arithmetic and stopping rules are those of p2s_assoc_kernel; results differ from
// it by rounding (tests/test_assoc_gpu.py runs both on the same frames).
template <int L>
__device__ __forceinline__ void group_sum1(double &c) {
    if constexpr (L >= 2) c += dpp_move<0xB1>(c);
    if constexpr (L >= 4) c += dpp_move<0x4E>(c);
    if constexpr (L >= 8) c += dpp_move<0x141>(c);
    if constexpr (L >= 16) c += dpp_move<0x140>(c);
}

__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_move<0xB1>(v); v += dpp_move<0x4E>(v); v += dpp_move<0x141>(v); v += dpp_move<0x140>(v);
    v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    return v;
}

// One step of the odd-even ordering for the group's pair (incoming column = LDS slot `col`, kept column = registers):
// rotate, then the two change places -- the rotated kept column goes to the slot, the rotated incoming one stays in
// registers.  FRESH: the squared norms come from the data (first step of a sweep: every column is in a pair),
// otherwise the incoming one from the slot's padding element and the kept one from a register.  Straight-line code:
// a group without a pair in this step (act false) runs the same instructions on a slot nobody else uses, with the
// "rotation" (c, s) = (0, -1) that leaves its kept column where it is.  Returns whether this lane's pair was further
// than the sweep tolerance from orthogonal.
template <int L, int RPL, bool FRESH>
__device__ __forceinline__ bool oe_step(double *col, bool act, int sub, double (&kept)[RPL], double &bk) {
    constexpr int R = L * RPL;
    double *mine = col + sub;                            // rows sub, sub + L, ...: see the bank arithmetic at the kernel
    double x[RPL];
#pragma unroll
    for (int i = 0; i < RPL; ++i) x[i] = mine[L * i];
    double al, ga = 0.0;
#pragma unroll
    for (int i = 0; i < RPL; ++i) ga = fma(x[i], kept[i], ga);
    if constexpr (FRESH) {
        al = 0.0; bk = 0.0;
#pragma unroll
        for (int i = 0; i < RPL; ++i) { al = fma(x[i], x[i], al); bk = fma(kept[i], kept[i], bk); }
        group_sum3<L>(al, bk, ga);
    } else {
        al = col[R];
        group_sum1<L>(ga);
    }
    const double ab = al * bk, g2 = ga * ga;
    const bool rot = act && (g2 > 1e-30 * ab);           // false for the all-zero padding columns and for NaN
    // Rutishauser's formulas.  The tangent comes straight from the hardware seeds of 1/x and 1/sqrt(x) (5e-8,
    // exp/seed_precision.hip): a relative error e of the angle leaves e times the pair's cosine behind, and the sweep
    // that ends the iteration only sees cosines under 1e-8.  The orthogonality of the rotation hangs on c, which keeps
    // its two Newton steps.
    const double zeta = (bk - al) * __builtin_amdgcn_rcp(2.0 * ga);
    const double zz = fma(zeta, zeta, 1.0);
    const double tt = __builtin_amdgcn_rcp(fma(zz, __builtin_amdgcn_rsq(zz), fabs(zeta)));
    const double t = rot ? __builtin_copysign(tt, zeta) : 0.0;
    double c = p2s_rsqrt(fma(t, t, 1.0)), sn = c * t;
    const double d = t * ga;
    c = act ? c : 0.0; sn = act ? sn : -1.0;
#pragma unroll
    for (int i = 0; i < RPL; ++i) {
        mine[L * i] = fma(sn, x[i], c * kept[i]);
        kept[i] = fma(c, x[i], -(sn * kept[i]));
    }
    col[R] = bk + d;                                     // the L lanes of the group store the same value
    bk = act ? al - d : bk;
    // a pair orthogonal to 1e-8 is orthogonal to rounding after its rotation (1e-7 here: 5 % less time and 3.6e-11
    // instead of 1.4e-13 from the oracle on tests/sweeps/sweep_assoc.py)
    return rot && (g2 > 1e-16 * ab);
}

// One-sided Jacobi iteration on the n x n matrix in G (column-major, leading dimension LD, rows and columns n .. R-1
// zero) until a whole sweep found every pair orthogonal to 1e-8.  Odd-even ordering: the columns stand in a line,
// even steps pair the positions (2k, 2k+1), odd steps (2k+1, 2k+2), and a pair changes places after its rotation, so
// that after n steps every two columns have met once (the line is then reversed).  Group k of L lanes keeps the column
// at position 2k+1 in registers and only ever exchanges the other one with the LDS slots 2k (even steps) and 2k+2
// (odd steps): one column read and one written per step instead of two and two -- the stores are what the CU's LDS
// path runs out of first (13 cycles per 1 KiB store instruction, MI355X_MICROARCH.md LDS).  The odd slots are not
// used while the iteration runs (they are rewritten at the end): a group without a pair plays with its own.
// Returns the sweeps.
template <int L, int RPL>
__device__ __forceinline__ int jacobi_oe(double *G, int n, int lane) {
    constexpr int R = L * RPL, LD = R + L / 2;
    const int k = lane / L, sub = lane % L, half = n >> 1;
    double *odd = G + (2 * k + 1) * LD;
    const bool actA = k < half, actB = k < half - 1;
    double *colA = actA ? odd - LD : odd, *colB = actB ? odd + LD : odd;
    double kept[RPL], bk = 0.0;
#pragma unroll
    for (int i = 0; i < RPL; ++i) kept[i] = odd[sub + L * i];
    int sweeps = 0;
    for (int sweep = 0; sweep < 40; ++sweep) {
        ++sweeps;
        bool far = oe_step<L, RPL, true>(colA, actA, sub, kept, bk);
        lds_fence<1>();
        for (int j = 1; j < half; ++j) {
            far = oe_step<L, RPL, false>(colB, actB, sub, kept, bk) || far;
            lds_fence<1>();
            far = oe_step<L, RPL, false>(colA, actA, sub, kept, bk) || far;
            lds_fence<1>();
        }
        far = oe_step<L, RPL, false>(colB, actB, sub, kept, bk) || far;
        lds_fence<1>();
        if (!__any(far)) break;
    }
#pragma unroll
    for (int i = 0; i < RPL; ++i) odd[sub + L * i] = kept[i];
    return sweeps;
}

template <typename T, int L, int RPL>
__global__ void __launch_bounds__(64, 3) p2s_assoc_kernel_s(const P2sAssocArgs a) {
    constexpr int R = L * RPL;                      // padded order: 32 (L = 4, RPL = 8) or 16 (L = 8, RPL = 2)
    // Lane (k, sub) of the Jacobi steps works on rows sub, sub + L, ... of the even columns 2k and 2k + 2.  An 8-byte
    // LDS read serves 32 lanes at a time from 32 double-wide banks and a store 16 lanes from 16: with 2 LD = L (mod 32)
    // the 32 / L groups of a half wave start L banks apart and the L lanes of a group fill the gap -- no conflicts.
    constexpr int LD = R + L / 2;
    constexpr int NPK = R * (R + 1) / 2;            // packed triangle of B
    constexpr int KP = (NPK + 63) / 64;             // pairs per lane while the affinity accumulates (pair lane + 64 k)
    constexpr int KR = (R == 32) ? 11 : 3;          // pairs per lane in the ADMM loop: up to KR neighbours of one matrix row
    constexpr int PARTS = 64 / R, CPL = R / PARTS;  // warm-start product: lane = (row, part), CPL columns per lane
    extern __shared__ __align__(16) unsigned char smem[];
    double *G = reinterpret_cast<double *>(smem);   // [R][LD]; element R of a column: its squared norm, later its SVT weight
    // B[i][l] = B[l][i], i <= l, at packed index l (l + 1) / 2 + i: rows l < R - 1 behind G, the last row (R entries) in
    // the second padding element of G's columns -- the 256 bytes between 11 and 12 frames per CU
    constexpr int NPKS = NPK - R;
    double *Bp = G + R * LD;
    auto bp = [&](int idx) -> double * { return idx < NPKS ? Bp + idx : G + (idx - NPKS) * LD + (R + 1); };
    int *view = reinterpret_cast<int *>(Bp + NPKS);
    double *rays = G;                               // [person][joint in chunk][7], aliases G and Bp
    const int lane = threadIdx.x;
    const int64_t f = blockIdx.x;
    const int C = a.C, Kj = a.Kj, n_max = a.Nmax;   // n_max even, <= R
    double *out = a.affinity + f * (int64_t)n_max * n_max;

    int N = 0;
    for (int c = 0; c < C; ++c) {
        const int pc = a.n_persons[f * C + c];
        for (int i = lane; i < pc; i += 64)
            if (N + i < n_max) view[N + i] = c;
        N += pc;
    }
    N = min(N, n_max);
    const int n = max(2, (N + 1) & ~1);
    for (int i = N + lane; i < R; i += 64) view[i] = -1 - i;
    for (int i = lane; i < n_max * n_max; i += 64) out[i] = 0.0;
    lds_fence<1>();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // the zeros land before the result of another lane
    if (N == 0) return;

    // Two ways of dealing the n (n + 1) / 2 pairs (i <= l) of the symmetric matrices to the lanes.  While the affinity
    // accumulates: pair lane + 64 k of the packed triangle, KP per lane, evenly loaded.  In the ADMM loop: up to KR
    // consecutive i of ONE row l per lane (rows of more than KR entries are cut into equal chunks: 63 chunks at
    // R = 32), so that rebuilding Q = V diag(h) V^T reads one element of every column for the row and KR consecutive
    // ones for the i -- two base addresses and immediate offsets instead of 2 KP scattered addresses per column.
    int pi[KP], pl[KP];
    bool own[KP];                                   // a pair of real detections (not the zero padding)
    double num[KP], den[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        const int pr = lane + k * 64;
        int l = (int)((sqrt(1.0 + 8.0 * (double)pr) - 1.0) * 0.5);
        while (l * (l + 1) / 2 > pr) --l;
        while ((l + 1) * (l + 2) / 2 <= pr) ++l;
        pl[k] = l;
        pi[k] = pr - l * (l + 1) / 2;               // i <= l
        own[k] = l < N;
        if (!own[k]) { pl[k] = 0; pi[k] = 0; }
        num[k] = 0.0; den[k] = 0.0;
    }
    unsigned sv = 0;                                // bit k: the two detections of pair k are in the same view (or are the same)
#pragma unroll
    for (int k = 0; k < KP; ++k) sv |= (view[pi[k]] == view[pl[k]]) ? (1u << k) : 0u;

    const T *kp = reinterpret_cast<const T *>(a.kpts) + a.offsets[f] * (int64_t)Kj * 3;
    const bool trace = P2S_DEBUG_MODE(a) == 7;
    uint64_t t_start = 0, t_aff = 0, t_prod = 0, t_svd = 0, t_upd = 0;
    int n_sweeps = 0, n_iter = 0, n_iter_all = 0;
    if (trace) t_start = __builtin_amdgcn_s_memtime();

    // ---- rays + affinity accumulation, a chunk of joints at a time ------------------------------
    int Kc = (R * LD + NPKS) / (7 * N);
    Kc = max(1, min(Kc, Kj));
    for (int j0 = 0; j0 < Kj; j0 += Kc) {
        const int kc = min(Kc, Kj - j0);
        for (int it = lane; it < N * kc; it += 64) {          // compute_rays (:293-314)
            const int i = it / kc, j = it - i * kc;
            const P2sCam &cam = a.cams[view[i]];
            const T *o = kp + ((int64_t)i * Kj + j0 + j) * 3;
            const double ox = (double)o[0], oy = (double)o[1], lik = (double)o[2];
            const double v0 = cam.iK[0] * ox + cam.iK[1] * oy + cam.iK[2] - cam.T[0];
            const double v1 = cam.iK[3] * ox + cam.iK[4] * oy + cam.iK[5] - cam.T[1];
            const double v2 = cam.iK[6] * ox + cam.iK[7] * oy + cam.iK[8] - cam.T[2];
            const double l0 = cam.R[0] * v0 + cam.R[3] * v1 + cam.R[6] * v2 - cam.center[0];
            const double l1 = cam.R[1] * v0 + cam.R[4] * v1 + cam.R[7] * v2 - cam.center[1];
            const double l2 = cam.R[2] * v0 + cam.R[5] * v1 + cam.R[8] * v2 - cam.center[2];
            const double nrm = sqrt(l0 * l0 + l1 * l1 + l2 * l2);
            double d0 = l0 / nrm, d1 = l1 / nrm, d2 = l2 / nrm;
            double m0 = cam.center[1] * d2 - cam.center[2] * d1;
            double m1 = cam.center[2] * d0 - cam.center[0] * d2;
            double m2 = cam.center[0] * d1 - cam.center[1] * d0;
            double lk = lik;
            const bool anynan = !(d0 == d0) || !(d1 == d1) || !(d2 == d2) || !(m0 == m0) || !(m1 == m1) ||
                                !(m2 == m2) || !(lk == lk);
            if (anynan) { d0 = d1 = d2 = m0 = m1 = m2 = lk = 0.0; }
            double *r = rays + ((size_t)i * kc + j) * 7;
            r[0] = d0; r[1] = d1; r[2] = d2; r[3] = m0; r[4] = m1; r[5] = m2; r[6] = lk;
        }
        lds_fence<1>();
#pragma unroll
        for (int k = 0; k < KP; ++k) {                        // compute_affinity (:383-394) for the owned pairs
            if (!own[k] || ((sv >> k) & 1)) continue;
            const double *r0 = rays + (size_t)pi[k] * kc * 7, *r1 = rays + (size_t)pl[k] * kc * 7;
            double nm = 0.0, dn = 0.0;
            for (int j = 0; j < kc; ++j) {
                const double *p0 = r0 + j * 7, *p1 = r1 + j * 7;
                const double prod = (p0[0] * p1[3] + p0[1] * p1[4] + p0[2] * p1[5]) +
                                    (p1[0] * p0[3] + p1[1] * p0[4] + p1[2] * p0[5]);
                const double lk = sqrt(p0[6] * p1[6]);
                nm = fma(fabs(prod), lk, nm);
                dn += lk;
            }
            num[k] += nm;
            den[k] += dn;
        }
        lds_fence<1>();
    }
    // distance -> affinity (:397-406), circular constraint (:794-795), matchSVT initialisation (:467-475)
    const double thr = a.recon_thr;
    for (int i = lane; i < R * LD + NPKS; i += 64) G[i] = 0.0;         // G (with the last row of B) and Bp
    lds_fence<1>();
    double sx = 0.0, sy = 0.0;                                         // ||X||_F^2, ||Y||_F^2 over the full matrices
    int cross = 0;                                                     // cross-view pairs (p2s_get_assoc_stats)
#pragma unroll
    for (int k = 0; k < KP; ++k) {
        if (!own[k] || ((sv >> k) & 1)) continue;                      // the diagonal and same-view pairs stay 0
        double d = num[k] / (1e-5 + den[k]);
        d = d > thr ? thr : d;
        const double aff = 1.0 - d / thr;
        *bp(lane + k * 64) = aff;                                      // B = X + Y/mu with Y = 0
        sx = fma(2.0 * aff, aff, sx);
        ++cross;
    }
    sx = wave_sum_dpp(sx);
    lds_fence<1>();
    // the ADMM loop's pairs: row rl, columns ri0 .. ri0 + rcnt - 1
    int rl = 0, ri0 = 0, rcnt = 0;
    {
        int first = 0;                                                 // first chunk of row l
        for (int l = 0; l < R; ++l) {
            const int m = l + 1, c = (m + KR - 1) / KR, sz = (m + c - 1) / c;
            if (lane >= first && lane < first + c) { rl = l; ri0 = (lane - first) * sz; rcnt = min(sz, m - ri0); }
            first += c;
        }
        if (rl >= N) rcnt = 0;                                         // zero padding is not part of the problem
    }
    double *const pb = bp(rl * (rl + 1) / 2 + ri0);                    // B[rl][ri0 + k] at pb[k * pbs]
    const int pbs = (rl == R - 1) ? LD : 1;
    const int dk = rl - ri0;                                           // the diagonal element is pair dk (if < rcnt)
    unsigned sv_r = 0;                                                 // bit k: same view (or the diagonal)
    double x[KR], y[KR], w[KR];
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        const bool mine = k < rcnt;
        const int kk = mine ? k : 0;                                   // a valid address for the lanes without pair k
        sv_r |= (mine && view[ri0 + kk] == view[rl]) ? (1u << k) : 0u;
        x[k] = mine ? pb[kk * pbs] : 0.0;                              // X = the affinity, 0 on the diagonal (:467-475)
        w[k] = a.w_sparse - x[k];
        y[k] = 0.0;
    }
    if (trace) t_aff = __builtin_amdgcn_s_memtime() - t_start;

    // ---- matchSVT (:477-505) --------------------------------------------------------------
    double mu = 64.0, inv_mu = 1.0 / 64.0;                             // mu stays a power of two: x / mu == x * inv_mu
    // the shift: c >= ||X||_F + ||Y||_F / mu >= ||B||_2, with a margin that keeps M = B + cI well conditioned
    double shift = fma(1.0625, sqrt(sx), 0.5);
    for (int iter = 0; iter < a.max_iter; ++iter) {
        uint64_t tt0 = 0;
        ++n_iter_all;
        if (trace) { tt0 = __builtin_amdgcn_s_memtime(); ++n_iter; }
        {
            const int row = lane % R, part = lane / R, j0 = part * CPL;
            double acc[CPL];
            if (iter == 0) {                                  // V = I: G = M
#pragma unroll
                for (int jj = 0; jj < CPL; ++jj) {
                    const int j = j0 + jj, lo = min(row, j), hi = max(row, j);
                    acc[jj] = *bp(hi * (hi + 1) / 2 + lo) + ((row == j && row < n) ? shift : 0.0);
                }
            } else {                                          // warm start: G <- M V, V = the normalised columns of G
#pragma unroll
                for (int jj = 0; jj < CPL; ++jj) acc[jj] = shift * G[(j0 + jj) * LD + row];
                int idx = row * (row + 1) / 2;
                const double *vrow = G + j0 * LD;
                for (int kk = 0; kk < n; ++kk) {                  // rolled: unrolled by 4 or fully it spills and is 3-4 % slower
                    const double bk = *bp(idx);               // B[row][kk]
                    idx += (kk < row) ? 1 : kk + 1;
#pragma unroll
                    for (int jj = 0; jj < CPL; ++jj) acc[jj] = fma(bk, vrow[jj * LD + kk], acc[jj]);
                }
            }
            lds_fence<1>();                                   // every lane has read V before anyone overwrites it
#pragma unroll
            for (int jj = 0; jj < CPL; ++jj) G[(j0 + jj) * LD + row] = acc[jj];
        }
        lds_fence<1>();
        if (trace) { const uint64_t t = __builtin_amdgcn_s_memtime(); t_prod += t - tt0; tt0 = t; }
        n_sweeps += jacobi_oe<L, RPL>(G, n, lane);
        if (trace) { const uint64_t t = __builtin_amdgcn_s_memtime(); t_svd += t - tt0; tt0 = t; }
        // singular values sigma_j = |g_j| of M, eigenvalues lambda_j = sigma_j - c of B; G <- V; the SVT weight
        // sgn(lambda) max(|lambda| - t, 0) (:443-445 with u_j = sgn(lambda_j) v_j) goes to the column's padding element
        const double tsv = a.w_rank / mu;
        {
            const int k = lane / L, sub = lane % L;
#pragma unroll
            for (int jb = 0; jb < R; jb += 64 / L) {
                double *col = G + (jb + k) * LD;
                double v[RPL], s2 = 0.0;
#pragma unroll
                for (int i = 0; i < RPL; ++i) { v[i] = col[sub + L * i]; s2 = fma(v[i], v[i], s2); }
                group_sum1<L>(s2);
                const double is = (s2 > 0.0) ? p2s_rsqrt(s2) : 0.0;
                const double lam = s2 * is - shift;
                const double mag = fabs(lam) - tsv;
#pragma unroll
                for (int i = 0; i < RPL; ++i) col[sub + L * i] = v[i] * is;
                if (sub == 0) col[R] = (mag > 0.0) ? __builtin_copysign(mag, lam) : 0.0;
            }
        }
        lds_fence<1>();
        double q[KR];
#pragma unroll
        for (int k = 0; k < KR; ++k) q[k] = 0.0;
        {
            const double *gl = G + rl, *gi = G + ri0;
#pragma unroll 2
            for (int j = 0; j < n; ++j) {
                const double hl = G[j * LD + R] * gl[j * LD];               // h_j v_j[l]
#pragma unroll
                for (int k = 0; k < KR; ++k) q[k] = fma(hl, gi[j * LD + k], q[k]);
            }
        }
        double pres2 = 0.0, dres2 = 0.0;
        sx = 0.0; sy = 0.0;
#pragma unroll
        for (int k = 0; k < KR; ++k) {
            if (k >= rcnt) continue;
            const bool diag = k == dk;
            double xn = q[k] - (w[k] + y[k]) * inv_mu;        // :482
            if ((sv_r >> k) & 1) xn = 0.0;                    // :485-487
            if (diag) xn = 1.0;                               // :490
            xn = xn < 0.0 ? 0.0 : xn; xn = xn > 1.0 ? 1.0 : xn;   // :491-492
            // :495 multiplies the same-view entries by 0 once more; :496 (X + X.T) / 2 of a symmetric X is X
            const double dq = xn - q[k], dx = xn - x[k];
            const double mult = diag ? 1.0 : 2.0;             // both triangles
            y[k] = fma(mu, dq, y[k]);                         // :497
            pres2 = fma(mult * dq, dq, pres2);
            dres2 = fma(mult * dx, dx, dres2);
            x[k] = xn;
            sx = fma(mult * xn, xn, sx);
            sy = fma(mult * y[k], y[k], sy);
        }
        if (trace) t_upd += __builtin_amdgcn_s_memtime() - tt0;
        pres2 = wave_sum_dpp(pres2); dres2 = wave_sum_dpp(dres2);
        const double pRes = sqrt(pres2) / (double)N;                    // :500
        const double dRes = mu * sqrt(dres2) / (double)N;               // :501
        if (pRes < a.tol && dRes < a.tol) break;                        // :502
        if (pRes > 10 * dRes) { mu = 2 * mu; inv_mu = 0.5 * inv_mu; }   // :504
        else if (dRes > 10 * pRes) { mu = mu / 2; inv_mu = 2 * inv_mu; }   // :505
        sx = wave_sum_dpp(sx); sy = wave_sum_dpp(sy);
        shift = fma(1.0625, sqrt(sx) + sqrt(sy) * inv_mu, 0.5);
        // the next pass's SVT input, B = X + Y/mu (:480), from the owners
#pragma unroll
        for (int k = 0; k < KR; ++k)
            if (k < rcnt) pb[k * pbs] = x[k] + y[k] * inv_mu;
        lds_fence<1>();
    }
    // ---- min_affinity cut (:800) and store --------------------------------------------------------
#pragma unroll
    for (int k = 0; k < KR; ++k) {
        if (k >= rcnt) continue;
        const double v = (x[k] < a.min_affinity) ? 0.0 : x[k];
        out[(ri0 + k) * n_max + rl] = v;
        out[rl * n_max + ri0 + k] = v;
    }
    if (a.stats) {
        // fp64 operations of this frame (p2s_get_assoc_stats): a column-pair step is one dot product and two columns of n
        // rows updated plus ~50 operations of rotation arithmetic (the first step of a sweep also takes both norms); a
        // pass has the warm-start product, the normalisation, V diag Vt for one triangle and the element-wise update
        cross = (int)wave_sum_dpp((double)cross);
        if (lane == 0) {
            const unsigned long long nn = (unsigned long long)n;
            const unsigned long long flops = (unsigned long long)n_sweeps * (nn * (nn / 2) * (8 * nn + 50) + (nn / 2) * 4 * nn) +
                                             (unsigned long long)n_iter_all * (2 * nn * nn * nn + 3 * nn * nn + 3 * nn * (nn * (nn + 1) / 2) + 20 * (nn * (nn + 1) / 2)) +
                                             (unsigned long long)cross * Kj * 17;
            assoc_count(a, 1, (unsigned long long)n_iter_all, (unsigned long long)n_sweeps, flops);
        }
    }
    if (trace) {
        lds_fence<1>();
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        if (lane == 0) {
            out[0] = (double)(__builtin_amdgcn_s_memtime() - t_start); out[1] = (double)t_aff; out[2] = (double)t_prod;
            out[3] = (double)t_svd; out[4] = (double)t_upd; out[5] = (double)n_sweeps; out[6] = (double)n_iter; out[7] = (double)N;
        }
    }
}

hipError_t p2s_launch_assoc(const P2sAssocArgs &a, int dtype, hipStream_t s) {
    if (a.Nmax <= 32 && a.form != P2S_ASSOC_FORM_GENERAL) {           // symmetric form, one wave per frame
        auto go_s = [&](auto kern, int R) -> hipError_t {
            const int LD = R + (R == 32 ? 2 : 4);                 // R + L / 2, as in the kernel
            const size_t lds = (size_t)(R * LD + R * (R - 1) / 2) * sizeof(double) + (size_t)R * sizeof(int);
            hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (err != hipSuccess) return err;
            hipLaunchKernelGGL(kern, dim3((unsigned)a.n_frames), dim3(64), lds, s, a);
            return hipGetLastError();
        };
        if (a.Nmax <= 16) return dtype == P2S_F32 ? go_s(&p2s_assoc_kernel_s<float, 8, 2>, 16) : go_s(&p2s_assoc_kernel_s<double, 8, 2>, 16);
        return dtype == P2S_F32 ? go_s(&p2s_assoc_kernel_s<float, 4, 8>, 32) : go_s(&p2s_assoc_kernel_s<double, 4, 8>, 32);
    }
    // general form: no symmetry assumed, V accumulated, four LDS matrices; two waves per frame from 18 detections up
    // (the Jacobi step is a chain of dependent operations and ~34 KB of LDS per frame leave one wave per SIMD otherwise)
    const size_t lds = (size_t)(4 * a.Nmax * (a.Nmax + 1) + a.Nmax + 4) * sizeof(double) + (size_t)a.Nmax * sizeof(int) + 16;
    const bool two = a.Nmax > 16;
    auto go = [&](auto kern, int threads) -> hipError_t {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (err != hipSuccess) return err;
        hipLaunchKernelGGL(kern, dim3((unsigned)a.n_frames), dim3(threads), lds, s, a);
        return hipGetLastError();
    };
    if (dtype == P2S_F32) return two ? go(&p2s_assoc_kernel<float, 2>, 128) : go(&p2s_assoc_kernel<float, 1>, 64);
    return two ? go(&p2s_assoc_kernel<double, 2>, 128) : go(&p2s_assoc_kernel<double, 1>, 64);
}
