// p2s_tri.hip -- fused robust multi-view triangulation for gfx950 (MI355X, CDNA4).
//
// One kernel does, per (block = frame x person, keypoint) unit, everything the reference does in
// triangulate_all's frame/person/keypoint loops (triangulation.py:796-845) and in
// triangulation_from_best_cameras (triangulation.py:363-604):
//
//   stage    a tile of FB consecutive blocks ([FB][C][K][3], contiguous in HBM) is copied into LDS
//            with 16-byte-per-lane coalesced loads; the C projection matrices sit beside it
//   prepare  (undistort only) each lane undistorts the observations of its unit in place in LDS
//            (triangulation.py:808-813); the likelihood mask (:817-821) is applied on the fly
//   level 0  one lane per unit: weighted-DLT normal matrix (common.py:327-354 restated as the
//            smallest eigenpair of A^T A, fp64), reprojection error (common.py:357-403);
//            projection matrices come through the scalar cache (SGPR operands), loops are
//            branch-free so that LDS reads of consecutive cameras overlap
//   search   units whose error exceeds the threshold are handed to groups of G lanes of the same
//            wavefront; lane j of a group evaluates camera subset #(round*G + j) of the level
//            (lexicographic itertools.combinations order, triangulation.py:411), the group's
//            argmin (first index on ties, :502) is taken with wave shuffles; levels proceed in
//            lock step across the wave (ballot of units still above threshold), and G is chosen
//            per level to minimise passes x rounds
//
// Everything is data-parallel fp64 VALU work on an HBM-streamed tensor: no MFMA (4x4 systems).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include <math.h>

#include "p2s_internal.h"

#include "p2s_tri_dev.h"

// ---------------------------------------------------------------------------------------------
// Kernel 1 -- streaming pass.  LDS: [tile: FB*C*K*3 of T].
// Every unit gets its level-0 result; units that must enter the camera-subset search append a
// compact record (unit id + their C observations as seen after undistortion) to the work list.
template <typename T, bool UNDISTORT, bool LRSWAP>
__global__ void __launch_bounds__(256, 4) p2s_tri_level0_kernel(const P2sTriArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int C = a.C, K = a.K, FB = a.FB;
    const int blk_elems = C * K * 3;
    const int strideC = K * 3;
    T *tile = reinterpret_cast<T *>(smem);
    cam_cptr cams = (cam_cptr)a.cams;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int64_t tile0 = a.block0 + (int64_t)blockIdx.x * FB;     // first block of this tile
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, st5 = 0;   // diagnostics (debug_mode 4)
    if (P2S_DEBUG_MODE(a) == 4) st0 = __builtin_amdgcn_s_memtime();
    const int nb = (int)min((int64_t)FB, a.block0 + a.n_blocks - tile0);
    const int n_units = nb * K;

    // ---- stage ---------------------------------------------------------------------------
    {
        const T *src = reinterpret_cast<const T *>(a.xyl) + (P2S_DEBUG_MODE(a) == 2 ? 0 : tile0 * blk_elems);
        const int n_elems = nb * blk_elems;
        constexpr int VEC = 16 / sizeof(T);
        const int n_vec = n_elems / VEC;                           // tile base is 16-B aligned (host)
        typedef uint32_t v4u __attribute__((ext_vector_type(4)));
        const v4u *src4 = reinterpret_cast<const v4u *>(src);
        v4u *dst4 = reinterpret_cast<v4u *>(tile);
        for (int i = tid; i < n_vec; i += blockDim.x) dst4[i] = __builtin_nontemporal_load(src4 + i);
        for (int i = n_vec * VEC + tid; i < n_elems; i += blockDim.x) tile[i] = src[i];
    }
    __syncthreads();
    if (P2S_DEBUG_MODE(a) == 4) st1 = __builtin_amdgcn_s_memtime();

    // ---- prepare: undistort in place (the mirrored keypoint of another lane reads it too) ----
    if (UNDISTORT) {
        for (int u = tid; u < n_units; u += blockDim.x) {
            const int b = u / K, k = u - b * K;
            T *o = tile + (size_t)b * blk_elems + k * 3;
            for (int c = 0; c < C; ++c) {
                T *p = o + c * strideC;
                double x = (double)p[0], y = (double)p[1];
                undistort_point(cams + c, x, y);
                p[0] = (T)x; p[1] = (T)y;
            }
        }
        __syncthreads();
    }

    const double thr = a.thr;
    const uint32_t allmask = (C == 32) ? 0xffffffffu : ((1u << C) - 1u);

    if (P2S_DEBUG_MODE(a) == 1) {   // diagnostics: memory skeleton only
        for (int u = tid; u < n_units; u += blockDim.x) {
            const int b = u / K, k = u - b * K;
            const T *o = tile + (size_t)b * blk_elems + k * 3;
            double s = 0;
            for (int c = 0; c < C; ++c) s += (double)o[c * strideC] + (double)o[c * strideC + 1] + (double)o[c * strideC + 2];
            const int64_t gu = tile0 * K + u;
            a.Q[gu * 3] = s; a.Q[gu * 3 + 1] = s; a.Q[gu * 3 + 2] = s;
            a.err[gu] = (float)s; a.n_excl[gu] = 0; a.mask[gu] = 0;
        }
        return;
    }

    for (int base = 0; base < n_units; base += blockDim.x) {
        if (base + (tid & ~63) >= n_units) break;                  // whole wave past the tile
        const int u = base + tid;
        const bool active = u < n_units;
        const int b = active ? u / K : 0, k = active ? u - b * K : 0;
        UnitObs<T> obs{tile + (size_t)b * blk_elems + k * 3, strideC, a.lik_thr};
        UnitObs<T> obs_sw = obs;
        if (LRSWAP) obs_sw.p = tile + (size_t)b * blk_elems + a.swap_idx[k] * 3;

        double N[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) N[i] = 0.0;
        uint32_t nanmask = 0, zeromask = 0;
        classify_and_accumulate<T, 0>(cams, C, obs, N, nanmask, zeromask);
        if (P2S_DEBUG_MODE(a) == 4) { __builtin_amdgcn_sched_barrier(0); st2 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
        const uint32_t dmask = nanmask | zeromask;                 // cameras already out (NaN or zero likelihood)
        const uint32_t valid = allmask & ~dmask;
        const int V = __popc(dmask);
        const int nvalid = C - V;
        const int Lmax = active ? C - a.min_cams - V : -1;         // last level that runs (triangulation.py:408, 437-441)

        double err_min = kInf;
        double Qb[3] = {d_nan(), d_nan(), d_nan()};
        int n_excl = C;                                            // :595-596 when no level completes
        uint32_t mask = allmask;
        {
            double q[3];
            smallest_eigvec(N, q);
            if (P2S_DEBUG_MODE(a) == 4) { __builtin_amdgcn_sched_barrier(0); st3 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
            if (nvalid < 2) { q[0] = d_nan(); q[1] = d_nan(); q[2] = d_nan(); }   // common.py:347: fewer than 4 rows
            const double e = mean_error<T, UNDISTORT, 0>(cams, C, obs, valid, q);
            if (P2S_DEBUG_MODE(a) == 4) { __builtin_amdgcn_sched_barrier(0); st4 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
            if (Lmax >= 0) {
                err_min = e; Qb[0] = q[0]; Qb[1] = q[1]; Qb[2] = q[2];
                n_excl = V; mask = nanmask;
            }
        }
        if (LRSWAP) {                                              // :509-579 at level 0: M = nvalid
            const bool want = (Lmax >= 0) && (err_min > thr) && (nvalid > 2);
            if (__any(want)) {
                double qs[3];
                const double es = swap_candidate<T, UNDISTORT, 0>(cams, C, obs, obs_sw, valid, nvalid, qs);
                if (want && es < err_min) { err_min = es; Qb[0] = qs[0]; Qb[1] = qs[1]; Qb[2] = qs[2]; }
            }
        }
        const bool need = (Lmax >= 1) && (err_min > thr);          // goes on to level 1 (kernel 2)

        // ---- level-0 result (triangulation.py:588-604); kernel 2 overwrites it for `need` units
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
            if (P2S_DEBUG_MODE(a) == 4 && tid == 0) {   // diagnostics: phase stamps of wave 0 instead of unit 0/1 results
                st5 = __builtin_amdgcn_s_memtime();
                Qo[0] = (double)(st1 - st0); Qo[1] = (double)(st2 - st1); Qo[2] = (double)(st3 - st2);
                Qo[3] = (double)(st4 - st3); Qo[4] = (double)(st5 - st4); Qo[5] = (double)(st5 - st0);
            }
        }

        // ---- work list: one atomic per wave, records written by their lanes --------------------
        const unsigned long long hard = __ballot(need);
        if (hard != 0ull) {
            uint32_t base_rec = 0;
            const uint32_t shard = blockIdx.x % P2S_WL_SHARDS;
            if (lane == 0) base_rec = atomicAdd(a.wl_count + shard, (uint32_t)__popcll(hard));
            base_rec = __shfl(base_rec, 0, 64);
            if (need) {
                const uint32_t slot = base_rec + (uint32_t)__popcll(hard & ((1ull << lane) - 1ull));
                unsigned char *rec = a.wl_rec + ((size_t)shard * a.wl_capacity + slot) * a.rec_bytes;
                reinterpret_cast<uint32_t *>(rec)[0] = (uint32_t)(tile0 * K + u - a.block0 * K);   // unit id within the chunk
                reinterpret_cast<uint32_t *>(rec)[1] = 0u;
                T *ro = reinterpret_cast<T *>(rec + P2S_REC_HDR);
                for (int c = 0; c < C; ++c) {
                    ro[c * 3 + 0] = obs.p[c * strideC + 0];
                    ro[c * 3 + 1] = obs.p[c * strideC + 1];
                    ro[c * 3 + 2] = obs.p[c * strideC + 2];
                }
                if (LRSWAP) {
                    T *rs = ro + C * 3;
                    for (int c = 0; c < C; ++c) {
                        rs[c * 3 + 0] = obs_sw.p[c * strideC + 0];
                        rs[c * 3 + 1] = obs_sw.p[c * strideC + 1];
                        rs[c * 3 + 2] = obs_sw.p[c * strideC + 2];
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel 1, direct form (C <= CT): one lane per unit, no LDS and no barrier.  Lane u loads its C
// observations straight from HBM into registers -- consecutive lanes are consecutive keypoints, so
// every load instruction of a wave reads runs of 12-byte triplets that are contiguous per (frame,
// person) block, and over the C loads the wave consumes every byte of the blocks it covers exactly
// once.  Without a staged tile the waves of a CU are independent: the ~2 us HBM latency of one wave
// hides under the arithmetic of the others instead of stalling a whole workgroup at a barrier.
template <typename T, bool UNDISTORT, bool LRSWAP, int CT>
__global__ void __launch_bounds__(256, (CT <= 8 ? 4 : 3)) p2s_tri_level0_direct_kernel(const P2sTriArgs a) {
    const int C = a.C, K = a.K;
    cam_cptr cams = (cam_cptr)a.cams;
    const int lane = threadIdx.x & 63;
    const int64_t n_units = a.n_blocks * K;
    const int64_t lu = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // unit within the chunk
    if (lu - lane >= n_units) return;                                      // whole wave past the chunk
    const bool active = lu < n_units;
    const int64_t u = active ? lu : 0;
    const uint32_t b32 = (uint32_t)u / (uint32_t)K;            // a chunk holds < 2^31 units: 32-bit division
    const int64_t b = (int64_t)b32;
    const int k = (int)((uint32_t)u - b32 * (uint32_t)K);
    const int64_t gb = a.block0 + b;
    const T *base = reinterpret_cast<const T *>(a.xyl) + (P2S_DEBUG_MODE(a) == 2 ? 0 : gb * (int64_t)C * K * 3);

    RegObs<T, CT> obs, obs_sw;
    obs.lik_thr = a.lik_thr;
    obs_sw.lik_thr = a.lik_thr;
    const int ks = LRSWAP ? a.swap_idx[k] : k;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (c < C) {
            const T *p = base + ((int64_t)c * K + k) * 3;
            obs.x[c] = __builtin_nontemporal_load(p);
            obs.y[c] = __builtin_nontemporal_load(p + 1);
            obs.w[c] = __builtin_nontemporal_load(p + 2);
            if (LRSWAP) {
                const T *ps = base + ((int64_t)c * K + ks) * 3;
                obs_sw.x[c] = ps[0]; obs_sw.y[c] = ps[1]; obs_sw.w[c] = ps[2];
            }
        } else {
            obs.x[c] = obs.y[c] = obs.w[c] = (T)0;
            obs_sw.x[c] = obs_sw.y[c] = obs_sw.w[c] = (T)0;
        }
    }
    if (UNDISTORT) {          // triangulation.py:808-813 (float32 in / out); the mirrored keypoint too
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            if (c < C) {
                double x = (double)obs.x[c], y = (double)obs.y[c];
                undistort_point(cams + c, x, y);
                obs.x[c] = (T)x; obs.y[c] = (T)y;
                if (LRSWAP) {
                    double xs = (double)obs_sw.x[c], ys = (double)obs_sw.y[c];
                    undistort_point(cams + c, xs, ys);
                    obs_sw.x[c] = (T)xs; obs_sw.y[c] = (T)ys;
                }
            }
        }
    }

    const double thr = a.thr;
    const uint32_t allmask = (C == 32) ? 0xffffffffu : ((1u << C) - 1u);
    const int64_t gu = gb * K + k;

    if (P2S_DEBUG_MODE(a) == 1) {   // diagnostics: memory skeleton only
        if (active) {
            double s = 0;
#pragma unroll
            for (int c = 0; c < CT; ++c) s += (double)obs.x[c] + (double)obs.y[c] + (double)obs.w[c];
            a.Q[gu * 3] = s; a.Q[gu * 3 + 1] = s; a.Q[gu * 3 + 2] = s;
            a.err[gu] = (float)s; a.n_excl[gu] = 0; a.mask[gu] = 0;
        }
        return;
    }

    double N[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) N[i] = 0.0;
    uint32_t nanmask = 0, zeromask = 0;
    classify_and_accumulate<T, CT>(cams, C, obs, N, nanmask, zeromask);
    const uint32_t dmask = nanmask | zeromask;                 // cameras already out (NaN or zero likelihood)
    const uint32_t valid = allmask & ~dmask;
    const int V = __popc(dmask);
    const int nvalid = C - V;
    const int Lmax = active ? C - a.min_cams - V : -1;         // last level that runs (triangulation.py:408, 437-441)

    double err_min = kInf;
    double Qb[3] = {d_nan(), d_nan(), d_nan()};
    int n_excl = C;                                            // :595-596 when no level completes
    uint32_t mask = allmask;
    {
        double q[3];
        smallest_eigvec(N, q);
        if (nvalid < 2) { q[0] = d_nan(); q[1] = d_nan(); q[2] = d_nan(); }   // common.py:347: fewer than 4 rows
        const double e = mean_error<T, UNDISTORT, CT>(cams, C, obs, valid, q);
        if (Lmax >= 0) {
            err_min = e; Qb[0] = q[0]; Qb[1] = q[1]; Qb[2] = q[2];
            n_excl = V; mask = nanmask;
        }
    }
    if (LRSWAP) {                                              // :509-579 at level 0: M = nvalid
        const bool want = (Lmax >= 0) && (err_min > thr) && (nvalid > 2);
        if (__any(want)) {
            double qs[3];
            const double es = swap_candidate<T, UNDISTORT, CT>(cams, C, obs, obs_sw, valid, nvalid, qs);
            if (want && es < err_min) { err_min = es; Qb[0] = qs[0]; Qb[1] = qs[1]; Qb[2] = qs[2]; }
        }
    }
    const bool need = (Lmax >= 1) && (err_min > thr);          // goes on to level 1 (kernel 2)

    {   // triangulation.py:588-604.  The three doubles of a unit sit 24 bytes apart: transposing the
        // wave's 64 x 3 block through LDS turns three 8-byte-strided stores into three stores of 512
        // contiguous bytes (sub-line stores inflated the HBM write traffic by ~1.5x).
        __shared__ __align__(16) double sQ[4][192];
        const int wv = threadIdx.x >> 6;
        const bool fail = !(err_min <= thr);
        sQ[wv][lane * 3 + 0] = fail ? d_nan() : Qb[0];
        sQ[wv][lane * 3 + 1] = fail ? d_nan() : Qb[1];
        sQ[wv][lane * 3 + 2] = fail ? d_nan() : Qb[2];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int64_t wave_u0 = a.block0 * K + (lu - lane);        // first unit of this wave (global)
        const int64_t n_left = n_units - (lu - lane);              // units this wave owns
        double *Qw = a.Q + wave_u0 * 3;
        if (n_left >= 64 && ((reinterpret_cast<uintptr_t>(Qw) & 15) == 0)) {
            // full wave: 1536 contiguous bytes as 16-byte stores (64 lanes, then the first 32)
            typedef double v2d __attribute__((ext_vector_type(2)));
            const v2d *src = reinterpret_cast<const v2d *>(&sQ[wv][0]);
            v2d *dst = reinterpret_cast<v2d *>(Qw);
            dst[lane] = src[lane];
            if (lane < 32) dst[64 + lane] = src[64 + lane];
        } else {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int idx = r * 64 + lane;
                if (idx < 3 * n_left) Qw[idx] = sQ[wv][idx];
            }
        }
        // the 4-byte and 1-byte outputs the same way: 16 bytes per storing lane
        __shared__ __align__(16) uint32_t sE[4][64], sM[4][64];
        __shared__ __align__(16) uint8_t sX[4][64];
        sE[wv][lane] = __float_as_uint(fail ? __builtin_nanf("") : (float)err_min);
        sM[wv][lane] = mask;
        sX[wv][lane] = (uint8_t)n_excl;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float *Ew = a.err + wave_u0;
        uint32_t *Mw = a.mask + wave_u0;
        uint8_t *Xw = a.n_excl + wave_u0;
        const bool al16 = ((reinterpret_cast<uintptr_t>(Ew) | reinterpret_cast<uintptr_t>(Mw) |
                            reinterpret_cast<uintptr_t>(Xw)) & 15) == 0;
        if (n_left >= 64 && al16) {
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            if (lane < 16) reinterpret_cast<v4u *>(Ew)[lane] = reinterpret_cast<const v4u *>(&sE[wv][0])[lane];
            else if (lane < 32) reinterpret_cast<v4u *>(Mw)[lane - 16] = reinterpret_cast<const v4u *>(&sM[wv][0])[lane - 16];
            else if (lane < 36) reinterpret_cast<v4u *>(Xw)[lane - 32] = reinterpret_cast<const v4u *>(&sX[wv][0])[lane - 32];
        } else if (active) {
            a.err[gu] = fail ? __builtin_nanf("") : (float)err_min;
            a.n_excl[gu] = (uint8_t)n_excl;
            a.mask[gu] = mask;
        }
    }

    const unsigned long long hard = __ballot(need);
    if (hard != 0ull) {
        uint32_t base_rec = 0;
        const uint32_t shard = blockIdx.x % P2S_WL_SHARDS;
        if (lane == 0) base_rec = atomicAdd(a.wl_count + shard, (uint32_t)__popcll(hard));
        base_rec = __shfl(base_rec, 0, 64);
        if (need) {
            const uint32_t slot = base_rec + (uint32_t)__popcll(hard & ((1ull << lane) - 1ull));
            unsigned char *rec = a.wl_rec + ((size_t)shard * a.wl_capacity + slot) * a.rec_bytes;
            // 16-byte stores: the record is a 16-byte header + the observations, padded to 16 bytes
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            v4u *r16 = reinterpret_cast<v4u *>(rec);
            r16[0] = v4u{(uint32_t)lu, 0u, 0u, 0u};               // unit id within the chunk
            constexpr int PER = 16 / sizeof(T);                    // values per 16-byte store
            T flat[2 * CT * 3 + PER];
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                flat[c * 3 + 0] = obs.x[c]; flat[c * 3 + 1] = obs.y[c]; flat[c * 3 + 2] = obs.w[c];
            }
            // C < CT: the swapped block starts right after the C real cameras
#pragma unroll
            for (int i = 0; i < (CT * 3 + PER - 1) / PER; ++i) {
                if (i * PER < C * 3) {
                    v4u v;
                    __builtin_memcpy(&v, &flat[i * PER], 16);
                    r16[1 + i] = v;
                }
            }
            if (LRSWAP) {
                T *rs = reinterpret_cast<T *>(rec + P2S_REC_HDR) + C * 3;
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    if (c < C) { rs[c * 3 + 0] = obs_sw.x[c]; rs[c * 3 + 1] = obs_sw.y[c]; rs[c * 3 + 2] = obs_sw.w[c]; }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel 2 -- camera-subset search over the work list.  One wave owns 64 records at a time
// (lane i = record i); LDS: [P: C*12 doubles][binom: 33*33 u32][per wave: 64 records].
// Persistent grid: waves pull jobs from the sharded list whose lengths kernel 1 left in wl_count.
template <typename T, bool UNDISTORT, bool LRSWAP>
__global__ void __launch_bounds__(256, 3) p2s_tri_search_kernel(const P2sTriArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int C = a.C;
    double *sP = reinterpret_cast<double *>(smem);
    uint32_t *sBinom = reinterpret_cast<uint32_t *>(smem + a.lds_binom_off);
    cam_cptr cams = (cam_cptr)a.cams;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int waves_per_block = blockDim.x >> 6;
    // per wave: a.job records, then a.job owner states {N[10] f64, nanmask u32, zeromask u32, pad} of 96 bytes
    unsigned char *recs = smem + a.lds_rec_off + (size_t)wave * a.job * (a.rec_bytes + 96);
    unsigned char *states = recs + (size_t)a.job * a.rec_bytes;

    for (int i = tid; i < C * 12; i += blockDim.x) sP[i] = a.cams[i / 12].P[i % 12];
    for (int i = tid; i < 33 * 33; i += blockDim.x) sBinom[i] = a.binom[i];
    __syncthreads();

    // Wave w serves shard w % SHARDS (shards fill evenly: every SHARDS-th tile); the waves of a shard
    // pull jobs of a.job records through the shard's atomic ticket until it is drained.
    const uint32_t gwave = blockIdx.x * waves_per_block + wave;
    const double thr = a.thr;
    const uint32_t allmask = (C == 32) ? 0xffffffffu : ((1u << C) - 1u);
    const uint32_t shard = gwave % P2S_WL_SHARDS;
    uint32_t *ticket = a.wl_count + P2S_WL_SHARDS + shard;
    const uint32_t count = min(a.wl_count[shard], a.wl_capacity);

    // diagnostics (debug_mode 5, exp/k2_trace.py): per-wave timeline instead of results.  What it showed on
    // cfg2: all waves start within 1 us, the first ends at 0.65 and the median at 0.78 of the kernel (the
    // oldest wave of a SIMD has issue priority, so a shard's last jobs run on its slowest waves); guided job
    // sizes or stealing across shards need a device-scope look at the tickets first, which costs more
    // (~4 us per job) than the tail they remove.
    const bool trace = P2S_DEBUG_MODE(a) == 5;
    uint64_t t_begin = 0, t_fetch = 0, t_n = 0, t_lvl1 = 0, t_search = 0;
    uint32_t n_jobs = 0;
    if (trace) t_begin = __builtin_amdgcn_s_memrealtime();
    uint32_t st_units = 0, st_evals = 0, st_passes = 0, st_capped = 0;     // p2s_get_tri_stats (wave-uniform counts)
    unsigned long long st_pcams = 0, st_psubs = 0;                          // candidates of the pruned passes, and their camera errors
    for (;;) {
        uint64_t t0 = 0;
        if (trace) t0 = __builtin_amdgcn_s_memtime();
        uint32_t job = 0;
        if (lane == 0) job = atomicAdd(ticket, 1u);
        job = __shfl(job, 0, 64);
        const uint32_t rec0 = job * (uint32_t)a.job;
        if (rec0 >= count) break;                                  // shard drained
        const int n = (int)min((uint32_t)a.job, count - rec0);
        // ---- this wave's records: contiguous copy into its LDS region ----------------------
        {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(a.wl_rec + ((size_t)shard * a.wl_capacity + rec0) * a.rec_bytes);
            uint32_t *dst = reinterpret_cast<uint32_t *>(recs);
            const int nw = n * (a.rec_bytes >> 2);
            for (int i = lane; i < nw; i += 64) dst[i] = src[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        if (trace) { __builtin_amdgcn_sched_barrier(0); t_fetch += __builtin_amdgcn_s_memtime() - t0; ++n_jobs; t0 = __builtin_amdgcn_s_memtime(); }
        const bool active = lane < n;
        const unsigned char *myrec = recs + (size_t)(active ? lane : 0) * a.rec_bytes;
        const uint32_t u = *reinterpret_cast<const uint32_t *>(myrec);          // unit id within the chunk
        UnitObs<T> obs{reinterpret_cast<const T *>(myrec + P2S_REC_HDR), 3, a.lik_thr};

        // level-0 state again (cheaper than carrying 80 bytes per unit through HBM)
        double N[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) N[i] = 0.0;
        uint32_t nanmask = 0, zeromask = 0;
        classify_and_accumulate<T, 0>(cams, C, obs, N, nanmask, zeromask);
        const int V = __popc(nanmask | zeromask);
        const int Lmax = active ? C - a.min_cams - V : -1;
        if (lane < a.job) {   // publish the owner state: the lanes of the group that works on this unit read it from LDS
            double *sN = reinterpret_cast<double *>(states + (size_t)lane * 96);
#pragma unroll
            for (int i = 0; i < 10; ++i) sN[i] = N[i];
            reinterpret_cast<uint32_t *>(sN + 10)[0] = nanmask;
            reinterpret_cast<uint32_t *>(sN + 10)[1] = zeromask;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        double err_min = kInf;
        double Qb[3] = {d_nan(), d_nan(), d_nan()};
        int n_excl = C;
        uint32_t mask = allmask;
        if (trace) { __builtin_amdgcn_sched_barrier(0); t_n += __builtin_amdgcn_s_memtime() - t0; t0 = __builtin_amdgcn_s_memtime(); }

        // ---- subset search, levels in lock step across the wave -----------------------------
        unsigned long long pend_level = __ballot(Lmax >= 1);
        for (int level = 1; pend_level != 0ull; ++level) {
            if (trace && level == 2) { __builtin_amdgcn_sched_barrier(0); t_lvl1 += __builtin_amdgcn_s_memtime() - t0; }
            unsigned long long pending = pend_level;
            bool cont = false;                                      // owner lanes: continue to level+1
            bool capped = false;
            const uint32_t nsub = sBinom[C * 33 + level];
            // lanes per unit for this level: the power of two that needs the fewest
            // (passes over the pending units) x (rounds over the level's subsets)
            int lg = 2;
            {
                const uint32_t npend = (uint32_t)__popcll(pend_level);
                uint32_t best_cost = 0xffffffffu;
                for (int l = 2; l <= 6; ++l) {
                    const uint32_t passes = ((npend << l) + 63u) >> 6;
                    const uint32_t rounds = (nsub + (1u << l) - 1u) >> l;
                    const uint32_t cost = passes * rounds;
                    if (cost <= best_cost) { best_cost = cost; lg = l; }
                }
            }
            const int G = 1 << lg, groups = 64 >> lg;
            const int grp = lane >> lg, lig = lane & (G - 1);

            while (pending != 0ull) {
                const unsigned long long before = pending;
                for (int i = 0; i < groups && pending; ++i) pending &= pending - 1;
                const unsigned long long batch = before & ~pending;   // owners served in this pass
                const int owner = nth_set_bit(batch, grp);            // record this group works on (lane id) or -1

                // the owner's state
                const int src = owner < 0 ? 0 : owner;
                const double *oN = reinterpret_cast<const double *>(states + (size_t)src * 96);
                const uint32_t o_nan = reinterpret_cast<const uint32_t *>(oN + 10)[0];
                const uint32_t o_zero = reinterpret_cast<const uint32_t *>(oN + 10)[1];
                const uint32_t o_d = o_nan | o_zero, o_valid = allmask & ~o_d;
                const int oV = __popc(o_d);
                const unsigned char *orec = recs + (size_t)src * a.rec_bytes;
                UnitObs<T> oobs{reinterpret_cast<const T *>(orec + P2S_REC_HDR), 3, a.lik_thr};
                UnitObs<T> oobs_sw = oobs;
                if (LRSWAP) oobs_sw.p = oobs.p + C * 3;
                const int M = C - oV - level;                       // cameras left when `level` valid ones go (:437, 513)

                // best candidates seen by this lane (plain / swap), lowest rank first
                double be = kInf, bq0 = d_nan(), bq1 = d_nan(), bq2 = d_nan();
                uint32_t brank = 0xffffffffu, bS = 0;
                double se = kInf, sq0 = d_nan(), sq1 = d_nan(), sq2 = d_nan();
                uint32_t srank = 0xffffffffu, sS = 0;

                // Exact pruning where the whole wave works on ONE unit (64 lanes per unit: the levels of hundreds of
                // subsets and more), as in the deep rounds (p2s_tri_deep.hip): cameras ranked by their level-0 residual,
                // the level's subsets enumerated over the ranked cameras, the error loop over the ranked cameras until
                // every lane of the wave is out; a wave with a survivor re-evaluates in camera order; ties by the rank
                // in itertools order.
                __shared__ uint8_t sPermAll[4][32];
                uint8_t *sPerm = sPermAll[(threadIdx.x >> 6) & 3];
                const bool pr64 = a.prune && G == 64 && owner >= 0;
                // no further level will replace this one's result: the unit's last level, or the next one is behind the valve
                const bool last_level = level >= __shfl(Lmax, src, 64) || sBinom[C * 33 + level + 1] > a.max_subsets;
                if (pr64) {
                    double Nf[10], q0[3];
#pragma unroll
                    for (int i = 0; i < 10; ++i) Nf[i] = oN[i];
                    smallest_eigvec(Nf, q0);
                    double res = -1.0;                              // cameras that are out already go last
                    if (lane < C && ((o_valid >> lane) & 1u)) {
                        double x, y, w;
                        oobs.raw(lane, x, y, w);
                        bool reg;
                        const double dd = camera_distance<UNDISTORT>(cams + lane, q0, x, y, reg);
                        res = (dd == dd && dd >= 0.0) ? dd : -1.0;
                    }
                    int pos = 0;
                    for (int c2 = 0; c2 < C; ++c2) {
                        const double r2 = shfl_d(res, c2);
                        pos += (r2 > res || (r2 == res && c2 < lane)) ? 1 : 0;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (lane < C) sPerm[pos] = (uint8_t)lane;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }

                for (uint32_t r0 = 0; r0 < nsub; r0 += G) {
                    const uint32_t r = r0 + lig;
                    bool go = (owner >= 0) && (r < nsub);
                    uint32_t S = 0;
                    if (go) {
                        S = unrank_subset(r, C, level, sBinom);
                        if (pr64) {                                 // r counts combinations of ranked positions
                            uint32_t Sm = 0;
                            for (uint32_t b = S; b != 0u; b &= b - 1) Sm |= 1u << sPerm[__builtin_ctz(b)];
                            S = Sm;
                        }
                        // duplicates of one effective configuration (quirk Q1) carry identical numbers;
                        // only the lexicographically first one -- padding = lowest cameras of D -- can win
                        const uint32_t pad = S & o_d;
                        const int np = __popc(pad);
                        uint32_t low = 0, dd = o_d;
                        for (int i = 0; i < np; ++i) { low |= dd & (0u - dd); dd &= dd - 1; }
                        go = (pad == low);
                    }
                    if (!__any(go)) continue;
                    st_evals += (uint32_t)__popcll(__ballot(go)); ++st_passes;
                    const uint32_t Rreal = S & o_valid;
                    const uint32_t kept = o_valid & ~Rreal;
                    const int nkept = __popc(kept);
                    double Ns[10];
#pragma unroll
                    for (int i = 0; i < 10; ++i) Ns[i] = oN[i];
                    for (uint32_t rr = go ? Rreal : 0u; __any(rr != 0u); rr &= rr - 1) {
                        const int c = rr ? __builtin_ctz(rr) : 0;
                        double x, y, w;
                        oobs.raw(c, x, y, w);
                        const bool on = rr != 0u;
                        accum_camera<-1>(Ns, sP + c * 12, on ? x : 0.0, on ? y : 0.0, on ? w : 0.0);
                    }
                    double q[3];
                    smallest_eigvec(Ns, q);
                    if (nkept < 2) { q[0] = d_nan(); q[1] = d_nan(); q[2] = d_nan(); }
                    if (pr64) {
                        const double bw = wave_min_d(be);           // any finished candidate bounds the level's minimum
                        // with swap candidates in play the plain minimum of a FAILED level still reports its camera count
                        // when a swap candidate rescues the level (:576-579): only the best-so-far bounds then
                        const double bmean = (last_level || (LRSWAP && M > 2)) ? bw : fmin(bw, thr);
                        const double bnd = bmean * (double)nkept * (1.0 + 1e-9);
                        double psum = 0.0;
                        int idx = 0;
                        for (; idx < C; ++idx) {
                            const int c = __builtin_amdgcn_readfirstlane((int)sPerm[idx]);
                            double x, y, w;
                            oobs.raw(c, x, y, w);
                            bool reg;
                            const double dd = camera_distance<UNDISTORT>(cams + c, q, x, y, reg);
                            psum += (((kept >> c) & 1u) && reg && dd == dd) ? dd : 0.0;
                            if ((idx & 3) == 3 && __all(!go || psum > bnd)) { ++idx; break; }
                        }
                        const unsigned long long ngo = (unsigned long long)__popcll(__ballot(go));
                        st_pcams += (unsigned long long)idx * ngo; st_psubs += ngo;
                        const bool alive = go && !(psum > bnd);
                        double e = kInf;
                        if (__any(alive)) { e = mean_error<T, UNDISTORT, 0>(cams, C, oobs, kept, q); st_pcams += (unsigned long long)C * ngo; }
                        const uint32_t rt = rank_subset(S, C, level, sBinom);
                        if (alive && better_candidate(e, rt, be, brank)) { be = e; bq0 = q[0]; bq1 = q[1]; bq2 = q[2]; brank = rt; bS = S; }
                    } else {
                        const double e = mean_error<T, UNDISTORT, 0>(cams, C, oobs, kept, q);
                        if (go && (brank == 0xffffffffu || e < be || (be != be && e == e))) { be = e; bq0 = q[0]; bq1 = q[1]; bq2 = q[2]; brank = r; bS = S; }
                    }
                }

                // group argmin, first (lowest-rank) index on ties (np.nanargmin, :502; np.argmin, :568)
                for (int off = G >> 1; off > 0; off >>= 1) {
                    const double oe = shfl_d(be, lane ^ off);
                    const uint32_t orank = __shfl(brank, lane ^ off, 64);
                    const bool take = better_candidate(oe, orank, be, brank);
                    const double t0 = shfl_d(bq0, lane ^ off), t1 = shfl_d(bq1, lane ^ off), t2 = shfl_d(bq2, lane ^ off);
                    const uint32_t tS = __shfl(bS, lane ^ off, 64);
                    if (take) { be = oe; brank = orank; bq0 = t0; bq1 = t1; bq2 = t2; bS = tS; }
                }
                // L/R-swap candidates (:509-579) only where the reference evaluates them: when the level's plain
                // minimum is still above the threshold.  The last level of a unit -- the largest one -- usually
                // succeeds without them, so this second pass over the subsets is mostly skipped.
                if (LRSWAP) {
                    const bool need_sw = (owner >= 0) && (M > 2) && (be > thr);
                    if (__any(need_sw)) {
                        double Nsw[10];
                        uint32_t nan_sw = 0;
                        swap_base<T>(cams, C, oobs, oobs_sw, o_valid, Nsw, nan_sw);   // all valid cameras mirrored, once per pass
                        for (uint32_t r0 = 0; r0 < nsub; r0 += G) {
                            const uint32_t r = r0 + lig;
                            bool go = need_sw && (r < nsub);
                            uint32_t S = 0;
                            if (go) {
                                S = unrank_subset(r, C, level, sBinom);
                                if (pr64) {                         // over the ranked cameras, as the plain candidates
                                    uint32_t Sm = 0;
                                    for (uint32_t b = S; b != 0u; b &= b - 1) Sm |= 1u << sPerm[__builtin_ctz(b)];
                                    S = Sm;
                                }
                                const uint32_t pad = S & o_d;
                                const int np = __popc(pad);
                                uint32_t low = 0, dd = o_d;
                                for (int i = 0; i < np; ++i) { low |= dd & (0u - dd); dd &= dd - 1; }
                                go = (pad == low);
                            }
                            if (!__any(go)) continue;
                            const uint32_t kept = o_valid & ~(S & o_valid);
                            double qs[3];
                            swap_solve_from_base<T>(Nsw, nan_sw, sP, oobs, oobs_sw, o_valid, kept, M, go, qs);
                            if (pr64) {
                                const double sw = wave_min_d(se);
                                const double es = swap_error_pruned<T, UNDISTORT>(cams, C, oobs_sw, kept, M, qs, sPerm, go, last_level ? sw : fmin(sw, thr));
                                const uint32_t rt = rank_subset(S, C, level, sBinom);
                                if (go && (srank == 0xffffffffu || es < se || (es == se && rt < srank))) { se = es; sq0 = qs[0]; sq1 = qs[1]; sq2 = qs[2]; srank = rt; sS = S; }
                            } else {
                                const double es = swap_error<T, UNDISTORT, 0>(cams, C, oobs_sw, kept, M, qs);
                                if (go && (es < se || srank == 0xffffffffu)) { se = es; sq0 = qs[0]; sq1 = qs[1]; sq2 = qs[2]; srank = r; sS = S; }
                            }
                        }
                        for (int off = G >> 1; off > 0; off >>= 1) {
                            const double xe = shfl_d(se, lane ^ off);
                            const uint32_t xrank = __shfl(srank, lane ^ off, 64);
                            const bool tk = (xrank != 0xffffffffu) && (srank == 0xffffffffu || xe < se || (xe == se && xrank < srank));
                            const double s0 = shfl_d(sq0, lane ^ off), s1 = shfl_d(sq1, lane ^ off), s2 = shfl_d(sq2, lane ^ off);
                            const uint32_t xS = __shfl(sS, lane ^ off, 64);
                            if (tk) { se = xe; srank = xrank; sq0 = s0; sq1 = s1; sq2 = s2; sS = xS; }
                        }
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
                    // safety valve: a level with more than 2^26 subsets (C(32, 11) and beyond; C(32, 10) = 64.5 M still
                    // runs) is not entered -- the reference would need hours of CPU for that one keypoint; the unit
                    // ends as "not triangulated" and is counted (p2s_get_tri_stats)
                    const bool more = (err_min > thr) && (level + 1 <= Lmax) && (P2S_DEBUG_MODE(a) != 3);
                    const uint32_t nsub_next = more ? sBinom[C * 33 + level + 1] : 0u;
                    cont = more && (nsub_next <= a.max_subsets);
                    capped = capped || (more && !cont);
                    if (cont && a.deep_entries && nsub_next > a.deep_min_subsets) {
                        // the next level is too long for one wave: hand the unit to the deep rounds (p2s_tri_deep.hip)
                        const uint32_t slot = atomicAdd(a.deep_ctl + P2S_DEEP_N_ENTRIES, 1u);
                        if (slot < a.deep_capacity) {
                            unsigned char *dst = a.deep_entries + (size_t)slot * a.deep_entry_bytes;
                            P2sDeepEntry *de = reinterpret_cast<P2sDeepEntry *>(dst);
                            de->unit = u; de->level = (uint32_t)(level + 1); de->Lmax = Lmax;
                            de->nanmask = nanmask; de->zeromask = zeromask; de->mask = mask; de->n_excl = (uint32_t)n_excl;
                            de->state = P2S_DEEP_WAITING; de->first_ticket = 0; de->n_chunks = 0; de->pad0 = de->pad1 = 0;
                            de->err_min = err_min; de->Q[0] = Qb[0]; de->Q[1] = Qb[1]; de->Q[2] = Qb[2];
                            const double *sN = reinterpret_cast<const double *>(states + (size_t)lane * 96);
                            for (int i = 0; i < 10; ++i) de->N[i] = sN[i];
                            const uint32_t *src = reinterpret_cast<const uint32_t *>(myrec + P2S_REC_HDR);
                            uint32_t *od = reinterpret_cast<uint32_t *>(dst + sizeof(P2sDeepEntry));
                            for (int i = 0; i < ((a.rec_bytes - P2S_REC_HDR) >> 2); ++i) od[i] = src[i];
                            cont = false;
                        }
                    }
                }
            }
            pend_level = __ballot(cont);
            st_capped += (uint32_t)__popcll(__ballot(capped));
        }
        st_units += (uint32_t)n;

        if (trace) { __builtin_amdgcn_sched_barrier(0); t_search += __builtin_amdgcn_s_memtime() - t0; }
        // ---- finalise (triangulation.py:588-604) ------------------------------------------
        if (active && !trace) {
            const int64_t gu = a.block0 * a.K + u;
            const bool fail = !(err_min <= thr);
            double *Qo = a.Q + gu * 3;
            Qo[0] = fail ? d_nan() : Qb[0];
            Qo[1] = fail ? d_nan() : Qb[1];
            Qo[2] = fail ? d_nan() : Qb[2];
            a.err[gu] = fail ? __builtin_nanf("") : (float)err_min;
            a.n_excl[gu] = (uint8_t)n_excl;
            a.mask[gu] = mask;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the next records overwrite this wave's LDS region
        __builtin_amdgcn_wave_barrier();
    }
    if (a.stats && lane == 0 && st_units) {
        unsigned long long *st = a.stats + (size_t)(blockIdx.x % P2S_STAT_SHARDS) * P2S_STAT_STRIDE;
        atomicAdd(st + 0, (unsigned long long)st_units);
        atomicAdd(st + 1, (unsigned long long)st_evals);
        atomicAdd(st + 2, (unsigned long long)st_passes);
        if (st_capped) atomicAdd(st + 3, (unsigned long long)st_capped);
        if (st_psubs) { atomicAdd(st + 4, st_pcams); atomicAdd(st + 5, st_psubs); }
    }
    if (trace && lane == 0) {
        double *o = a.Q + (size_t)gwave * 8;
        o[0] = (double)t_begin; o[1] = (double)__builtin_amdgcn_s_memrealtime(); o[2] = (double)n_jobs;
        o[3] = (double)t_fetch; o[4] = (double)t_n; o[5] = (double)t_lvl1; o[6] = (double)t_search; o[7] = 0.0;
    }
}

// ---------------------------------------------------------------------------------------------
// Single-person association (personAssociation.py:67-257, pinhole branch): one wave per frame.
// Lane i of a pass takes combination #(pass*64 + i) of "one person per camera" (itertools.product
// order, last camera fastest), switches off the cameras whose tracked-keypoint likelihood is below the
// threshold (:212-213), and evaluates every subset of `extra` further cameras switched off
// (itertools.combinations over the active cameras, :219-231) with the weighted DLT + reprojection
// error of kernel 1.  The reference's sequential rules are then applied across the lanes in
// combination order: the scan of a level stops at the first combination below the threshold
// (:242-243), `error_min` is that of the LAST combination looked at (:233) and drives the while loop
// (:192), the best solution is kept across levels with a strict '<' (:237).
// LDS: [tracked keypoints: C x P2S_MAX_PERSONS_PER_CAM x 3 doubles][ids: 64 x C bytes][binom]
template <typename T>
__global__ void __launch_bounds__(64) p2s_single_kernel(const P2sSingleArgs a) {
    constexpr int PMAX = 16;
    extern __shared__ __align__(16) unsigned char smem[];
    const int C = a.C;
    double *tk = reinterpret_cast<double *>(smem);                                  // [C][PMAX][3]
    uint8_t *ids = smem + (size_t)C * PMAX * 3 * sizeof(double);                    // [64][C]
    uint32_t *sBinom = reinterpret_cast<uint32_t *>(smem + (((size_t)C * PMAX * 24 + 64 * C + 15) / 16) * 16);
    cam_cptr cams = (cam_cptr)a.cams;
    const int lane = threadIdx.x;
    const int64_t f = blockIdx.x;
    const double thr = a.thr;

    for (int i = lane; i < 33 * 33; i += 64) sBinom[i] = a.binom[i];
    // persons per camera and their tracked keypoints
    const T *src = reinterpret_cast<const T *>(a.tracked) + a.offsets[f] * 3;
    int row = 0;
    uint32_t present = 0;
    for (int c = 0; c < C; ++c) {
        const int pc = min(a.n_persons[f * C + c], PMAX);
        if (pc > 0) present |= 1u << c;
        for (int i = lane; i < pc * 3; i += 64) tk[(c * PMAX) * 3 + i] = (double)src[row * 3 + i];
        row += pc;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    uint32_t n_comb = 1;
    for (int c = 0; c < C; ++c) n_comb *= (uint32_t)max(1, min(a.n_persons[f * C + c], PMAX));
    const int missing = C - __popc(present);

    double error_min = kInf;
    double best_e = kInf, best_q0 = d_nan(), best_q1 = d_nan(), best_q2 = d_nan();
    uint32_t best_ci = 0xffffffffu, best_kept = 0;
    uint8_t *my_ids = ids + (size_t)lane * C;

    for (int extra = 0; error_min > thr && C - (missing + extra) >= a.min_cams; ++extra) {
        bool stop = false;
        for (uint32_t c0 = 0; c0 < n_comb && !stop; c0 += 64) {
            const uint32_t ci = c0 + lane;
            const bool in_range = ci < n_comb;
            // decode the combination (itertools.product: the last camera varies fastest)
            uint32_t rem = in_range ? ci : 0u, active = 0;
            for (int c = C - 1; c >= 0; --c) {
                const uint32_t rc = (uint32_t)max(1, min(a.n_persons[f * C + c], PMAX));
                const uint32_t id = rem % rc;
                rem /= rc;
                my_ids[c] = (uint8_t)id;
                if ((present >> c) & 1u) {
                    const double lk = tk[(c * PMAX + id) * 3 + 2];
                    if (!(lk < a.lik_thr) && lk != 0.0) active |= 1u << c;        // :212-213
                }
            }
            const int nact = __popc(active);
            bool valid = in_range && nact >= a.min_cams && extra < nact;          // :216-217, :234 (all NaN)
            // every subset of `extra` active cameras switched off
            double ce = kInf, cq0 = d_nan(), cq1 = d_nan(), cq2 = d_nan();
            uint32_t ckept = 0;
            bool first = true;
            const uint32_t nsub = valid ? sBinom[nact * 33 + extra] : 0u;
            uint32_t max_sub = nsub;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) max_sub = max(max_sub, (uint32_t)__shfl_xor((int)max_sub, off, 64));
            for (uint32_t r = 0; r < max_sub; ++r) {
                const bool go = valid && r < nsub;
                uint32_t kept = active;
                if (go) {
                    const uint32_t S = unrank_subset(r, nact, extra, sBinom);     // positions among the active cameras
                    uint32_t act = active;
                    for (int pos = 0; act; ++pos, act &= act - 1)
                        if ((S >> pos) & 1u) kept &= ~(act & (0u - act));
                }
                double N[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) N[i] = 0.0;
                for (int c = 0; c < C; ++c) {
                    const double *o = tk + ((size_t)c * PMAX + my_ids[c]) * 3;
                    const bool k = go && ((kept >> c) & 1u);
                    accum_camera<1>(N, cams[c].P, k ? o[0] : 0.0, k ? o[1] : 0.0, k ? o[2] : 0.0);
                }
                double q[3];
                smallest_eigvec(N, q);
                if (__popc(kept) < 2) { q[0] = d_nan(); q[1] = d_nan(); q[2] = d_nan(); }
                // reprojection error as in kernel 1: one reciprocal square root per camera, the literal formula with
                // the reference's NaN rules only when some wanted camera is degenerate
                double sum = 0.0;
                bool irregular = false;
                for (int c = 0; c < C; ++c) {
                    const double *o = tk + ((size_t)c * PMAX + my_ids[c]) * 3;
                    const bool k = go && ((kept >> c) & 1u);
                    bool reg;
                    const double d = camera_distance<false>(cams + c, q, o[0], o[1], reg);
                    irregular = irregular || (k && !reg);
                    sum += k ? d : 0.0;
                }
                if (__any(irregular)) {
                    double sum2 = 0.0;
                    for (int c = 0; c < C; ++c) {
                        const double *o = tk + ((size_t)c * PMAX + my_ids[c]) * 3;
                        const double d = camera_distance_exact(cams + c, q[0], q[1], q[2], o[0], o[1]);
                        sum2 += (go && ((kept >> c) & 1u)) ? d : 0.0;
                    }
                    sum = irregular ? sum2 : sum;
                }
                // One camera left: weighted_triangulation (common.py:327-356) returns NaN for fewer than 4 rows
                // and euclidean_distance turns the all-NaN difference into inf.
                const double e = __popc(kept) < 2 ? kInf : sum * p2s_inline_rcp((double)__popc(kept));
                if (go && (first || e < ce)) { ce = e; cq0 = q[0]; cq1 = q[1]; cq2 = q[2]; ckept = kept; first = false; }
            }
            // the reference's sequential scan over the combinations of this pass, in order
            const unsigned long long vmask = __ballot(valid);
            const unsigned long long below = __ballot(valid && ce < thr);
            const int brk = below ? __builtin_ctzll(below) : 64;
            const unsigned long long seen = (brk >= 63) ? vmask : (vmask & ((2ull << brk) - 1ull));
            if (seen != 0ull) {
                // lowest error among the combinations looked at, earliest one on ties
                double me = ((seen >> lane) & 1ull) ? ce : kInf;
                uint32_t ml = ((seen >> lane) & 1ull) ? (uint32_t)lane : 0xffffffffu;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const double oe = shfl_d(me, lane ^ off);
                    const uint32_t ol = __shfl(ml, lane ^ off, 64);
                    const bool take = (ol != 0xffffffffu) && (ml == 0xffffffffu || oe < me || (oe == me && ol < ml));
                    if (take) { me = oe; ml = ol; }
                }
                if (ml != 0xffffffffu && me < best_e) {                          // :237 strict
                    best_e = me;
                    best_q0 = shfl_d(cq0, (int)ml); best_q1 = shfl_d(cq1, (int)ml); best_q2 = shfl_d(cq2, (int)ml);
                    best_kept = __shfl(ckept, (int)ml, 64);
                    best_ci = c0 + ml;
                }
                const int last = 63 - __builtin_clzll(seen);
                error_min = shfl_d(ce, last);                                      // :233: the last one looked at
            }
            stop = below != 0ull;                                                  // :242-243
        }
    }

    // ---- result: chosen person per camera, -1 where the camera is off -------------------------------
    if (lane < C) {
        int id = -1;
        if (best_ci != 0xffffffffu) {
            uint32_t rem = best_ci;
            int mine = 0;
            for (int c = C - 1; c >= 0; --c) {
                const uint32_t rc = (uint32_t)max(1, min(a.n_persons[f * C + c], PMAX));
                if (c == lane) mine = (int)(rem % rc);
                rem /= rc;
            }
            id = ((best_kept >> lane) & 1u) ? mine : -1;
        }
        a.comb[f * C + lane] = id;
    }
    if (lane == 0) {
        a.err[f] = best_e;
        a.Q[f * 3 + 0] = best_q0; a.Q[f * 3 + 1] = best_q1; a.Q[f * 3 + 2] = best_q2;
    }
}

hipError_t p2s_launch_single(const P2sSingleArgs &a, int dtype, hipStream_t s) {
    const size_t lds = (((size_t)a.C * 16 * 24 + 64 * a.C + 15) / 16) * 16 + 33 * 33 * 4;
    if (dtype == 0)
        hipLaunchKernelGGL((p2s_single_kernel<float>), dim3((unsigned)a.n_frames), dim3(64), lds, s, a);
    else
        hipLaunchKernelGGL((p2s_single_kernel<double>), dim3((unsigned)a.n_frames), dim3(64), lds, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Kernel 1 goes to the main stream, kernel 2 to the side stream behind an event, so that the search
// of one chunk runs beside the streaming pass of the next.
template <typename T, bool U, bool L>
static hipError_t launch_level0(const P2sTriArgs &a, const P2sTriLaunch &g, hipStream_t s) {
    if (a.C <= 8 && !g.force_tiled) {        // register-resident observations, no LDS
        const int64_t n_units = a.n_blocks * a.K;
        const unsigned grid = (unsigned)((n_units + 255) / 256);
        hipLaunchKernelGGL((p2s_tri_level0_direct_kernel<T, U, L, 8>), dim3(grid), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    if constexpr (sizeof(T) == 4 && !L) {    // 16 float32 triplets still fit in registers at 3 waves per SIMD (not with the swapped copy)
        if (a.C <= 16 && !g.force_tiled) {
            const int64_t n_units = a.n_blocks * a.K;
            const unsigned grid = (unsigned)((n_units + 255) / 256);
            hipLaunchKernelGGL((p2s_tri_level0_direct_kernel<T, U, L, 16>), dim3(grid), dim3(256), 0, s, a);
            return hipGetLastError();
        }
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&p2s_tri_level0_kernel<T, U, L>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, g.lds0);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((p2s_tri_level0_kernel<T, U, L>), dim3(g.grid0), dim3(g.threads0), g.lds0, s, a);
    return hipGetLastError();
}

template <typename T, bool U, bool L>
static hipError_t launch_both(const P2sTriArgs &a, const P2sTriLaunch &g, hipStream_t s, hipStream_t side,
                              hipEvent_t k1_done) {
    hipError_t e = launch_level0<T, U, L>(a, g, s);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(&p2s_tri_search_kernel<T, U, L>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, g.lds1);
    if (e != hipSuccess) return e;
    if (P2S_DEBUG_MODE(a) == 1) return hipSuccess;
    if (side != s) {
        if ((e = hipEventRecord(k1_done, s)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(side, k1_done, 0)) != hipSuccess) return e;
    }
    hipLaunchKernelGGL((p2s_tri_search_kernel<T, U, L>), dim3(g.grid1), dim3(g.threads1), g.lds1, side, a);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_t(const P2sTriArgs &a, const P2sTriLaunch &g, hipStream_t s, hipStream_t side, hipEvent_t ev) {
    if (a.undistort)
        return a.lr_swap ? launch_both<T, true, true>(a, g, s, side, ev) : launch_both<T, true, false>(a, g, s, side, ev);
    return a.lr_swap ? launch_both<T, false, true>(a, g, s, side, ev) : launch_both<T, false, false>(a, g, s, side, ev);
}

hipError_t p2s_launch_tri(const P2sTriArgs &a, int dtype, const P2sTriLaunch &g, hipStream_t s, hipStream_t side,
                          hipEvent_t k1_done) {
    return dtype == 0 ? launch_t<float>(a, g, s, side, k1_done) : launch_t<double>(a, g, s, side, k1_done);
}
