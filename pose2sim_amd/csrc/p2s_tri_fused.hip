// p2s_tri_fused.hip -- one-launch robust triangulation for gfx950 (MI355X, CDNA4): the streaming pass AND the
// camera-subset search of triangulation_from_best_cameras (triangulation.py:363-604) in the same wave.
//
// The two-kernel form (p2s_tri.hip) hands the units whose level-0 error exceeds the threshold to a second, persistent
// kernel through a work list in HBM: 1.4x the algorithmic traffic, the level-0 state of every such unit computed
// twice, a second launch whose waves finish unevenly.  Here the wave that streamed the units keeps the ones that need
// the search (12 % on BASELINE configs[1]) in LDS slots -- normal matrix, masks AND observations; nobody else's
// observations are staged -- and walks their levels itself, in lock step: at level k the lanes split into groups of
// G = 2^g lanes, one group per pending unit; lane j of a group evaluates subset #(round G + j) of the level
// (itertools.combinations order, looked up in a table built once per calibration) -- normal matrix of the unit minus
// the removed cameras, the same eigen-solve and reprojection error as level 0 -- and the group's argmin (lowest rank on
// ties, np.nanargmin) goes back to the unit's slot.  The lane with the slot's number looks after the unit (decides on
// the next level, rewrites its parked result), not the lane that streamed it, so the level-0 state of a tile does not
// outlive the tile.  Results leave the wave once, as 16-byte stores.
//
// Up to 8 cameras (float32 input) a wave streams TWO tiles of 64 units and searches their failures together: with one
// tile it has 7.7 searching units on average and its evaluation passes hold 8 (level 1: one group of 8 lanes per unit)
// or 2 (level 2) -- 69 % / 44 % full; pooled, the same evaluations take 13 % fewer passes (1.87 -> 1.62 per 64 units on
// configs[1], counted beforehand on the workload with the oracle).  9-16 cameras and float64 input: one tile per wave.
//
// Scope: pinhole path without L/R swap (the two options are off in every shipped configuration), up to
// 16 cameras (8 for float64 input); everything else takes the kernels of p2s_tri.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "p2s_internal.h"
#include "p2s_tri_dev.h"

namespace {

constexpr uint32_t kNone = 0xffffffffu;
constexpr int kSlots = 32;                 // units a wave searches at a time (more: further rounds, observations re-read)

template <typename T, int CT>
struct alignas(16) PSlot {
    double N[10];
    double err;
    double q[3];
    uint32_t nan, zero, S, owner;          // owner = (tile within the wave) * 64 + lane that streamed the unit
    T o[CT * 3];                           // x, y, likelihood per camera
};

template <typename T>
struct SlotObs {
    const T *o;
    double lik_thr;
    __device__ __forceinline__ void raw(int c, double &x, double &y, double &wo) const {
        x = (double)o[3 * c]; y = (double)o[3 * c + 1]; wo = 0.0;
    }
    __device__ __forceinline__ void rawT(int c, T &x, T &y, T &wo) const { x = o[3 * c]; y = o[3 * c + 1]; wo = o[3 * c + 2]; }
};

template <typename T, int CT, bool EXACT>
__device__ __forceinline__ void load_obs(const P2sTriArgs &a, int C, uint32_t b, uint32_t k, RegObs<T, CT> &obs) {
    const unsigned char *chunk = reinterpret_cast<const unsigned char *>(a.xyl) +
                                 (size_t)a.block0 * (size_t)C * (size_t)a.K * 3u * sizeof(T);
    const uint32_t voff = (b * (uint32_t)(C * a.K) + k) * (uint32_t)(3 * sizeof(T));   // < 2^32: chunked on the host
    const uint32_t cam_stride = (uint32_t)a.K * (uint32_t)(3 * sizeof(T));
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        if (EXACT || c < C) {
            const T *p = reinterpret_cast<const T *>(chunk + (size_t)c * cam_stride + voff);
            obs.x[c] = __builtin_nontemporal_load(p);
            obs.y[c] = __builtin_nontemporal_load(p + 1);
            obs.w[c] = __builtin_nontemporal_load(p + 2);
        } else {
            obs.x[c] = obs.y[c] = obs.w[c] = (T)0;
        }
    }
}

__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T, int CT, bool EXACT, int TPW>
__global__ void __launch_bounds__(64, 3) p2s_tri_fused_kernel(const P2sTriArgs a) {
    typedef PSlot<T, CT> slot_t;
    __shared__ __align__(16) unsigned char smem[sizeof(slot_t) * kSlots];
    __shared__ __align__(16) double sP[CT * 12];
    __shared__ uint32_t sList[kSlots];
    __shared__ uint8_t sOver[64 * TPW];                 // units that found no slot: owner ids, in order
    // results of the wave's TPW x 64 units, staged for the 16-byte stores at the end; the searching units overwrite
    // theirs level by level
    __shared__ __align__(16) double sQ[TPW * 64 * 3];
    __shared__ __align__(16) uint32_t sE[TPW * 64];
    __shared__ __align__(16) uint32_t sM[TPW * 64];
    __shared__ __align__(16) uint8_t sX[TPW * 64];
    slot_t *slots = reinterpret_cast<slot_t *>(smem);

    const int C = EXACT ? CT : a.C;
    const int K = a.K;
    cam_cptr cams = (cam_cptr)a.cams;
    const int lane = threadIdx.x;
    const int64_t n_units = a.n_blocks * K;
    const uint32_t n_tiles = (uint32_t)((n_units + 63) >> 6);
    // Workgroups are dealt round-robin over the 8 XCDs: every XCD gets a contiguous range of tiles, so that the partial
    // (frame, camera) runs two neighbouring tiles share are fetched into one L2 only; within it the first workgroups take TPW
    // tiles each and the last ones a single tile: the waves that start last are the short ones, so the grid drains in
    // half the time (a wave of two tiles runs ~32 us of a 240 us kernel).
    const uint32_t per_xcd = (n_tiles + 7u) >> 3;
    const uint32_t xj = blockIdx.x >> 3;
    if (xj >= a.pool_pairs + a.pool_singles) return;
    const bool paired = xj < a.pool_pairs;
    // A paired workgroup takes tiles xj and xj + pairs of the range (not two neighbours): neighbouring tiles share the
    // cache lines at their common edge, and this way they are still streamed by neighbouring workgroups at about the
    // same time (as neighbours of one wave, 16 us apart, the edge lines had left the L2: +16 % fetched bytes).
    const uint32_t tstride = paired ? a.pool_pairs : 0u;
    const uint32_t tile0 = (blockIdx.x & 7u) * per_xcd + (paired ? xj : TPW * a.pool_pairs + (xj - a.pool_pairs));
    const int my_tiles = paired ? TPW : 1;
    if (tile0 >= n_tiles) return;
    const double thr = a.thr;
    const uint32_t allmask = (C == 32) ? 0xffffffffu : ((1u << C) - 1u);
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int i = lane; i < C * 12; i += 64) sP[i] = a.cams[i / 12].P[i % 12];

    // ---- level 0 of every tile (triangulation.py:404-505 with nb_cams_off = 0) ----------------------------------------
    int n_hard = 0;                                                     // searching units so far (wave-uniform)
    auto unit_of = [&](uint32_t tile, bool &active) -> uint32_t {
        const int64_t lu = ((int64_t)tile << 6) + lane;
        active = lu < n_units;
        return active ? (uint32_t)lu : (uint32_t)(tile << 6);
    };
    // Level 0 of one tile.  `prefetch` is called once the eigen-solve is through (the point of highest register
    // pressure): the first tile requests the second tile's observations there, so that they travel during its
    // reprojection pass instead of after it.
    auto level0 = [&](const int t, const bool active, const RegObs<T, CT> &obs, auto &&prefetch) {
        double N[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) N[i] = 0.0;
        uint32_t nanmask = 0, zeromask = 0;
        classify_and_accumulate<T, CT>(cams, C, obs, N, nanmask, zeromask);
        const uint32_t dmask = nanmask | zeromask;
        const uint32_t valid = allmask & ~dmask;
        const int V = __popc(dmask);
        const int Lmax = active ? C - a.min_cams - V : -1;              // last level that runs (:408, :437-441)
        double q[3];
        smallest_eigvec(N, q);
        prefetch(q[0]);
        if (C - V < 2) { q[0] = d_nan(); q[1] = d_nan(); q[2] = d_nan(); }   // common.py:347: fewer than 4 rows
        const double e = mean_error<T, false, CT>(cams, C, obs, valid, q);
        const bool ran = Lmax >= 0;                                     // else no level completes (:595-596)
        const bool ok = ran && (e <= thr);                              // :600-602
        const int o = t * 64 + lane;
        sQ[o * 3 + 0] = ok ? q[0] : d_nan();
        sQ[o * 3 + 1] = ok ? q[1] : d_nan();
        sQ[o * 3 + 2] = ok ? q[2] : d_nan();
        sE[o] = __float_as_uint(ok ? (float)e : __builtin_nanf(""));
        sM[o] = ran ? nanmask : allmask;
        sX[o] = (uint8_t)(ran ? V : C);
        const bool need = (Lmax >= 1) && (e > thr);                     // goes on to level 1
        const unsigned long long hard = __ballot(need);
        if (need) {
            const int ord = n_hard + __popcll(hard & lt);
            if (ord < kSlots) {
                slot_t &s = slots[ord];
#pragma unroll
                for (int i = 0; i < 10; ++i) s.N[i] = N[i];
                s.nan = nanmask; s.zero = zeromask; s.owner = (uint32_t)o;
#pragma unroll
                for (int c = 0; c < CT; ++c) { s.o[3 * c] = obs.x[c]; s.o[3 * c + 1] = obs.y[c]; s.o[3 * c + 2] = obs.w[c]; }
            } else {
                sOver[ord - kSlots] = (uint8_t)o;
            }
        }
        n_hard += __popcll(hard);
    };
    RegObs<T, CT> obs0, obs1, obs2;
    obs0.lik_thr = a.lik_thr; obs1.lik_thr = a.lik_thr; obs2.lik_thr = a.lik_thr;
    bool act0, act1 = false, act2 = false;
    const uint32_t u0 = unit_of(tile0, act0);
    load_obs<T, CT, EXACT>(a, C, u0 / (uint32_t)K, u0 % (uint32_t)K, obs0);
    const bool two = my_tiles > 1 && tile0 + tstride < n_tiles;
    const bool three = TPW > 2 && my_tiles > 2 && tile0 + 2 * tstride < n_tiles;
    level0(0, act0, obs0, [&](double dep) {
        if (two) {
            uint32_t u1 = unit_of(tile0 + tstride, act1);
            asm volatile("" : "+v"(u1) : "v"(dep));                     // not before the eigen-solve
            load_obs<T, CT, EXACT>(a, C, u1 / (uint32_t)K, u1 % (uint32_t)K, obs1);
        }
    });
    if (two) level0(1, act1, obs1, [&](double dep) {
        if constexpr (TPW > 2) {
            if (three) {
                uint32_t u2 = unit_of(tile0 + 2 * tstride, act2);
                asm volatile("" : "+v"(u2) : "v"(dep));
                load_obs<T, CT, EXACT>(a, C, u2 / (uint32_t)K, u2 % (uint32_t)K, obs2);
            }
        }
    });
    if constexpr (TPW > 2) {
        if (three) level0(2, act2, obs2, [](double) {});
    }

    // ---- camera-subset search over the pooled units: lane s looks after slot s -----------------------------------------
    if (n_hard != 0) {
        uint32_t st_evals = 0, st_passes = 0;
        for (int first = 0; first < n_hard; first += kSlots) {
            const int cnt = min(kSlots, n_hard - first);
            if (first > 0) {
                // more searching units than slots (rare): the later ones read their observations again and rebuild
                // their normal matrix
                wsync();
                if (lane < cnt) {
                    const int o = sOver[first - kSlots + lane];
                    const uint32_t u = ((tile0 + (uint32_t)(o >> 6) * tstride) << 6) + (uint32_t)(o & 63);
                    const uint32_t b = u / (uint32_t)K, k = u - b * (uint32_t)K;
                    RegObs<T, CT> ob;
                    ob.lik_thr = a.lik_thr;
                    load_obs<T, CT, EXACT>(a, C, b, k, ob);
                    double N2[10];
#pragma unroll
                    for (int i = 0; i < 10; ++i) N2[i] = 0.0;
                    uint32_t nan2 = 0, zero2 = 0;
                    classify_and_accumulate<T, CT>(cams, C, ob, N2, nan2, zero2);
                    slot_t &s = slots[lane];
#pragma unroll
                    for (int i = 0; i < 10; ++i) s.N[i] = N2[i];
                    s.nan = nan2; s.zero = zero2; s.owner = (uint32_t)o;
#pragma unroll
                    for (int c = 0; c < CT; ++c) { s.o[3 * c] = ob.x[c]; s.o[3 * c + 1] = ob.y[c]; s.o[3 * c + 2] = ob.w[c]; }
                }
            }
            wsync();
            bool cont = lane < cnt;                                     // this lane's slot goes on to `level`
            const slot_t &mine = slots[cont ? lane : 0];
            const uint32_t m_nan = mine.nan, m_d = mine.nan | mine.zero, m_valid = allmask & ~m_d;
            const int m_V = __popc(m_d), m_Lmax = C - a.min_cams - m_V, m_owner = (int)mine.owner;
            for (int level = 1;; ++level) {
                const unsigned long long pend = __ballot(cont);
                if (pend == 0ull) break;
                const int npend = __popcll(pend);
                if (cont) sList[__popcll(pend & lt)] = (uint32_t)lane;
                wsync();
                const uint32_t sub0 = a.sub_off[level];
                const uint32_t nsub = a.sub_off[level + 1] - sub0;
                int lg = 2;
                {
                    uint32_t best_cost = 0xffffffffu;
                    for (int l = 2; l <= 6; ++l) {
                        const uint32_t passes = (((uint32_t)npend << l) + 63u) >> 6;
                        const uint32_t rounds = (nsub + (1u << l) - 1u) >> l;
                        const uint32_t cost = passes * rounds;
                        if (cost <= best_cost) { best_cost = cost; lg = l; }
                    }
                }
                const int G = 1 << lg, groups = 64 >> lg;
                const int grp = lane >> lg, lig = lane & (G - 1);
                for (int p0 = 0; p0 < npend; p0 += groups) {
                    const bool has = p0 + grp < npend;
                    const slot_t &s = slots[has ? sList[p0 + grp] : sList[p0]];
                    const uint32_t o_d = s.nan | s.zero, o_valid = allmask & ~o_d;
                    SlotObs<T> sobs{s.o, a.lik_thr};
                    double be = kInf, bq0 = d_nan(), bq1 = d_nan(), bq2 = d_nan();
                    uint32_t brank = kNone, bS = 0;
                    for (uint32_t r0 = 0; r0 < nsub; r0 += G) {
                        const uint32_t r = r0 + lig;
                        bool go = has && (r < nsub);
                        uint32_t S = 0;
                        if (go) {
                            // level 1 is rank r <-> camera r (itertools.combinations order); only the deeper levels go to
                            // the table in global memory (a load per pass, and its latency)
                            S = (level == 1) ? (1u << r) : (uint32_t)a.sub_tab[sub0 + r];
                            // quirk Q1 duplicates: only the lexicographically first padding can win the argmin
                            const uint32_t pad = S & o_d;
                            const uint32_t below = pad ? ((2u << (31 - __builtin_clz(pad))) - 1u) : 0u;
                            go = (o_d & below) == pad;
                        }
                        if (!__any(go)) continue;
                        st_evals += (uint32_t)__popcll(__ballot(go)); ++st_passes;
                        const uint32_t Rreal = S & o_valid;
                        const uint32_t kept = o_valid & ~Rreal;
                        const int nkept = __popc(kept);
                        double Ns[10];
#pragma unroll
                        for (int i = 0; i < 10; ++i) Ns[i] = s.N[i];
                        for (uint32_t rr = go ? Rreal : 0u; __any(rr != 0u); rr &= rr - 1) {
                            const bool on = rr != 0u;
                            const int c = on ? __builtin_ctz(rr) : 0;
                            const T x = s.o[3 * c], y = s.o[3 * c + 1], w = s.o[3 * c + 2];
                            accum_camera<-1>(Ns, sP + c * 12, (double)(on ? x : (T)0), (double)(on ? y : (T)0),
                                             (double)(on ? w : (T)0));
                        }
                        double q[3];
                        smallest_eigvec(Ns, q);
                        if (nkept < 2) { q[0] = d_nan(); q[1] = d_nan(); q[2] = d_nan(); }
                        // the projection matrices are scalar loads at their point of use: an opaque copy of the pointer
                        // keeps the compiler from hoisting 192 SGPRs' worth of them out of the loops (it then spills
                        // them into VGPR lanes and pays a v_readlane per operand)
                        cam_cptr cams_here = cams;
                        asm volatile("" : "+s"(cams_here));
                        const double e = mean_error<T, false, CT>(cams_here, C, sobs, kept, q);
                        // (a lane meets its candidates in rising rank: a lower error wins, and a number replaces a NaN --
                        // np.nanargmin, triangulation.py:500-503)
                        if (go && (brank == kNone || e < be || (be != be && e == e))) { be = e; bq0 = q[0]; bq1 = q[1]; bq2 = q[2]; brank = r; bS = S; }
                    }
                    double ge = be;
                    uint32_t grank = brank;
                    for (int off = G >> 1; off > 0; off >>= 1) {
                        const double oe = __shfl_xor(ge, off, 64);
                        const uint32_t orank = (uint32_t)__shfl_xor((int)grank, off, 64);
                        const bool take = better_candidate(oe, orank, ge, grank);
                        if (take) { ge = oe; grank = orank; }
                    }
                    grank = (uint32_t)__shfl((int)grank, lane & ~(G - 1), 64);
                    if (has && brank != kNone && brank == grank) {
                        slot_t &d = slots[sList[p0 + grp]];
                        d.err = be; d.q[0] = bq0; d.q[1] = bq1; d.q[2] = bq2; d.S = bS;
                    }
                }
                wsync();
                if (cont) {
                    const slot_t &s = slots[lane];
                    const double e = s.err;
                    const uint32_t bS = s.S;
                    const bool ok = e <= thr;
                    cont = (e > thr) && (level + 1 <= m_Lmax);
                    sQ[m_owner * 3 + 0] = ok ? s.q[0] : d_nan();
                    sQ[m_owner * 3 + 1] = ok ? s.q[1] : d_nan();
                    sQ[m_owner * 3 + 2] = ok ? s.q[2] : d_nan();
                    sE[m_owner] = __float_as_uint(ok ? (float)e : __builtin_nanf(""));
                    sM[m_owner] = m_nan | bS;
                    sX[m_owner] = (uint8_t)(m_V + __popc(bS & m_valid));   // :436 counts NaN or zero
                }
                wsync();
            }
        }
        if (a.stats && lane == 0) {
            unsigned long long *st = a.stats + (size_t)(blockIdx.x % P2S_STAT_SHARDS) * P2S_STAT_STRIDE;
            atomicAdd(st + 0, (unsigned long long)n_hard);
            atomicAdd(st + 1, (unsigned long long)st_evals);
            atomicAdd(st + 2, (unsigned long long)st_passes);
        }
    }

    // ---- results (triangulation.py:588-604), 16-byte stores of contiguous memory per tile ------------------------------
    wsync();
#pragma unroll 1
    for (int t = 0; t < my_tiles; ++t) {
        const uint32_t tile = tile0 + t * tstride;
        if (tile >= n_tiles) continue;
        const int64_t wave_u0 = (int64_t)tile << 6;
        const int64_t gu0 = a.block0 * K + wave_u0;
        const int64_t n_left = n_units - wave_u0;
        double *Qw = a.Q + gu0 * 3;
        float *Ew = a.err + gu0;
        uint32_t *Mw = a.mask + gu0;
        uint8_t *Xw = a.n_excl + gu0;
        const double *tQ = sQ + t * 192;
        const uint32_t *tE = sE + t * 64, *tM = sM + t * 64;
        const uint8_t *tX = sX + t * 64;
        const bool al16 = ((reinterpret_cast<uintptr_t>(Qw) | reinterpret_cast<uintptr_t>(Ew) |
                            reinterpret_cast<uintptr_t>(Mw) | reinterpret_cast<uintptr_t>(Xw)) & 15) == 0;
        if (n_left >= 64 && al16) {
            typedef double v2d __attribute__((ext_vector_type(2)));
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            const v2d *src = reinterpret_cast<const v2d *>(tQ);
            v2d *dst = reinterpret_cast<v2d *>(Qw);
            dst[lane] = src[lane];
            if (lane < 32) dst[64 + lane] = src[64 + lane];
            else if (lane < 48) reinterpret_cast<v4u *>(Ew)[lane - 32] = reinterpret_cast<const v4u *>(tE)[lane - 32];
            else if (lane < 52) reinterpret_cast<v4u *>(Xw)[lane - 48] = reinterpret_cast<const v4u *>(tX)[lane - 48];
            if (lane < 16) reinterpret_cast<v4u *>(Mw)[lane] = reinterpret_cast<const v4u *>(tM)[lane];
        } else {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int idx = r * 64 + lane;
                if (idx < 3 * n_left) Qw[idx] = tQ[idx];
            }
            if (lane < n_left) { Ew[lane] = __uint_as_float(tE[lane]); Mw[lane] = tM[lane]; Xw[lane] = tX[lane]; }
        }
    }
}

template <typename T, int CT, int TPW>
hipError_t launch_fused(P2sTriArgs a, int singles_pct, hipStream_t s) {
    const int64_t n_units = a.n_blocks * a.K;
    const int64_t n_tiles = (n_units + 63) / 64;
    const int64_t per_xcd = (n_tiles + 7) / 8;
    int64_t singles = per_xcd * singles_pct / 100;
    singles += (per_xcd - singles) % TPW;                      // the rest in whole groups of TPW
    a.pool_singles = (uint32_t)singles;
    a.pool_pairs = (uint32_t)((per_xcd - singles) / TPW);
    const unsigned grid = (unsigned)(8 * (a.pool_pairs + a.pool_singles));
    if (a.C == CT)
        hipLaunchKernelGGL((p2s_tri_fused_kernel<T, CT, true, TPW>), dim3(grid), dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL((p2s_tri_fused_kernel<T, CT, false, TPW>), dim3(grid), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace

bool p2s_tri_fused_supports(int C, int dtype, int undistort, int lr_swap) {
    if (undistort || lr_swap) return false;
    return dtype == 0 ? C <= 16 : C <= 8;
}

hipError_t p2s_launch_tri_fused(const P2sTriArgs &a, int dtype, int singles_pct, hipStream_t s) {
    if (dtype != 0) {                                          // float64 observations: one tile per wave
        if (a.C <= 4) return launch_fused<double, 4, 1>(a, 100, s);
        return launch_fused<double, 8, 1>(a, 100, s);
    }
    // two tiles per wave: three (14.6 KB of LDS, 11 waves per CU) have 6 % fewer passes again and take 2 % longer
    if (a.C <= 4) return launch_fused<float, 4, 2>(a, singles_pct, s);
    // 9-16 cameras: ONE tile per wave (12.7 KB of LDS; staging the observations of all 64 units took 19.6 KB and left two
    // waves per SIMD: 3.70 -> 3.60 ms on the 16-camera shard; two tiles: 16.3 KB, spills, 22 % slower)
    if (a.C > 8) return launch_fused<float, 16, 1>(a, 100, s);
    return launch_fused<float, 8, 2>(a, singles_pct, s);
}
