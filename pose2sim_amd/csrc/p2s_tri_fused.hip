// p2s_tri_fused.hip -- one-launch robust triangulation for gfx950 (MI355X, CDNA4): the streaming pass AND the
// camera-subset search of triangulation_from_best_cameras (triangulation.py:363-604) in the same wave.
//
// The two-kernel form (p2s_tri.hip) hands the units whose level-0 error exceeds the threshold to a second, persistent
// kernel through a work list in HBM: 1.4x the algorithmic traffic, the level-0 state of every such unit computed
// twice, a second launch whose 3 072 waves finish unevenly.  Here the wave that streamed 64 units keeps the ones that
// need the search (12 % on BASELINE configs[1]) in LDS slots and walks their levels itself, in lock step: at level k the
// lanes split into groups of G = 2^g lanes, one group per pending unit; lane j of a group evaluates subset #(round G + j)
// of the level (itertools.combinations order, looked up in a table built once per calibration) -- normal matrix of the
// unit minus the removed cameras, the same eigen-solve and reprojection error as level 0 -- and the group's argmin
// (lowest rank on ties, np.nanargmin) goes back to the unit's slot.  Results leave the wave once, as 16-byte stores.
//
// Scope: pinhole path without L/R swap (the two options are off in every shipped configuration), up to
// 16 cameras; everything else takes the kernels of p2s_tri.hip.  One 64-lane workgroup per 64 consecutive units: the
// waves are independent, so a wave with many hard units delays nobody.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <algorithm>

#include "p2s_internal.h"
#include "p2s_tri_dev.h"

namespace {

constexpr int kSlots = 28;                 // units a wave searches at a time (more: further rounds)
constexpr uint32_t kNone = 0xffffffffu;

// A searching unit's state in LDS: what the lanes of its group need (normal matrix, masks, the lane that holds its
// observations in the wave's staging arrays) and the result of the level just finished (written by the winning lane
// of the group, read by the unit's own lane).
struct alignas(16) Slot {
    double N[10];
    uint32_t nan, zero;
    double err;
    double q[3];
    uint32_t S, owner;
};

// Observations of lane `owner` in the wave's staging arrays [camera][lane], for mean_error (x, y only; the
// kept-camera mask says which count).
template <typename T>
struct StagedObs {
    const T *xy;        // &sXY[0][owner][0]
    const T *w;         // &sW[0][owner]
    double lik_thr;
    __device__ __forceinline__ void raw(int c, double &x, double &y, double &wo) const {
        x = (double)xy[c * 128]; y = (double)xy[c * 128 + 1]; wo = 0.0;
    }
    __device__ __forceinline__ void rawT(int c, T &x, T &y, T &wo) const {
        x = xy[c * 128]; y = xy[c * 128 + 1]; wo = w[c * 64];
    }
};

// Lane u loads its C (x, y, likelihood) triplets straight into registers; consecutive lanes are consecutive keypoints,
// so each load instruction of the wave reads runs of 12-byte triplets that are contiguous per (frame, person) block,
// and over the C loads every byte of the covered blocks exactly once.
template <typename T, int CT, bool EXACT>
__device__ __forceinline__ void load_observations(const P2sTriArgs &a, int C, uint32_t b, uint32_t k, RegObs<T, CT> &obs) {
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

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename T, int CT, bool EXACT>
__global__ void __launch_bounds__(64, (CT <= 8 ? 3 : 2)) p2s_tri_fused_kernel(const P2sTriArgs a) {
    typedef Slot slot_t;
    // LDS of the wave: the observations of its 64 units [camera][lane] (written once, right after the loads: the
    // search reads them from here, and they need not stay in registers across the eigen-solve), the search slots, the
    // projection matrices for per-lane camera indices, the list of pending slots.
    __shared__ __align__(16) T sXY[CT][64][2];
    __shared__ __align__(16) T sW[CT][64];
    __shared__ __align__(16) unsigned char smem[sizeof(slot_t) * kSlots];
    __shared__ __align__(16) double sP[CT * 12];
    __shared__ uint32_t sList[kSlots];
    // results of the wave's 64 units, staged for the 16-byte stores at the end; they are parked here before the search
    // (instead of in 10 registers per lane across it) and the searching units overwrite theirs level by level
    __shared__ __align__(16) double sQ[64 * 3];
    __shared__ __align__(16) uint32_t sE[64];
    __shared__ __align__(16) uint32_t sM[64];
    __shared__ __align__(16) uint8_t sX[64];
    slot_t *slots = reinterpret_cast<slot_t *>(smem);

    const int C = EXACT ? CT : a.C;
    const int K = a.K;
    cam_cptr cams = (cam_cptr)a.cams;
    const int lane = threadIdx.x;
    const int64_t n_units = a.n_blocks * K;
    // Workgroups are dealt round-robin over the 8 XCDs: give each XCD a contiguous range of tiles, so that the partial
    // (frame, camera) runs two neighbouring tiles share are fetched into one L2 only.  (A persistent grid walking
    // several tiles per workgroup was 15 % slower: 313 vs 268 us on cfg2 under rocprofv3.)
    const uint32_t n_tiles = (uint32_t)((n_units + 63) >> 6);
    const uint32_t per_xcd = (n_tiles + 7u) >> 3;
    const uint32_t tile = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (tile >= n_tiles) return;
    const int64_t wave_u0 = (int64_t)tile << 6;                        // first unit of this wave within the chunk
    const int64_t lu = wave_u0 + lane;
    const bool active = lu < n_units;
    const uint32_t u = active ? (uint32_t)lu : (uint32_t)wave_u0;       // a chunk holds < 2^31 units
    const uint32_t b = u / (uint32_t)K;
    const uint32_t k = u - b * (uint32_t)K;

    // results (triangulation.py:588-604): the three doubles of a unit sit 24 bytes apart; transposing the wave's 64 x 3
    // block through LDS turns 8-byte-strided stores into 16-byte-per-lane stores of contiguous memory
    auto store_staged = [&]() {
        wave_sync();
        const int64_t gu0 = a.block0 * K + wave_u0;                         // first unit of this wave (global)
        const int64_t n_left = n_units - wave_u0;                           // units this wave owns
        double *Qw = a.Q + gu0 * 3;
        float *Ew = a.err + gu0;
        uint32_t *Mw = a.mask + gu0;
        uint8_t *Xw = a.n_excl + gu0;
        const bool al16 = ((reinterpret_cast<uintptr_t>(Qw) | reinterpret_cast<uintptr_t>(Ew) |
                            reinterpret_cast<uintptr_t>(Mw) | reinterpret_cast<uintptr_t>(Xw)) & 15) == 0;
        if (n_left >= 64 && al16) {
            typedef double v2d __attribute__((ext_vector_type(2)));
            typedef uint32_t v4u __attribute__((ext_vector_type(4)));
            const v2d *src = reinterpret_cast<const v2d *>(sQ);
            v2d *dst = reinterpret_cast<v2d *>(Qw);
            dst[lane] = src[lane];                                          // 1536 contiguous bytes: 64 lanes, then 32
            if (lane < 32) dst[64 + lane] = src[64 + lane];
            else if (lane < 48) reinterpret_cast<v4u *>(Ew)[lane - 32] = reinterpret_cast<const v4u *>(sE)[lane - 32];
            else if (lane < 52) reinterpret_cast<v4u *>(Xw)[lane - 48] = reinterpret_cast<const v4u *>(sX)[lane - 48];
            if (lane < 16) reinterpret_cast<v4u *>(Mw)[lane] = reinterpret_cast<const v4u *>(sM)[lane];
        } else {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int idx = r * 64 + lane;
                if (idx < 3 * n_left) Qw[idx] = sQ[idx];
            }
            if (active) { Ew[lane] = __uint_as_float(sE[lane]); Mw[lane] = sM[lane]; Xw[lane] = sX[lane]; }
        }
    };

    RegObs<T, CT> obs;
    obs.lik_thr = a.lik_thr;
    load_observations<T, CT, EXACT>(a, C, b, k, obs);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        sXY[c][lane][0] = obs.x[c]; sXY[c][lane][1] = obs.y[c];
        sW[c][lane] = obs.w[c];
    }

    const double thr = a.thr;
    const uint32_t allmask = (C == 32) ? 0xffffffffu : ((1u << C) - 1u);

    // ---- level 0 (triangulation.py:404-505 with nb_cams_off = 0) ------------------------------------------------------
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

    bool need;
    {
        double q[3];
        smallest_eigvec(N, q);
        if (nvalid < 2) { q[0] = d_nan(); q[1] = d_nan(); q[2] = d_nan(); }   // common.py:347: fewer than 4 rows
        const double e = mean_error<T, false, CT>(cams, C, obs, valid, q);
        const bool ran = Lmax >= 0;                            // else no level completes: inf, all cameras (:595-596)
        const bool ok = ran && (e <= thr);                     // :600-602
        sQ[lane * 3 + 0] = ok ? q[0] : d_nan();
        sQ[lane * 3 + 1] = ok ? q[1] : d_nan();
        sQ[lane * 3 + 2] = ok ? q[2] : d_nan();
        sE[lane] = __float_as_uint(ok ? (float)e : __builtin_nanf(""));
        sM[lane] = ran ? nanmask : allmask;
        sX[lane] = (uint8_t)(ran ? V : C);
        need = (Lmax >= 1) && (e > thr);                       // goes on to level 1
    }

    // ---- camera-subset search, in this wave ---------------------------------------------------------------------------
    const unsigned long long hard = __ballot(need);
    if (hard != 0ull) {
        uint32_t st_evals = 0, st_passes = 0;                  // p2s_get_tri_stats (wave-uniform counts)
        for (int i = lane; i < C * 12; i += 64) sP[i] = a.cams[i / 12].P[i % 12];
        const unsigned long long lt = (1ull << lane) - 1ull;
        const int n_hard = __popcll(hard);
        const int my_ord = __popcll(hard & lt);
        // the first kSlots searching units take their slots here, straight from registers (N dies with this write: left
        // inside the rounds loop it was spilled to scratch around it, 130 MB of HBM writes per cfg2 step)
        if (need && my_ord < kSlots) {
            slot_t &s = slots[my_ord];
#pragma unroll
            for (int i = 0; i < 10; ++i) s.N[i] = N[i];
            s.nan = nanmask; s.zero = zeromask; s.owner = (uint32_t)lane;
        }
        for (int first = 0; first < n_hard; first += kSlots) {             // rounds of at most kSlots units
            const bool mine = need && my_ord >= first && my_ord < first + kSlots;
            const int my_slot = mine ? my_ord - first : 0;
            if (first > 0) {
                // more than kSlots searching units in one wave (rare): the later ones rebuild their normal matrix from
                // the staged observations, so that it need not stay in registers across the search
                wave_sync();                                                // the previous round's slots are done with
                StagedObs<T> own{&sXY[0][lane][0], &sW[0][lane], a.lik_thr};
                double N2[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) N2[i] = 0.0;
                uint32_t nan2 = 0, zero2 = 0;
                classify_and_accumulate<T, CT>(cams, C, own, N2, nan2, zero2);
                if (mine) {
                    slot_t &s = slots[my_slot];
#pragma unroll
                    for (int i = 0; i < 10; ++i) s.N[i] = N2[i];
                    s.nan = nan2; s.zero = zero2; s.owner = (uint32_t)lane;
                }
            }
            bool cont = mine;                                               // this lane's unit goes on to `level`
            for (int level = 1;; ++level) {
                const unsigned long long pend = __ballot(cont);
                if (pend == 0ull) break;
                const int npend = __popcll(pend);
                if (cont) sList[__popcll(pend & lt)] = (uint32_t)my_slot;
                wave_sync();
                const uint32_t sub0 = a.sub_off[level];
                const uint32_t nsub = a.sub_off[level + 1] - sub0;
                // lanes per unit for this level: the power of two that needs the fewest
                // (passes over the pending units) x (rounds over the level's subsets)
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
                    const bool has = p0 + grp < npend;                      // this group has a unit in this pass
                    const slot_t &s = slots[has ? sList[p0 + grp] : sList[p0]];
                    const uint32_t o_d = s.nan | s.zero, o_valid = allmask & ~o_d;
                    const uint32_t owner = s.owner;
                    StagedObs<T> sobs{&sXY[0][owner][0], &sW[0][owner], a.lik_thr};

                    double be = kInf, bq0 = d_nan(), bq1 = d_nan(), bq2 = d_nan();   // best of this lane, lowest rank first
                    uint32_t brank = kNone, bS = 0;
                    for (uint32_t r0 = 0; r0 < nsub; r0 += G) {
                        const uint32_t r = r0 + lig;
                        bool go = has && (r < nsub);
                        uint32_t S = 0;
                        if (go) {
                            S = a.sub_tab[sub0 + r];
                            // duplicates of one effective configuration (quirk Q1: a subset that "removes" cameras which
                            // are out already) carry identical numbers; only the lexicographically first one -- its
                            // padding is the LOWEST cameras of the excluded set -- can win the argmin
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
                            const T x = sXY[c][owner][0], y = sXY[c][owner][1], w = sW[c][owner];
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
                        if (go && (e < be || brank == kNone)) { be = e; bq0 = q[0]; bq1 = q[1]; bq2 = q[2]; brank = r; bS = S; }
                    }
                    // group argmin, first (lowest-rank) index on ties (np.nanargmin, :502): what the group's first lane
                    // ends up with is the level's result, and the lane that holds it writes it to the slot
                    double ge = be;
                    uint32_t grank = brank;
                    for (int off = G >> 1; off > 0; off >>= 1) {
                        const double oe = __shfl_xor(ge, off, 64);
                        const uint32_t orank = (uint32_t)__shfl_xor((int)grank, off, 64);
                        const bool take = (orank != kNone) && (grank == kNone || oe < ge || (oe == ge && orank < grank));
                        if (take) { ge = oe; grank = orank; }
                    }
                    grank = (uint32_t)__shfl((int)grank, lane & ~(G - 1), 64);
                    if (has && brank != kNone && brank == grank) {
                        slot_t &d = slots[sList[p0 + grp]];
                        d.err = be; d.q[0] = bq0; d.q[1] = bq1; d.q[2] = bq2; d.S = bS;
                    }
                }
                wave_sync();
                if (cont) {
                    const slot_t &s = slots[my_slot];
                    const double e = s.err;
                    const uint32_t bS = s.S;
                    const bool ok = e <= thr;
                    cont = (e > thr) && (level + 1 <= Lmax);
                    sQ[lane * 3 + 0] = ok ? s.q[0] : d_nan();
                    sQ[lane * 3 + 1] = ok ? s.q[1] : d_nan();
                    sQ[lane * 3 + 2] = ok ? s.q[2] : d_nan();
                    sE[lane] = __float_as_uint(ok ? (float)e : __builtin_nanf(""));
                    sM[lane] = nanmask | bS;
                    sX[lane] = (uint8_t)(V + __popc(bS & valid));           // :436 counts NaN or zero
                }
                wave_sync();                                                // sList and the result fields are rewritten
            }
        }
        if (a.stats && lane == 0) {
            unsigned long long *st = a.stats + (size_t)(blockIdx.x % P2S_STAT_SHARDS) * P2S_STAT_STRIDE;
            atomicAdd(st + 0, (unsigned long long)n_hard);
            atomicAdd(st + 1, (unsigned long long)st_evals);
            atomicAdd(st + 2, (unsigned long long)st_passes);
        }
    }

    store_staged();
}

template <typename T, int CT>
hipError_t launch_ct(const P2sTriArgs &a, hipStream_t s) {
    const int64_t n_units = a.n_blocks * a.K;
    const int64_t n_tiles = (n_units + 63) / 64;
    const unsigned grid = (unsigned)(((n_tiles + 7) / 8) * 8);
    if (a.C == CT)
        hipLaunchKernelGGL((p2s_tri_fused_kernel<T, CT, true>), dim3(grid), dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL((p2s_tri_fused_kernel<T, CT, false>), dim3(grid), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace

bool p2s_tri_fused_supports(int C, int dtype, int undistort, int lr_swap) {
    if (undistort || lr_swap) return false;
    return dtype == 0 ? C <= 16 : C <= 8;
}

hipError_t p2s_launch_tri_fused(const P2sTriArgs &a, int dtype, hipStream_t s) {
    if (dtype == 0) {
        if (a.C <= 4) return launch_ct<float, 4>(a, s);
        if (a.C <= 8) return launch_ct<float, 8>(a, s);
        return launch_ct<float, 16>(a, s);
    }
    if (a.C <= 4) return launch_ct<double, 4>(a, s);
    return launch_ct<double, 8>(a, s);
}
