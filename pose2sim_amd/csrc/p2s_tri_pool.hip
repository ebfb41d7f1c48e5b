// p2s_tri_pool.hip -- one-launch robust triangulation for gfx950 (MI355X, CDNA4), round 3: a wave streams a few tiles of
// 64 units, POOLS the units whose level-0 error exceeds the threshold in LDS slots across its tiles, and searches their
// camera subsets (triangulation_from_best_cameras, triangulation.py:363-604) in two tiers:
//
//   tier A (screen, fp32, two subsets per lane in packed registers): every candidate subset of a level is evaluated in
//     single precision in coordinates centred on the unit's level-0 point -- normal matrix of the kept cameras by
//     downdate, smallest eigenpair by two Rayleigh steps, mean reprojection error.  It decides nothing by itself: a
//     candidate is dropped only if its fp32 error, minus a margin that is 30x the largest deviation from the fp64 error
//     seen on any workload (exp/screen_proto.py, tests/sweeps), cannot be the level's minimum and -- on a level that is
//     not the unit's last, so that a failed level leaves nothing behind -- cannot be under the threshold.  Candidates
//     the screen cannot vouch for (ill-conditioned system, eigen-iteration not settled, degenerate projection, level-0
//     point far away or not finite) always go on.
//   tier B (fp64): the survivors -- 0.93 per searching unit on BASELINE configs[1] instead of 8 -- of ALL the pooled
//     units and levels are evaluated together, one per lane, by exactly the arithmetic of level 0; the argmin per unit
//     (error, then rank: np.nanargmin's first index) goes through LDS atomics.
//
// A unit whose level certainly failed in tier A goes on to the next level's screen at once, so that one fp64 pass
// serves the survivors of several levels.  Every number that reaches a result comes from tier B: with the screen switched
// off (P2S_TUNE_SCREEN 0: every candidate survives) the outputs are bit-identical (tests/test_tri_gpu.py).
//
// The results of a tile leave the wave as soon as its level 0 is through (16-byte stores); the searched units' results
// are patched over them after the search.  That takes the results' staging out of the LDS budget, and up to 8 cameras the
// slot keeps no fp64 normal matrix either (tier B accumulates the kept cameras from the observations: 192 bytes per
// slot), so a wave pools five tiles (38 searching units on configs[1]) where round 2's kernel pooled two; the tiles are
// taken in straight-line code, fresh waves by the dispatcher: a persistent loop over tiles streamed 15-60 % slower
// (DESIGN.md 4.9).
//
// 9-16 cameras: the observations are taken eight cameras at a time (48 registers of them beside the eigen-solve
// spilled); x and y stay in registers for the reprojection pass, the lanes that park a unit read theirs again.
//
// Scope: pinhole path without L/R swap (both off in every shipped configuration), float32 observations,
// up to 16 cameras; float64 observations take round 2's one-launch kernel (p2s_tri_fused.hip), everything else the
// kernels of p2s_tri.hip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "p2s_internal.h"
#include "p2s_tri_dev.h"

namespace {

#ifndef P2S_POOL_SLOTS4
#define P2S_POOL_SLOTS4 32
#endif
#ifndef P2S_POOL_WIDE_REREAD
#define P2S_POOL_WIDE_REREAD 0       // 9-16 cameras: 1 = x and y are read again for the reprojection pass instead of kept (32 registers)
#endif
#ifndef P2S_POOL_WIDE_STREAM
#define P2S_POOL_WIDE_STREAM 0
#endif
#ifndef P2S_POOL_CACHED_LOADS
#define P2S_POOL_CACHED_LOADS 1   // the observations of up to 8 cameras through ordinary loads, not non-temporal ones: the 128-byte
                                  // lines that two neighbouring tiles share then survive in L2 until the second one comes
                                  // (HBM reads 1.095x -> 1.013x of the algorithmic bytes on configs[1], same time)
#endif
#ifndef P2S_POOL_WPS
#define P2S_POOL_WPS 3          // waves per SIMD the register allocation aims at
#endif
constexpr uint32_t kNone = 0xffffffffu;
constexpr float kInfF = __builtin_huge_valf();

// Up to 8 cameras tier B builds a survivor's normal matrix from the unit's observations (8 cameras accumulated cost less
// than the 80 bytes per slot that would keep level 0's matrix: slots are what limits the tiles a wave can pool); beyond
// that it removes the subset's cameras from level 0's matrix, kept in the slot (with level 0's point: the screen's base is
// then built when the search starts, not in the streaming part where registers are short).
template <bool KEEP> struct SlotNormal { double N[10]; double q0[3]; };
template <> struct SlotNormal<false> {};
template <typename T, int CT>
struct alignas(16) QSlot : SlotNormal<(CT > 8)> {
    unsigned long long ebits;              // verified best error of the current level (bits of a non-negative double), ~0 = none
    T o[CT * 3];                           // x, y, likelihood per camera (a camera that does not count: zeros)
    float M[6], g[3], h, c0[3];            // screen base: normal matrix of all valid cameras about c0 (the level-0 point)
    uint32_t rank, S;                      // verified best: rank in itertools order (atomic min among equal errors), subset
    uint32_t nan, zero;
    uint32_t unit;                         // unit index within the chunk
    uint32_t surv;                         // the current level's screen let somebody through
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
        // (a camera beyond C reads camera 0's address and is zeroed: loads under a branch made the compiler keep the
        // whole structure in scratch memory in the kernels compiled for a range of camera counts)
        const bool there = EXACT || c < C;
        const T *p = reinterpret_cast<const T *>(chunk + (size_t)(there ? c : 0) * cam_stride + voff);
#if P2S_POOL_CACHED_LOADS
        const T x = p[0], y = p[1], w = p[2];
#else
        const T x = __builtin_nontemporal_load(p), y = __builtin_nontemporal_load(p + 1), w = __builtin_nontemporal_load(p + 2);
#endif
        obs.x[c] = there ? x : (T)0; obs.y[c] = there ? y : (T)0; obs.w[c] = there ? w : (T)0;
    }
}

// Cameras c0 .. c0 + 7 of one unit (the 9-16 camera kernels work on the observations eight cameras at a time: 48 registers
// of observations beside the eigen-solve spilled ~400 B per lane in round 2).
template <typename T, bool EXACT, bool WITH_W = true, bool STREAM = true>
__device__ __forceinline__ void load_half(const P2sTriArgs &a, int C, uint32_t b, uint32_t k, int c0, RegObs<T, 8> &obs) {
    const unsigned char *chunk = reinterpret_cast<const unsigned char *>(a.xyl) +
                                 (size_t)a.block0 * (size_t)C * (size_t)a.K * 3u * sizeof(T);
    const uint32_t voff = (b * (uint32_t)(C * a.K) + k) * (uint32_t)(3 * sizeof(T));
    const uint32_t cam_stride = (uint32_t)a.K * (uint32_t)(3 * sizeof(T));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool there = EXACT || c0 + i < C;
        const T *p = reinterpret_cast<const T *>(chunk + (size_t)(there ? c0 + i : 0) * cam_stride + voff);
        T x, y, w = (T)0;
        if (STREAM) {
            x = __builtin_nontemporal_load(p); y = __builtin_nontemporal_load(p + 1);
            if (WITH_W) w = __builtin_nontemporal_load(p + 2);
        } else {                                                   // read again later: let the caches keep the lines
            x = p[0]; y = p[1];
            if (WITH_W) w = p[2];
        }
        obs.x[i] = there ? x : (T)0; obs.y[i] = there ? y : (T)0; obs.w[i] = there ? w : (T)0;
    }
}

__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- tier A arithmetic (fp32, TWO candidates per lane in the halves of packed registers) ---------------------------------
// A wave64 v_pk_fma_f32 does two FMAs per lane in ~4.5 cycles of its SIMD against ~2.9 for one v_fma_f32
// (exp/valu_rates.hip), and everything that is not arithmetic -- observations and projection matrices fetched, loop and
// mask bookkeeping -- is paid once for the two.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f splat2(float x) { return v2f{x, x}; }
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f rcp2(v2f a) { return v2f{__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}; }
__device__ __forceinline__ v2f rsq2(v2f a) { return v2f{__builtin_amdgcn_rsqf(a.x), __builtin_amdgcn_rsqf(a.y)}; }
struct Sym3v { v2f m00, m01, m02, m11, m12, m22; };

// q = (M - lam I)^-1 r through the adjugate; det and trace of M - lam I for the conditioning guard.
__device__ __forceinline__ void solve3v(const Sym3v &M, v2f lam, v2f r0, v2f r1, v2f r2, v2f &q0, v2f &q1, v2f &q2, v2f &det,
                                        v2f &tr, v2f &a00, v2f &c22) {
    a00 = M.m00 - lam;
    const v2f a11 = M.m11 - lam, a22 = M.m22 - lam;
    const v2f c00 = fma2(a11, a22, -M.m12 * M.m12);
    const v2f c01 = fma2(M.m02, M.m12, -M.m01 * a22);
    const v2f c02 = fma2(M.m01, M.m12, -M.m02 * a11);
    const v2f c11 = fma2(a00, a22, -M.m02 * M.m02);
    const v2f c12 = fma2(M.m01, M.m02, -a00 * M.m12);
    c22 = fma2(a00, a11, -M.m01 * M.m01);
    det = fma2(a00, c00, fma2(M.m01, c01, M.m02 * c02));
    const v2f id = rcp2(det);
    q0 = fma2(c00, r0, fma2(c01, r1, c02 * r2)) * id;
    q1 = fma2(c01, r0, fma2(c11, r1, c12 * r2)) * id;
    q2 = fma2(c02, r0, fma2(c12, r1, c22 * r2)) * id;
    tr = a00 + a11 + a22;
}

// Rayleigh quotient of v = (q, 1) for the pencil (N', G) of the shifted problem: N' = [[M, g], [g^T, h]],
// G = T^T T with T the translation by c0, i.e. v^T G v = |c0 + q|^2 + 1.
__device__ __forceinline__ v2f rayleighv(const Sym3v &M, v2f g0, v2f g1, v2f g2, v2f h, v2f c0, v2f c1, v2f c2, v2f q0, v2f q1, v2f q2) {
    const v2f t0 = fma2(M.m00, q0, fma2(M.m01, q1, M.m02 * q2));
    const v2f t1 = fma2(M.m01, q0, fma2(M.m11, q1, M.m12 * q2));
    const v2f t2 = fma2(M.m02, q0, fma2(M.m12, q1, M.m22 * q2));
    const v2f two = splat2(2.0f);
    const v2f num = fma2(q0, fma2(two, g0, t0), fma2(q1, fma2(two, g1, t1), fma2(q2, fma2(two, g2, t2), h)));
    const v2f Q0 = c0 + q0, Q1 = c1 + q1, Q2 = c2 + q2;
    const v2f den = fma2(Q0, Q0, fma2(Q1, Q1, fma2(Q2, Q2, splat2(1.0f))));
    return num * rcp2(den);
}

// The screen's margin and guards (exp/screen_proto.py): with det / tr^3 >= 3e-3 (condition number of M - lam I below
// ~330), the last two eigenvalue estimates within 25 % of each other, a finite result and the level-0 point within 30 m,
// |e32 - e64| stayed under 0.03 (0.02 px + e (1e-3 + 2 dlam)) on every candidate of every workload tried.
constexpr float kCondMin = 3e-3f, kDlamMax = 0.25f, kMargAbs = 0.02f, kMargRel = 1e-3f, kMargDlam = 2.0f, kCentreMax2 = 900.0f;

template <typename T, int CT, int NSLOT, bool EXACT, int TPW>
__global__ void __launch_bounds__(64, P2S_POOL_WPS) p2s_tri_pool_kernel(const P2sTriArgs a) {
    typedef QSlot<T, CT> slot_t;
    constexpr bool KEEPN = CT > 8;
    __shared__ __align__(16) unsigned char smem[sizeof(slot_t) * NSLOT];
    __shared__ __align__(16) double sP[KEEPN ? CT * 12 : 1];
    __shared__ __align__(16) float sPf[CT * 12];
    __shared__ uint16_t sOver[64 * TPW];                  // units that found no slot: (tile within the wave) * 64 + lane, in order
    // One region, two lives: the results of the tile being streamed, staged for its 16-byte stores -- and, while a search
    // runs (the tile's results have left by then), the list of pending slots and the survivors awaiting tier B.
    __shared__ __align__(16) unsigned char sShare[64 * 24 + 64 * 4 + 64 * 4 + 64];
    slot_t *slots = reinterpret_cast<slot_t *>(smem);
    double *sQ = reinterpret_cast<double *>(sShare);
    uint32_t *sE = reinterpret_cast<uint32_t *>(sShare + 1536);
    uint32_t *sM = reinterpret_cast<uint32_t *>(sShare + 1792);
    uint8_t *sX = sShare + 2048;
    uint2 *sSurv = reinterpret_cast<uint2 *>(sShare);    // 192 entries: {slot | subset << 8, rank}
    uint32_t *sList = reinterpret_cast<uint32_t *>(sShare + 1536);

    const int C = EXACT ? CT : a.C;
    const int K = a.K;
    cam_cptr cams = (cam_cptr)a.cams;
    const int lane = threadIdx.x;
    const int64_t n_units = a.n_blocks * K;
    const uint32_t n_tiles = (uint32_t)((n_units + 63) >> 6);
    // Workgroups are dealt round-robin over the 8 XCDs: every XCD gets a contiguous range of tiles, so that the partial
    // (frame, camera) runs two neighbouring tiles share are fetched into one L2 only; within it the first workgroups take
    // TPW tiles each (tiles xj, xj + n, xj + 2n of the range, n = such workgroups per XCD: neighbouring tiles are streamed
    // by neighbouring workgroups at about the same time) and the last ones a single tile: the waves that start last are
    // the short ones, so the grid drains quickly.
    const uint32_t per_xcd = (n_tiles + 7u) >> 3;
    const uint32_t xj = blockIdx.x >> 3;
    if (xj >= a.pool_pairs + a.pool_singles) return;
    const bool multi = xj < a.pool_pairs;
    const uint32_t tstride = multi ? a.pool_pairs : 0u;
    const uint32_t tile0 = (blockIdx.x & 7u) * per_xcd + (multi ? xj : TPW * a.pool_pairs + (xj - a.pool_pairs));
    const int my_tiles = multi ? TPW : 1;
    if (tile0 >= n_tiles) return;
    const double thr = a.thr;
    const float thr_f = (float)(thr * (1.0 + 1e-6));
    const uint32_t allmask = (C == 32) ? 0xffffffffu : ((1u << C) - 1u);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const bool screen_on = a.screen != 0;
    // likelihood test in the observations' own precision: for a float w, (double)w < t  <=>  w < (smallest float >= t)
    T lik_t = (T)a.lik_thr;
    if ((double)lik_t < a.lik_thr) lik_t = (T)nextafterf((float)lik_t, __builtin_huge_valf());
    for (int i = lane; i < C * 12; i += 64) {
        sPf[i] = (float)a.cams[i / 12].P[i % 12];
        if constexpr (KEEPN) sP[i] = a.cams[i / 12].P[i % 12];
    }

    int n_used = 0, n_over = 0;                          // slots in use, units waiting in sOver (wave-uniform)
    uint32_t st_units = 0, st_evals = 0, st_passes = 0, st_screened = 0, st_spasses = 0;

    auto unit_of = [&](uint32_t t, bool &active) -> uint32_t {
        const int64_t lu = ((int64_t)t << 6) + lane;
        active = lu < n_units;
        return active ? (uint32_t)lu : (uint32_t)(t << 6);
    };
    // A slot keeps what the screen needs of level 0 -- its base: N' = T^T N T about c0 = the level-0 point, in fp64 and
    // then rounded (g and h are what is left of M c0 + b and c0.(M c0 + 2b) + c after cancellation) -- and the unit's
    // observations; tier B builds its normal matrices from those.
    auto build_base = [&](slot_t &s, const double N[10], const double q[3]) {
        const double c0 = q[0], c1 = q[1], c2 = q[2];
        const double g0 = fma(N[0], c0, fma(N[1], c1, fma(N[2], c2, N[3])));
        const double g1 = fma(N[1], c0, fma(N[4], c1, fma(N[5], c2, N[6])));
        const double g2 = fma(N[2], c0, fma(N[5], c1, fma(N[7], c2, N[8])));
        const double hh = fma(c0, g0 + N[3], fma(c1, g1 + N[6], fma(c2, g2 + N[8], N[9])));
        s.M[0] = (float)N[0]; s.M[1] = (float)N[1]; s.M[2] = (float)N[2];
        s.M[3] = (float)N[4]; s.M[4] = (float)N[5]; s.M[5] = (float)N[7];
        s.g[0] = (float)g0; s.g[1] = (float)g1; s.g[2] = (float)g2; s.h = (float)hh;
        const bool centred = (c0 * c0 + c1 * c1 + c2 * c2) < (double)kCentreMax2;   // false for NaN
        s.c0[0] = centred ? (float)c0 : __builtin_nanf(""); s.c0[1] = (float)c1; s.c0[2] = (float)c2;
        s.ebits = ~0ull; s.rank = kNone; s.S = 0u; s.surv = 0u;
    };
    auto fill_head = [&](slot_t &s, const double N[10], const double q[3], uint32_t nanmask, uint32_t zeromask, uint32_t unit) {
        s.nan = nanmask; s.zero = zeromask; s.unit = unit;
        if constexpr (KEEPN) {
#pragma unroll
            for (int i = 0; i < 10; ++i) s.N[i] = N[i];
            s.q0[0] = q[0]; s.q0[1] = q[1]; s.q0[2] = q[2];
        } else {
            build_base(s, N, q);
        }
    };
    auto fill_slot = [&](slot_t &s, const double N[10], const double q[3], uint32_t nanmask, uint32_t zeromask, uint32_t unit,
                         const RegObs<T, CT> &obs) {
        fill_head(s, N, q, nanmask, zeromask, unit);
#pragma unroll
        for (int c = 0; c < CT; ++c) { s.o[3 * c] = obs.x[c]; s.o[3 * c + 1] = obs.y[c]; s.o[3 * c + 2] = obs.w[c]; }
    };

    // ---- the search over the pooled slots (and then over the units that found no slot) ---------------------------------
    auto flush = [&]() {
        int cnt = n_used;
        int over_done = 0;
        for (;;) {
            wsync();
            const bool own = lane < cnt;                                    // lane s looks after slot s
            slot_t &mine = slots[own ? lane : 0];
            const uint32_t m_nan = mine.nan, m_d = mine.nan | mine.zero, m_valid = allmask & ~m_d;
            const int m_V = __popc(m_d), m_Lmax = C - a.min_cams - m_V;
            if constexpr (KEEPN) {
                if (own) build_base(mine, mine.N, mine.q0);
            }
            enum { ST_DONE = 0, ST_NEED_A = 1, ST_WAIT_B = 2 };
            int state = own ? ST_NEED_A : ST_DONE, mylevel = 1;
            st_units += (uint32_t)cnt;
            wsync();

            // Wave-uniform scheduler: screen rounds level by level; an fp64 pass whenever 64 survivors wait, and to drain
            // the list once no unit can be screened further; then the owners read their level's verdict.
            enum { PH_NEXT_LEVEL, PH_ROUND, PH_DRAIN };
            int phase = PH_NEXT_LEVEL, level = 0, npend = 0, lg = 2, p0 = 0;
            uint32_t r0 = 0, sub0 = 0, nsub = 0, nSurv = 0;
            unsigned long long levmask = 0ull;
            float run_hi = kInfF;
            for (;;) {
                while (nSurv >= 64u || (phase == PH_DRAIN && nSurv > 0u)) {
                    // ---- tier B: fp64 evaluation of up to 64 survivors, one per lane --------------------------------------
                    wsync();
                    const uint32_t n = min(64u, nSurv);
                    const bool go = (uint32_t)lane < n;
                    const uint2 ent = sSurv[go ? lane : 0];
                    slot_t &s = slots[ent.x & 0xffu];
                    const uint32_t S = ent.x >> 8, r = ent.y;
                    const uint32_t o_d = s.nan | s.zero, o_valid = allmask & ~o_d;
                    const uint32_t Rreal = S & o_valid, kept = o_valid & ~Rreal;
                    const int nkept = __popc(kept);
                    SlotObs<T> sobs{s.o, a.lik_thr};
                    // the projection matrices are scalar loads at their point of use: an opaque copy of the pointer keeps the
                    // compiler from hoisting 192 SGPRs' worth of them out of the loops
                    cam_cptr cams_here = cams;
                    asm volatile("" : "+s"(cams_here));
                    double Ns[10];
                    if constexpr (KEEPN) {
                        // level 0's normal matrix less the subset's cameras
#pragma unroll
                        for (int i = 0; i < 10; ++i) Ns[i] = s.N[i];
                        for (uint32_t rr = go ? Rreal : 0u; __any(rr != 0u); rr &= rr - 1) {
                            const bool on = rr != 0u;
                            const int c = on ? __builtin_ctz(rr) : 0;
                            const T x = s.o[3 * c], y = s.o[3 * c + 1], w = s.o[3 * c + 2];
                            accum_camera<-1>(Ns, sP + c * 12, (double)x, (double)y, on ? (double)w : 0.0);
                        }
                    } else {
                        // normal matrix of the kept cameras, as the reference builds it (:469-476) and as level 0 did
#pragma unroll
                        for (int i = 0; i < 10; ++i) Ns[i] = 0.0;
#pragma unroll
                        for (int c = 0; c < CT; ++c) {
                            if (EXACT || c < C) {
                                const T x = s.o[3 * c], y = s.o[3 * c + 1], w = s.o[3 * c + 2];
                                accum_camera<1>(Ns, cams_here[c].P, (double)x, (double)y, ((kept >> c) & 1u) ? (double)w : 0.0);
                            }
                        }
                    }
                    double q[3];
                    smallest_eigvec(Ns, q);
                    if (nkept < 2) { q[0] = d_nan(); q[1] = d_nan(); q[2] = d_nan(); }
                    asm volatile("" : "+s"(cams_here));
                    const double e = mean_error<T, false, CT>(cams_here, C, sobs, kept, q);
                    // argmin per unit on (error, rank), np.nanargmin (triangulation.py:500-503): the bits of a non-negative
                    // double order like the value, inf above every number and NaN above inf, so a NaN only wins among NaNs
                    const unsigned long long bits = (e == e) ? (unsigned long long)__double_as_longlong(e) : 0x7ff8000000000000ull;
                    // (a unit's survivors may come in several passes: a pass that lowers the error starts the rank afresh)
                    const unsigned long long before = s.ebits;
                    wsync();
                    if (go) atomicMin(&s.ebits, bits);
                    wsync();
                    const bool first = go && (s.ebits == bits);
                    if (first && bits < before) s.rank = kNone;
                    wsync();
                    if (first) atomicMin(&s.rank, r);
                    wsync();
                    __builtin_amdgcn_s_waitcnt(0);                         // (an earlier pass's points have been written)
                    if (first && s.rank == r) {
                        // the unit's best so far: its point goes straight over what the tile stored (the owner puts NaN there
                        // again should the unit fail in the end); stores of one wave to one address arrive in program order
                        s.S = S;
                        double *Qu = a.Q + (a.block0 * K + (int64_t)s.unit) * 3;
                        Qu[0] = q[0]; Qu[1] = q[1]; Qu[2] = q[2];
                    }
                    st_evals += n; ++st_passes;
                    // what is left of the list moves to its head (at most 127 entries)
                    const uint32_t rem = nSurv - n;
                    const uint2 mv0 = sSurv[((uint32_t)lane < rem) ? 64 + lane : 0];
                    const uint2 mv1 = sSurv[((uint32_t)lane + 64u < rem) ? 128 + lane : 0];
                    wsync();
                    if ((uint32_t)lane < rem) sSurv[lane] = mv0;
                    if ((uint32_t)lane + 64u < rem) sSurv[64 + lane] = mv1;
                    nSurv = rem;
                    wsync();
                }
                if (phase == PH_DRAIN) {
                    wsync();
                    if (state == ST_WAIT_B) {
                        const double e = __longlong_as_double((long long)mine.ebits);
                        if (!(e <= thr) && (e > thr) && (mylevel + 1 <= m_Lmax)) {          // :408: on to the next level
                            ++mylevel; state = ST_NEED_A;
                            mine.ebits = ~0ull; mine.rank = kNone; mine.surv = 0u;
                        } else {
                            state = ST_DONE;
                        }
                    }
                    if (!__any(state == ST_NEED_A)) break;
                    wsync();
                    phase = PH_NEXT_LEVEL; level = 0;
                    continue;
                }
                if (phase == PH_NEXT_LEVEL) {
                    if (!__any(state == ST_NEED_A)) { phase = PH_DRAIN; continue; }
                    ++level;
                    const bool at = (state == ST_NEED_A) && (mylevel == level);
                    const unsigned long long pend = __ballot(at);
                    if (pend == 0ull) continue;
                    wsync();
                    npend = __popcll(pend);
                    if (at) sList[__popcll(pend & lt)] = (uint32_t)lane;
                    wsync();
                    sub0 = a.sub_off[level];
                    nsub = a.sub_off[level + 1] - sub0;
                    {
                        uint32_t best_cost = 0xffffffffu;
                        for (int l = 1; l <= 6; ++l) {                              // 2^l lanes per unit, two subsets per lane
                            const uint32_t passes = (((uint32_t)npend << l) + 63u) >> 6;
                            const uint32_t rounds = (nsub + (2u << l) - 1u) >> (l + 1);
                            const uint32_t cost = passes * rounds;
                            if (cost <= best_cost) { best_cost = cost; lg = l; }
                        }
                    }
                    p0 = 0; r0 = 0; run_hi = kInfF; levmask = pend;
                    phase = PH_ROUND;
                    continue;
                }
                // ---- tier A: one round of the screen; lane j of a group looks at subsets #(r0 + j) and #(r0 + G + j) of its
                // unit's level, one in each half of its packed registers
                {
                    const int G = 1 << lg, groups = 64 >> lg;
                    const int grp = lane >> lg, lig = lane & (G - 1);
                    const bool has = p0 + grp < npend;
                    const uint32_t si = sList[has ? p0 + grp : p0];
                    slot_t &s = slots[si];
                    const uint32_t o_d = s.nan | s.zero, o_valid = allmask & ~o_d;
                    const bool last = level >= C - a.min_cams - __popc(o_d);         // the unit's last level: its minimum counts whatever it is
                    const uint32_t rA = r0 + (uint32_t)lig, rB = rA + (uint32_t)G;
                    auto subset_of = [&](const uint32_t r, bool &go) -> uint32_t {
                        uint32_t S = 0;
                        go = has && (r < nsub);
                        if (go) {
                            // level 1 is rank r <-> camera r (itertools.combinations order); the deeper levels go to the table
                            S = (level == 1) ? (1u << r) : (uint32_t)a.sub_tab[sub0 + r];
                            // quirk Q1 duplicates: only the lexicographically first padding can win the argmin
                            const uint32_t pad = S & o_d;
                            const uint32_t below = pad ? ((2u << (31 - __builtin_clz(pad))) - 1u) : 0u;
                            go = (o_d & below) == pad;
                        }
                        return S;
                    };
                    bool goA, goB;
                    const uint32_t SA = subset_of(rA, goA), SB = subset_of(rB, goB);
                    if (__any(goA || goB)) {
                        const uint32_t RA = SA & o_valid, RB = SB & o_valid;
                        const uint32_t keptA = o_valid & ~RA, keptB = o_valid & ~RB;
                        const int nkA = __popc(keptA), nkB = __popc(keptB);
                        float loA = -kInfF, hiA = kInfF, loB = -kInfF, hiB = kInfF;
                        if (screen_on) {
                            const float c0 = s.c0[0], c1 = s.c0[1], c2 = s.c0[2];
                            // the removed cameras leave the normal matrix: each half walks its own subset
                            struct Base { float m00, m01, m02, m11, m12, m22, g0, g1, g2, h; };
                            const Base base{s.M[0], s.M[1], s.M[2], s.M[3], s.M[4], s.M[5], s.g[0], s.g[1], s.g[2], s.h};
                            auto downdate = [&](const bool go, const uint32_t Rreal) -> Base {
                                Base b = base;
                                for (uint32_t rr = go ? Rreal : 0u; __any(rr != 0u); rr &= rr - 1) {
                                    // (a lane with nothing left to remove goes through the motions on one of its unit's valid
                                    // cameras with weight 0: finite numbers, loads without branches)
                                    const bool on = rr != 0u;
                                    const int c = __builtin_ctz(on ? rr : o_valid);
                                    const float x = (float)s.o[3 * c], y = (float)s.o[3 * c + 1];
                                    const float w = on ? (float)s.o[3 * c + 2] : 0.0f;
                                    const float *P = sPf + c * 12;
                                    const float A0 = fmaf(-x, P[8], P[0]), A1 = fmaf(-x, P[9], P[1]), A2 = fmaf(-x, P[10], P[2]), A3 = fmaf(-x, P[11], P[3]);
                                    const float B0 = fmaf(-y, P[8], P[4]), B1 = fmaf(-y, P[9], P[5]), B2 = fmaf(-y, P[10], P[6]), B3 = fmaf(-y, P[11], P[7]);
                                    const float u = fmaf(A0, c0, fmaf(A1, c1, fmaf(A2, c2, A3)));
                                    const float v = fmaf(B0, c0, fmaf(B1, c1, fmaf(B2, c2, B3)));
                                    const float w2 = w * w;
                                    const float a0 = A0 * w2, a1 = A1 * w2, a2 = A2 * w2, b0 = B0 * w2, b1 = B1 * w2, b2 = B2 * w2;
                                    b.m00 = fmaf(-a0, A0, fmaf(-b0, B0, b.m00)); b.m01 = fmaf(-a0, A1, fmaf(-b0, B1, b.m01));
                                    b.m02 = fmaf(-a0, A2, fmaf(-b0, B2, b.m02)); b.m11 = fmaf(-a1, A1, fmaf(-b1, B1, b.m11));
                                    b.m12 = fmaf(-a1, A2, fmaf(-b1, B2, b.m12)); b.m22 = fmaf(-a2, A2, fmaf(-b2, B2, b.m22));
                                    b.g0 = fmaf(-a0, u, fmaf(-b0, v, b.g0)); b.g1 = fmaf(-a1, u, fmaf(-b1, v, b.g1)); b.g2 = fmaf(-a2, u, fmaf(-b2, v, b.g2));
                                    b.h = fmaf(-w2 * u, u, fmaf(-w2 * v, v, b.h));
                                }
                                return b;
                            };
                            const Base bA = downdate(goA, RA), bB = downdate(goB, RB);
                            const Sym3v M{v2f{bA.m00, bB.m00}, v2f{bA.m01, bB.m01}, v2f{bA.m02, bB.m02}, v2f{bA.m11, bB.m11}, v2f{bA.m12, bB.m12}, v2f{bA.m22, bB.m22}};
                            const v2f g0{bA.g0, bB.g0}, g1{bA.g1, bB.g1}, g2{bA.g2, bB.g2}, h{bA.h, bB.h};
                            // smallest eigenpair of the pencil: (M - lam) q = lam c0 - g, lam = Rayleigh quotient; from
                            // lam = 0 (the inhomogeneous least-squares point) two steps leave |dlam / lam| ~ 1e-4 or less
                            const v2f cc0 = splat2(c0), cc1 = splat2(c1), cc2 = splat2(c2);
                            v2f q0, q1, q2, det, tr, a00, c22;
                            solve3v(M, splat2(0.0f), -g0, -g1, -g2, q0, q1, q2, det, tr, a00, c22);
                            const v2f lam1 = rayleighv(M, g0, g1, g2, h, cc0, cc1, cc2, q0, q1, q2);
                            solve3v(M, lam1, fma2(lam1, cc0, -g0), fma2(lam1, cc1, -g1), fma2(lam1, cc2, -g2), q0, q1, q2, det, tr, a00, c22);
                            const v2f lam2 = rayleighv(M, g0, g1, g2, h, cc0, cc1, cc2, q0, q1, q2);
                            solve3v(M, lam2, fma2(lam2, cc0, -g0), fma2(lam2, cc1, -g1), fma2(lam2, cc2, -g2), q0, q1, q2, det, tr, a00, c22);
                            const v2f dl = (lam2 - lam1) * rcp2(lam2);
                            const float dlamA = fabsf(dl.x), dlamB = fabsf(dl.y);
                            const v2f Q0 = cc0 + q0, Q1 = cc1 + q1, Q2 = cc2 + q2;
                            // mean reprojection error of the kept cameras, |(a/z - x, b/z - y)| = s / sqrt(s z^2); a camera that is
                            // not kept adds +0 whatever it computes, a kept one that is degenerate leaves NaN or inf in the sum
                            v2f sum = splat2(0.0f);
                            cam_cptr cams_here = cams;
                            asm volatile("" : "+s"(cams_here));
                            for_each_cam<CT>(C, [&](int c) {
                                const __attribute__((address_space(4))) float *P = cams_here[c].Pf;
                                const float x = (float)s.o[3 * c], y = (float)s.o[3 * c + 1];
                                const v2f pa = fma2(splat2(P[0]), Q0, fma2(splat2(P[1]), Q1, splat2(P[2]) * Q2)) + splat2(P[3]);
                                const v2f pb = fma2(splat2(P[4]), Q0, fma2(splat2(P[5]), Q1, splat2(P[6]) * Q2)) + splat2(P[7]);
                                const v2f pz = fma2(splat2(P[8]), Q0, fma2(splat2(P[9]), Q1, splat2(P[10]) * Q2)) + splat2(P[11]);
                                const v2f du = fma2(splat2(-x), pz, pa), dv = fma2(splat2(-y), pz, pb);
                                const v2f ss = fma2(du, du, dv * dv);
                                const v2f d = ss * rsq2(ss * pz * pz);
                                const uint32_t mA = 0u - ((keptA >> c) & 1u), mB = 0u - ((keptB >> c) & 1u);
                                sum.x += __uint_as_float(__float_as_uint(d.x) & mA);
                                sum.y += __uint_as_float(__float_as_uint(d.y) & mB);
                            });
                            const v2f e32 = sum * rcp2(v2f{(float)nkA, (float)nkB});
                            const v2f cond_lim = splat2(kCondMin) * tr * tr * tr;
                            const v2f marg = fma2(e32, fma2(splat2(kMargDlam), v2f{dlamA, dlamB}, splat2(kMargRel)), splat2(kMargAbs));
                            // (every comparison is false for a NaN: anything that went wrong above leaves the candidate unvouched)
                            const bool gA = (a00.x > 0.0f) && (c22.x > 0.0f) && (det.x > 0.0f) && (nkA >= 2) && (det.x >= cond_lim.x) &&
                                            (dlamA <= kDlamMax) && (e32.x < kInfF);
                            const bool gB = (a00.y > 0.0f) && (c22.y > 0.0f) && (det.y > 0.0f) && (nkB >= 2) && (det.y >= cond_lim.y) &&
                                            (dlamB <= kDlamMax) && (e32.y < kInfF);
                            loA = gA ? e32.x - marg.x : -kInfF; hiA = gA ? e32.x + marg.x : kInfF;
                            loB = gB ? e32.y - marg.y : -kInfF; hiB = gB ? e32.y + marg.y : kInfF;
                        }
                        if (!goA) { loA = kInfF; hiA = kInfF; }
                        if (!goB) { loB = kInfF; hiB = kInfF; }
                        float gmin = fminf(hiA, hiB);
                        for (int off = G >> 1; off > 0; off >>= 1) gmin = fminf(gmin, __shfl_xor(gmin, off, 64));
                        run_hi = fminf(run_hi, gmin);
                        // the true minimum w of the level has e_lo(w) <= e64(w) <= e64(S) <= e_hi(S) for every S seen so far
                        const bool survA = goA && (loA <= run_hi) && (last || loA <= thr_f);
                        const bool survB = goB && (loB <= run_hi) && (last || loB <= thr_f);
                        const unsigned long long sbA = __ballot(survA), sbB = __ballot(survB);
                        if ((sbA | sbB) != 0ull) {
                            const uint32_t nA = (uint32_t)__popcll(sbA);
                            if (survA) sSurv[nSurv + (uint32_t)__popcll(sbA & lt)] = make_uint2(si | (SA << 8), rA);
                            if (survB) sSurv[nSurv + nA + (uint32_t)__popcll(sbB & lt)] = make_uint2(si | (SB << 8), rB);
                            if (survA || survB) s.surv = 1u;
                            nSurv += nA + (uint32_t)__popcll(sbB);
                        }
                        st_screened += (uint32_t)(__popcll(__ballot(goA)) + __popcll(__ballot(goB))); ++st_spasses;
                    }
                    r0 += (uint32_t)(2 * G);
                    if (r0 >= nsub) { r0 = 0; p0 += groups; run_hi = kInfF; }
                    if (p0 >= npend) {
                        // the level's screen is through: a unit nobody survived for has certainly failed the level and,
                        // unless it was its last (where somebody always survives), goes on to the next one at once
                        wsync();
                        if ((levmask >> lane) & 1ull) {
                            if (mine.surv != 0u || level >= m_Lmax) state = ST_WAIT_B;
                            else mylevel = level + 1;
                        }
                        phase = PH_NEXT_LEVEL;
                    }
                }
            }

            // ---- results of the searched units (triangulation.py:588-604), patched over what their tile stored ------------
            __builtin_amdgcn_s_waitcnt(0);
            if (own) {
                const double e = __longlong_as_double((long long)mine.ebits);
                const uint32_t bS = mine.S;
                const bool ok = e <= thr;
                const int64_t gu = a.block0 * K + (int64_t)mine.unit;
                if (!ok) { a.Q[gu * 3 + 0] = d_nan(); a.Q[gu * 3 + 1] = d_nan(); a.Q[gu * 3 + 2] = d_nan(); }
                a.err[gu] = ok ? (float)e : __builtin_nanf("");
                a.mask[gu] = m_nan | bS;
                a.n_excl[gu] = (uint8_t)(m_V + __popc(bS & m_valid));          // :436 counts NaN or zero
            }
            // ---- units that found no slot: read their observations again, rebuild level 0, search them the same way -------
            if (n_over - over_done <= 0) break;
            cnt = min(NSLOT, n_over - over_done);
            wsync();
            if (lane < cnt) {
                // (opaque copies: nothing of this rare path is to be prepared ahead and held in registers through the search)
                uint32_t Kq = (uint32_t)K, ln = (uint32_t)lane;
                asm volatile("" : "+s"(Kq), "+v"(ln));
                const uint32_t o = sOver[(uint32_t)over_done + ln];
                const uint32_t u = ((tile0 + (o >> 6) * tstride) << 6) + (o & 63u);
                const uint32_t b = u / Kq, k = u - b * Kq;
                RegObs<T, CT> ob;
                ob.lik_thr = a.lik_thr;
                load_obs<T, CT, EXACT>(a, C, b, k, ob);
                double N2[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) N2[i] = 0.0;
                uint32_t nan2 = 0, zero2 = 0;
                classify_and_accumulate<T, CT>(cams, C, ob, N2, nan2, zero2);
                double q2[3];
                smallest_eigvec(N2, q2);
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    if (((nan2 | zero2) >> c) & 1u) { ob.x[c] = (T)0; ob.y[c] = (T)0; ob.w[c] = (T)0; }
                }
                fill_slot(slots[lane], N2, q2, nan2, zero2, u, ob);
            }
            over_done += cnt;
        }
        n_used = 0; n_over = 0;
    };

    // ---- a tile's results leave at once, 16-byte stores of contiguous memory; the searching units' results are patched
    // over them after the search
    auto store_tile = [&](const uint32_t tile) {
        {
            const int64_t wave_u0 = (int64_t)tile << 6;
            const int64_t gu0 = a.block0 * K + wave_u0;
            const int64_t n_left = n_units - wave_u0;
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
                dst[lane] = src[lane];
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
                if (lane < n_left) { Ew[lane] = __uint_as_float(sE[lane]); Mw[lane] = sM[lane]; Xw[lane] = sX[lane]; }
            }
        }
    };

    // ---- level 0 of the wave's tiles (triangulation.py:404-505 with nb_cams_off = 0), one after the other ---------------
    // `prefetch` is called once the eigen-solve is through (the point of highest register pressure): a tile requests the
    // next tile's observations there, so that they travel during its reprojection pass instead of after it.
    auto level0 = [&](const int t, const bool act, const RegObs<T, CT> &cur, auto &&prefetch) {
        const uint32_t tile = tile0 + (uint32_t)t * tstride;
        double N[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) N[i] = 0.0;
        uint32_t nanmask = 0, zeromask = 0;
        // classify_and_accumulate / mean_error of p2s_tri_dev.h with two instruction-count savings (same arithmetic, same
        // bits): a camera that does not count is zeroed once, in single precision, and everything after -- normal matrix,
        // reprojection pass (its distance is masked anyway), the unit's slot -- reads the zeroed copy (3 selects per camera
        // instead of 6 on the converted values); a degenerate camera shows as a NaN in the sum instead of being looked for
        // camera by camera.
        RegObs<T, CT> obs;
        obs.lik_thr = a.lik_thr;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const T x = cur.x[c], y = cur.y[c], w = cur.w[c];
            const bool isn = (EXACT || c < C) && (!(w == w) || (w < lik_t) || !finite_xy(x, y));   // (see classify_and_accumulate)
            const bool isz = (EXACT || c < C) && (w == (T)0) && !isn;
            nanmask |= isn ? (1u << c) : 0u;
            zeromask |= isz ? (1u << c) : 0u;
            // (the bits are set here: left to itself the compiler keeps the 16 comparison results in scalar register pairs,
            // spills those to lanes of a vector register and assembles the masks after the eigen-solve: 18 instructions per
            // camera instead of 9)
            asm volatile("" : "+v"(nanmask), "+v"(zeromask));
            const bool okc = (EXACT || c < C) && !(isn || isz);
            obs.x[c] = okc ? x : (T)0; obs.y[c] = okc ? y : (T)0; obs.w[c] = okc ? w : (T)0;
            if (EXACT || c < C) accum_camera<1>(N, cams[c].P, (double)obs.x[c], (double)obs.y[c], (double)obs.w[c]);
        }
        const uint32_t dmask = nanmask | zeromask;
        const uint32_t valid = allmask & ~dmask;
        const int V = __popc(dmask);
        const int Lmax = act ? C - a.min_cams - V : -1;                   // last level that runs (:408, :437-441)
        double q[3];
        smallest_eigvec(N, q);
        prefetch(q[0]);
        // (the reprojection pass converts x and y again: kept as doubles through the eigen-solve they are 16 registers more)
#pragma unroll
        for (int c = 0; c < CT; ++c) asm volatile("" : "+v"(obs.x[c]), "+v"(obs.y[c]));
        if (C - V < 2) { q[0] = d_nan(); q[1] = d_nan(); q[2] = d_nan(); }  // common.py:347: fewer than 4 rows
        double e;
        {
            double sum = 0.0;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                if (EXACT || c < C) {
                    bool reg;
                    const double d = camera_distance<false>(cams + c, q, (double)obs.x[c], (double)obs.y[c], reg);
                    sum += ((valid >> c) & 1u) ? d : 0.0;
                }
            }
            // a wanted camera with s z^2 not in (0, inf) left a NaN behind (its reciprocal square root): such units take the
            // literal formula with the reference's NaN rules (all-NaN -> inf, nansum)
            const bool irregular = sum != sum;
            if (__any(irregular)) {
                double sum2 = 0.0;
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    if (EXACT || c < C) {
                        const double d = camera_distance_exact(cams + c, q[0], q[1], q[2], (double)obs.x[c], (double)obs.y[c]);
                        sum2 += ((valid >> c) & 1u) ? d : 0.0;
                    }
                }
                sum = irregular ? sum2 : sum;
            }
            e = sum * fast_rcp((double)__popc(valid));                     // no camera kept -> NaN, as np.mean([])
        }
        const bool ran = Lmax >= 0;                                        // else no level completes (:595-596)
        const bool ok = ran && (e <= thr);                                 // :600-602
        if (t > 0) wsync();                                                // the previous tile's staged results have left
        sQ[lane * 3 + 0] = ok ? q[0] : d_nan();
        sQ[lane * 3 + 1] = ok ? q[1] : d_nan();
        sQ[lane * 3 + 2] = ok ? q[2] : d_nan();
        sE[lane] = __float_as_uint(ok ? (float)e : __builtin_nanf(""));
        sM[lane] = ran ? nanmask : allmask;
        sX[lane] = (uint8_t)(ran ? V : C);
        // goes on to level 1 (then at least 3 cameras are valid: q is the solve's own point, the centre of the screen)
        const bool need = (Lmax >= 1) && (e > thr);
        const unsigned long long hard = __ballot(need);
        if (need) {
            const int ord = n_used + n_over + __popcll(hard & lt);
            if (ord < NSLOT) fill_slot(slots[ord], N, q, nanmask, zeromask, (tile << 6) + (uint32_t)lane, obs);
            else sOver[ord - NSLOT] = (uint16_t)(t * 64 + lane);
        }
        {
            const int tot = n_used + n_over + __popcll(hard);
            n_over = max(0, tot - NSLOT);
            n_used = min(NSLOT, tot);
        }
        wsync();
        store_tile(tile);
    };
    // 9-16 cameras: the same steps with the observations taken eight cameras at a time -- read for the normal matrix, read
    // again (from L2) for the reprojection error, and a third time by the few lanes that park their unit in a slot -- so
    // that no more than 48 registers of observations are ever live, and none during the eigen-solve.
    auto level0_wide = [&](const int t) {
        const uint32_t tile = tile0 + (uint32_t)t * tstride;
        bool act;
        const uint32_t u = unit_of(tile, act);
        const uint32_t ub = u / (uint32_t)K, uk = u % (uint32_t)K;
        RegObs<T, 8> h0, h1;
        h0.lik_thr = a.lik_thr; h1.lik_thr = a.lik_thr;
        load_half<T, EXACT, true, P2S_POOL_WIDE_STREAM != 0>(a, C, ub, uk, 0, h0);
        load_half<T, EXACT, true, P2S_POOL_WIDE_STREAM != 0>(a, C, ub, uk, 8, h1);
        double N[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) N[i] = 0.0;
        uint32_t nanmask = 0, zeromask = 0;
        auto accumulate = [&](const int c0, const RegObs<T, 8> &h) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (EXACT || c0 + i < C) {
                    const T x = h.x[i], y = h.y[i], w = h.w[i];
                    const bool isn = !(w == w) || (w < lik_t) || !finite_xy(x, y);
                    const bool isz = (w == (T)0) && !isn;
                    nanmask |= isn ? (1u << (c0 + i)) : 0u;
                    zeromask |= isz ? (1u << (c0 + i)) : 0u;
                    asm volatile("" : "+v"(nanmask), "+v"(zeromask));
                    const bool ok = !(isn || isz);
                    accum_camera<1>(N, cams[c0 + i].P, (double)(ok ? x : (T)0), (double)(ok ? y : (T)0), (double)(ok ? w : (T)0));
                }
            }
        };
        accumulate(0, h0);
        accumulate(8, h1);
        const uint32_t dmask = nanmask | zeromask;
        const uint32_t valid = allmask & ~dmask;
        const int V = __popc(dmask);
        const int Lmax = act ? C - a.min_cams - V : -1;
        double q[3];
        smallest_eigvec(N, q);
        uint32_t ub2 = ub;
#if P2S_POOL_WIDE_REREAD
        asm volatile("" : "+v"(ub2) : "v"(q[0]));                          // the second read not before the eigen-solve
        load_half<T, EXACT, false>(a, C, ub2, uk, 0, h0);                  // x and y only
#endif
        if (C - V < 2) { q[0] = d_nan(); q[1] = d_nan(); q[2] = d_nan(); }
        double sum = 0.0;
        bool irregular = false;
        auto distances = [&](const int c0, const RegObs<T, 8> &h) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (EXACT || c0 + i < C) {
                    const bool kk = (valid >> (c0 + i)) & 1u;
                    bool reg;
                    const double d = camera_distance<false>(cams + (c0 + i), q, (double)h.x[i], (double)h.y[i], reg);
                    irregular = irregular || (kk && !reg);
                    sum += kk ? d : 0.0;
                }
            }
        };
        distances(0, h0);
#if P2S_POOL_WIDE_REREAD
        asm volatile("" : "+v"(ub2));                                      // (addresses are recomputed, not kept: 2 registers per camera)
        load_half<T, EXACT, false>(a, C, ub2, uk, 8, h0);
        distances(8, h0);
#else
        distances(8, h1);
#endif
        if (__any(irregular)) {                                            // rare: some wanted camera is degenerate / NaN
            double sum2 = 0.0;
            auto exact = [&](const int c0, const RegObs<T, 8> &h) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (EXACT || c0 + i < C) {
                        const double d = camera_distance_exact(cams + (c0 + i), q[0], q[1], q[2], (double)h.x[i], (double)h.y[i]);
                        sum2 += ((valid >> (c0 + i)) & 1u) ? d : 0.0;
                    }
                }
            };
#if P2S_POOL_WIDE_REREAD
            load_half<T, EXACT, false>(a, C, ub2, uk, 0, h0);
            exact(0, h0);
            load_half<T, EXACT, false>(a, C, ub2, uk, 8, h0);
            exact(8, h0);
#else
            exact(0, h0);
            exact(8, h1);
#endif
            sum = irregular ? sum2 : sum;
        }
        const double e = sum * fast_rcp((double)__popc(valid));
        const bool ran = Lmax >= 0;
        const bool ok = ran && (e <= thr);
        if (t > 0) wsync();
        sQ[lane * 3 + 0] = ok ? q[0] : d_nan();
        sQ[lane * 3 + 1] = ok ? q[1] : d_nan();
        sQ[lane * 3 + 2] = ok ? q[2] : d_nan();
        sE[lane] = __float_as_uint(ok ? (float)e : __builtin_nanf(""));
        sM[lane] = ran ? nanmask : allmask;
        sX[lane] = (uint8_t)(ran ? V : C);
        const bool need = (Lmax >= 1) && (e > thr);
        const unsigned long long hard = __ballot(need);
        if (hard != 0ull) {
            const int ord = n_used + n_over + __popcll(hard & lt);
            const bool slotted = need && ord < NSLOT;
            if (need && !slotted) sOver[ord - NSLOT] = (uint16_t)(t * 64 + lane);
            slot_t &s = slots[slotted ? ord : 0];
            if (slotted) fill_head(s, N, q, nanmask, zeromask, (tile << 6) + (uint32_t)lane);
            // another read, by the lanes that park a unit only (they need the likelihoods again)
            if (slotted) {
                asm volatile("" : "+v"(ub2));
                load_half<T, EXACT>(a, C, ub2, uk, 0, h0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bool okc = (valid >> i) & 1u;
                    s.o[3 * i] = okc ? h0.x[i] : (T)0; s.o[3 * i + 1] = okc ? h0.y[i] : (T)0; s.o[3 * i + 2] = okc ? h0.w[i] : (T)0;
                }
                asm volatile("" : "+v"(ub2));
                load_half<T, EXACT>(a, C, ub2, uk, 8, h0);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const bool okc = (valid >> (8 + i)) & 1u;
                    s.o[24 + 3 * i] = okc ? h0.x[i] : (T)0; s.o[24 + 3 * i + 1] = okc ? h0.y[i] : (T)0;
                    s.o[24 + 3 * i + 2] = okc ? h0.w[i] : (T)0;
                }
            }
            const int tot = n_used + n_over + __popcll(hard);
            n_over = max(0, tot - NSLOT);
            n_used = min(NSLOT, tot);
        }
        wsync();
        store_tile(tile);
    };

    RegObs<T, CT> obs0, obs1, obs2, obs3, obs4, obs5;
    obs0.lik_thr = a.lik_thr; obs1.lik_thr = a.lik_thr; obs2.lik_thr = a.lik_thr; obs3.lik_thr = a.lik_thr;
    obs4.lik_thr = a.lik_thr; obs5.lik_thr = a.lik_thr;
    bool act0, act1 = false, act2 = false, act3 = false, act4 = false, act5 = false;
    if constexpr (CT <= 8) {
        const uint32_t u0 = unit_of(tile0, act0);
        load_obs<T, CT, EXACT>(a, C, u0 / (uint32_t)K, u0 % (uint32_t)K, obs0);
    } else {
        act0 = false;
    }
    const bool two = my_tiles > 1 && tile0 + tstride < n_tiles;
    const bool three = TPW > 2 && my_tiles > 2 && tile0 + 2 * tstride < n_tiles;
    const bool four = TPW > 3 && my_tiles > 3 && tile0 + 3 * tstride < n_tiles;
    const bool five = TPW > 4 && my_tiles > 4 && tile0 + 4 * tstride < n_tiles;
    const bool six = TPW > 5 && my_tiles > 5 && tile0 + 5 * tstride < n_tiles;
    auto request = [&](const bool wanted, const int t, bool &act, RegObs<T, CT> &obs, double dep) {
        if (wanted) {
            uint32_t u = unit_of(tile0 + (uint32_t)t * tstride, act);
            asm volatile("" : "+v"(u) : "v"(dep));                         // not before the eigen-solve
            load_obs<T, CT, EXACT>(a, C, u / (uint32_t)K, u % (uint32_t)K, obs);
        }
    };
    if constexpr (CT > 8) {
        level0_wide(0);
        if constexpr (TPW > 1) { if (two) level0_wide(1); }
        if constexpr (TPW > 2) { if (three) level0_wide(2); }
        if constexpr (TPW > 3) { if (four) level0_wide(3); }
    } else {
    level0(0, act0, obs0, [&](double dep) { if constexpr (TPW > 1) request(two, 1, act1, obs1, dep); });
    if constexpr (TPW > 1) {
        if (two) level0(1, act1, obs1, [&](double dep) { if constexpr (TPW > 2) request(three, 2, act2, obs2, dep); });
    }
    if constexpr (TPW > 2) {
        if (three) level0(2, act2, obs2, [&](double dep) { if constexpr (TPW > 3) request(four, 3, act3, obs3, dep); });
    }
    if constexpr (TPW > 3) {
        if (four) level0(3, act3, obs3, [&](double dep) { if constexpr (TPW > 4) request(five, 4, act4, obs4, dep); });
    }
    if constexpr (TPW > 4) {
        if (five) level0(4, act4, obs4, [&](double dep) { if constexpr (TPW > 5) request(six, 5, act5, obs5, dep); });
    }
    if constexpr (TPW > 5) {
        if (six) level0(5, act5, obs5, [](double) {});
    }
    }

    // ---- camera-subset search over the pooled units; their results are patched over what their tiles stored --------------
    if (n_used > 0) {
        __builtin_amdgcn_s_waitcnt(0);                                     // the tiles' stores have been written
        flush();
    }

    if (a.stats && lane == 0) {
        unsigned long long *st = a.stats + (size_t)(blockIdx.x % P2S_STAT_SHARDS) * P2S_STAT_STRIDE;
        if (st_units) atomicAdd(st + 0, (unsigned long long)st_units);
        if (st_evals) atomicAdd(st + 1, (unsigned long long)st_evals);
        if (st_passes) atomicAdd(st + 2, (unsigned long long)st_passes);
        if (st_screened) atomicAdd(st + 6, (unsigned long long)st_screened);
        if (st_spasses) atomicAdd(st + 7, (unsigned long long)st_spasses);
    }
}

}  // namespace

namespace {

template <typename T, int CT, int NSLOT, int TPW>
hipError_t launch_pool(P2sTriArgs a, int singles_pct, hipStream_t s) {
    const int64_t n_units = a.n_blocks * a.K;
    const int64_t n_tiles = (n_units + 63) / 64;
    const int64_t per_xcd = (n_tiles + 7) / 8;
    int64_t singles = per_xcd * singles_pct / 100;
    singles += (per_xcd - singles) % TPW;                      // the rest in whole groups of TPW
    a.pool_singles = (uint32_t)singles;
    a.pool_pairs = (uint32_t)((per_xcd - singles) / TPW);
    const unsigned grid = (unsigned)(8 * (a.pool_pairs + a.pool_singles));
    if (a.C == CT)
        hipLaunchKernelGGL((p2s_tri_pool_kernel<T, CT, NSLOT, true, TPW>), dim3(grid), dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL((p2s_tri_pool_kernel<T, CT, NSLOT, false, TPW>), dim3(grid), dim3(64), 0, s, a);
    return hipGetLastError();
}

}  // namespace

bool p2s_tri_pool_supports(int C, int dtype, int undistort, int lr_swap) {
    if (undistort || lr_swap) return false;
    return dtype == 0 ? C <= 16 : false;
}

hipError_t p2s_launch_tri_pool(const P2sTriArgs &a, int dtype, int singles_pct, int tiles_per_wave, hipStream_t s) {
    (void)dtype;
    if (a.C > 8) return launch_pool<float, 16, 20, 2>(a, singles_pct, s);
    if (a.C <= 4) return launch_pool<float, 4, 48, 5>(a, singles_pct, s);
    if (tiles_per_wave == 2) return launch_pool<float, 8, 32, 2>(a, singles_pct, s);
    if (tiles_per_wave == 3) return launch_pool<float, 8, 32, 3>(a, singles_pct, s);
    if (tiles_per_wave == 4) return launch_pool<float, 8, 40, 4>(a, singles_pct, s);
    if (tiles_per_wave == 6) return launch_pool<float, 8, 52, 6>(a, singles_pct, s);
    return launch_pool<float, 8, 48, 5>(a, singles_pct, s);
}
