// p2s_tri_deep.hip -- deep levels of the camera-subset search, spread over the whole GPU.
//
// With many cameras a single unit can reach a level of millions of subsets (C(32, 8) = 10.5 M, C(32, 10) = 64.5 M).
// The work-list search kernel (p2s_tri.hip) walks a unit's level inside ONE wave: such a unit keeps that wave busy for
// seconds while the rest of the chip has long finished -- on the 32-camera shard of BASELINE configs[4] the whole step
// took as long as its worst unit (2 - 6 s for ~0.1 s of aggregate arithmetic).  Here a level with more than
// `deep_min_subsets` subsets is cut into chunks of P2S_DEEP_CHUNK consecutive ranks (itertools.combinations order),
// one wave per chunk:
//
//   search kernel   a unit about to enter such a level is exported: its record, level-0 normal matrix, masks and
//                   best-so-far go to the deep list (p2s_tri.hip)
//   plan            one workgroup lays the pending entries' chunks out as tickets, as many entries as fit the
//                   partial-result buffer (the others wait for the next round)
//   eval            persistent waves draw tickets; lane j of a wave evaluates ranks chunk0 + 64 i + j: normal matrix
//                   minus the removed cameras, eigen-solve, reprojection error, L/R-swap candidate -- the arithmetic
//                   of the search kernel -- and the wave's best plain and best swap candidate (lowest rank on ties)
//                   become the chunk's partial result
//   reduce          one wave per entry: argmin over its chunks, then the level's bookkeeping exactly as the search
//                   kernel does it (triangulation.py:500-505, 509-579, 588-604); the unit is finished (results
//                   stored) or moves to its next level
//
// and the host repeats plan / eval / reduce until no entry is pending (one 4-byte read per round).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "p2s_internal.h"
#include "p2s_tri_dev.h"

namespace {

constexpr uint32_t kNone = 0xffffffffu;

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ P2sDeepEntry *entry_at(const P2sDeepArgs &d, uint32_t i) {
    return reinterpret_cast<P2sDeepEntry *>(d.entries + (size_t)i * d.entry_bytes);
}

// ---- plan: tickets for the entries that fit this round.  One workgroup of 16 waves; wave w scans the w-th sixteenth of
// the list 64 entries at a time (exclusive scan of their chunk counts), twice: first for its total, then -- with the
// totals of the waves before it -- for the tickets themselves.  An entry is taken while the tickets up to and including
// its own fit the buffer (the running sum never decreases, so the first entry that does not fit closes the round for
// everything after it; the very first entry always fits: max_tickets >= max_subsets / P2S_DEEP_CHUNK, checked on the host).
__global__ void __launch_bounds__(1024) p2s_deep_plan_kernel(const P2sDeepArgs d, const uint32_t *binom, int C) {
    __shared__ uint32_t sTickets[16], sPending[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t n = min(d.ctl[P2S_DEEP_N_ENTRIES], d.capacity);
    const uint32_t per = ((n + 15u) / 16u + 63u) / 64u * 64u;          // entries per wave, whole rounds of 64
    const uint32_t i_begin = min(n, (uint32_t)wave * per), i_end = min(n, i_begin + per);
    uint32_t base = 0;
    for (int pass = 0; pass < 2; ++pass) {
        uint32_t tickets = base, pending = 0;
        for (uint32_t i0 = i_begin; i0 < i_end; i0 += 64) {
            const uint32_t i = i0 + lane;
            P2sDeepEntry *e = i < i_end ? entry_at(d, i) : nullptr;
            const bool live = e && e->state != P2S_DEEP_DONE;
            pending += (uint32_t)__popcll(__ballot(live));
            const uint32_t chunks = live ? (binom[C * 33 + e->level] + P2S_DEEP_CHUNK - 1) / P2S_DEEP_CHUNK : 0u;
            uint32_t incl = chunks;                                         // inclusive scan over the wave
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = (uint32_t)__shfl_up((int)incl, off, 64);
                if (lane >= off) incl += v;
            }
            const uint32_t first = tickets + incl - chunks;
            if (pass == 1) {
                const bool take = live && (first + chunks <= d.max_tickets || first == 0);
                if (live) e->state = take ? P2S_DEEP_SCHEDULED : P2S_DEEP_WAITING;
                if (take) {
                    e->first_ticket = first;
                    e->n_chunks = chunks;
                    e->pad0 = 0;                            // "a plain candidate of this level is under the threshold" (eval kernel)
                    e->pad1 = 0x7f800000u;                  // the level's best plain error so far, as a float rounded up (eval kernel)
                    for (uint32_t c = 0; c < chunks; ++c) { d.sched_entry[first + c] = i; d.sched_chunk[first + c] = c; }
                }
                // tickets issued = the end of the last entry taken (the taken entries are a prefix of the live ones)
                const unsigned long long taken = __ballot(take);
                if (taken != 0ull && lane == 63 - __builtin_clzll(taken)) atomicMax(d.ctl + P2S_DEEP_N_TICKETS, first + chunks);
            }
            tickets += (uint32_t)__shfl((int)incl, 63, 64);
        }
        if (pass == 0) {
            if (lane == 0) { sTickets[wave] = tickets; sPending[wave] = pending; }
            __syncthreads();
            for (int w = 0; w < wave; ++w) base += sTickets[w];
            if (threadIdx.x == 0) {
                uint32_t pend = 0;
                for (int w = 0; w < 16; ++w) pend += sPending[w];
                d.ctl[P2S_DEEP_TICKET] = 0;
                d.ctl[P2S_DEEP_PENDING] = pend;             // the reduce kernel takes the finished ones off
                d.ctl[P2S_DEEP_N_TICKETS] = 0;              // raised in the second pass (atomicMax of the taken entries' ends)
                __threadfence();
            }
            __syncthreads();
        }
    }
}

// ---- eval: one chunk of one entry's level per ticket ----------------------------------------------------------------------
// LDS per wave: [P: C*12 doubles][binom: 33*33 u32] shared by the workgroup, then per wave the entry's observations.
template <typename T, bool UNDISTORT, bool LRSWAP>
__global__ void __launch_bounds__(256, 3) p2s_deep_eval_kernel(const P2sTriArgs a, const P2sDeepArgs d) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int C = a.C;
    double *sP = reinterpret_cast<double *>(smem);
    uint32_t *sBinom = reinterpret_cast<uint32_t *>(smem + a.lds_binom_off);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    T *sObs = reinterpret_cast<T *>(smem + a.lds_rec_off + (size_t)wave * d.obs_bytes);
    cam_cptr cams = (cam_cptr)a.cams;
    for (int i = tid; i < C * 12; i += blockDim.x) sP[i] = a.cams[i / 12].P[i % 12];
    for (int i = tid; i < 33 * 33; i += blockDim.x) sBinom[i] = a.binom[i];
    __syncthreads();
    const double thr = a.thr;
    const uint32_t allmask = (C == 32) ? 0xffffffffu : ((1u << C) - 1u);
    const uint32_t n_tickets = d.ctl[P2S_DEEP_N_TICKETS];

    for (;;) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(d.ctl + P2S_DEEP_TICKET, 1u);
        t = __shfl(t, 0, 64);
        if (t >= n_tickets) break;
        const P2sDeepEntry *e = entry_at(d, d.sched_entry[t]);
        const uint32_t chunk = d.sched_chunk[t];
        const int level = (int)e->level;
        const uint32_t nsub = sBinom[C * 33 + level];
        const uint32_t r_begin = chunk * P2S_DEEP_CHUNK, r_end = min(nsub, r_begin + P2S_DEEP_CHUNK);
        {   // the entry's observations (as the streaming kernel left them in the record) -> this wave's LDS
            const uint32_t *src = reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(e) + sizeof(P2sDeepEntry));
            uint32_t *dst = reinterpret_cast<uint32_t *>(sObs);
            for (int i = lane; i < (int)(d.obs_bytes >> 2); i += 64) dst[i] = src[i];
        }
        wave_sync();
        UnitObs<T> oobs{sObs, 3, a.lik_thr};
        UnitObs<T> oobs_sw = oobs;
        if (LRSWAP) oobs_sw.p = sObs + C * 3;
        const uint32_t o_nan = e->nanmask, o_d = e->nanmask | e->zeromask, o_valid = allmask & ~o_d;
        const int oV = __popc(o_d);
        const int M = C - oV - level;                       // cameras left when `level` valid ones go (:437, 513)
        double oN[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) oN[i] = e->N[i];

        // ---- exact pruning (d.prune) --------------------------------------------------------------------------------
        // A candidate matters only if its mean error is at most the threshold (a level that is not the unit's last and has
        // no swap candidates: a failed level leaves nothing behind) and not above the best one seen so far.  Every term of the error sum is
        // >= 0, so a candidate whose PARTIAL sum already exceeds bound x (cameras kept) is out -- and the partial sum
        // grows fastest over the cameras with the largest level-0 residuals.  So: (1) the cameras are ranked by that
        // residual once per ticket; (2) the level's subsets are enumerated over the RANKED cameras (rank r of the chunk
        // = r-th combination of ranked positions), which puts the subsets that remove the most suspicious cameras --
        // the only ones with a chance -- at the head of the enumeration and gives every later chunk a tight bound in
        // whole waves of hopeless candidates; (3) the error loop runs over the ranked cameras and stops as soon as every
        // lane of the wave is out.  A wave with a survivor re-evaluates in camera order (mean_error), so that every
        // number that can reach the result is the one the unpruned kernel computes; ties are broken by the subset's
        // rank in itertools order (rank_subset), which the enumeration order no longer provides.
        __shared__ uint8_t sPermAll[4][32];
        uint8_t *sPerm = sPermAll[wave];
        const bool prune = d.prune != 0;
        if (lane < 32) sPerm[lane] = (uint8_t)lane;
        if (prune) {
            double q0[3];
            smallest_eigvec(oN, q0);
            double res = -1.0;                                  // cameras that are out already go last
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
            wave_sync();
            if (lane < C) sPerm[pos] = (uint8_t)lane;
        }
        wave_sync();
        // no further level will replace this one's result: the unit's last level, or the next one is behind the valve
        double Nsw[10];
        uint32_t nan_sw = 0;
        if (LRSWAP && M > 2) swap_base<T>(cams, C, oobs, oobs_sw, o_valid, Nsw, nan_sw);   // all valid cameras mirrored, once per ticket
        const bool last_level = level >= (int)e->Lmax || sBinom[C * 33 + level + 1] > a.max_subsets;
        unsigned long long st_cams = 0;                         // camera-error evaluations of plain candidates (lane level)

        double be = kInf, bq0 = d_nan(), bq1 = d_nan(), bq2 = d_nan();
        uint32_t brank = kNone, bS = 0;
        double se = kInf, sq0 = d_nan(), sq1 = d_nan(), sq2 = d_nan();
        uint32_t srank = kNone, sS = 0;
        // The swap candidates of a level count only if its plain minimum stays above the threshold (triangulation.py:509),
        // which no single chunk knows -- but once ANY chunk has seen a plain candidate under the threshold they cannot
        // count any more: the wave that sees one raises a flag in the entry, and every wave stops evaluating swap
        // candidates of that entry from its next round on.  Exact whatever the timing: the reduction ignores them then.
        uint32_t *plain_ok = const_cast<uint32_t *>(&e->pad0);
        bool skip_swap = false;
        // The waves that work on the other chunks of this level share their best plain error: a wave whose own chunk holds
        // nothing good -- in the ranked enumeration every chunk but the first -- would otherwise prune against its own
        // best only.  One word per entry, the bits of a non-negative float rounded UP from the double (atomicMin on the bits
        // orders like the values): always an upper bound of the level's minimum, so what it prunes cannot be the argmin,
        // whatever the timing (the candidate that holds the minimum, and any tie of it, stay under the bound).
        uint32_t *shared_best = const_cast<uint32_t *>(&e->pad1);
        double published = kInf;
        // Every lane walks a run of consecutive ranks of the chunk (lane l: ranks r_begin + l per_lane ...): it unranks the
        // first one and steps to the next combination with bit arithmetic -- the combination of positions p is kept as
        // the bits C-1-p, where itertools' lexicographic order is the DECREASING order of the integers of one bit
        // count, and the next lower one comes by Gosper's step on the complement.  (Unranking every candidate, 64
        // consecutive ranks per round, was a walk of ~C dependent LDS reads per candidate.)  Rank 0 of the level, the
        // most suspicious subset, is still in the chunk's first round.
        const uint32_t per_lane = (r_end - r_begin + 63u) >> 6;
        const uint32_t my_begin = r_begin + (uint32_t)lane * per_lane, my_end = min(r_end, my_begin + per_lane);
        uint32_t xr = 0;
        if (my_begin < my_end) xr = __brev(unrank_subset(my_begin, C, level, sBinom)) >> (32 - C);
        for (uint32_t j = 0; j < per_lane; ++j) {
            bool go = my_begin + j < my_end;
            if (LRSWAP && !skip_swap) skip_swap = __hip_atomic_load(plain_ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
            uint32_t S = 0;
            if (go) {
                for (uint32_t b = xr; b != 0u; b &= b - 1) S |= 1u << sPerm[C - 1 - __builtin_ctz(b)];   // ranked position -> camera
                // duplicates of one effective configuration (quirk Q1): only the lexicographically first one -- padding =
                // the lowest cameras of the excluded set -- can win
                const uint32_t pad = S & o_d;
                const uint32_t below = pad ? ((2u << (31 - __builtin_clz(pad))) - 1u) : 0u;
                go = (o_d & below) == pad;
            }
            {   // the lane's next combination (what a lane past its run computes is never looked at)
                const uint32_t yc = ~xr & allmask;
                const uint32_t lowest = yc & (0u - yc), carried = yc + lowest;
                xr = ~(carried | (((yc ^ carried) >> 2) >> __builtin_ctz(yc | 0x80000000u))) & allmask;
            }
            if (!__any(go)) continue;
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
            bool alive = go;
            if (prune) {
                const double own = wave_min_d(be);              // any finished candidate of the wave bounds the level's minimum
                if (own < published) {                          // (wave-uniform) tell the other chunks' waves
                    published = own;
                    float up = (float)own;
                    if ((double)up < own) up = nextafterf(up, __builtin_huge_valf());
                    if (lane == 0) atomicMin(shared_best, __float_as_uint(up));
                }
                const double bw = fmin(own, (double)__uint_as_float(__hip_atomic_load(shared_best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)));
                // with L/R-swap candidates in play the plain minimum of a FAILED level still reports its camera count
                // when a swap candidate rescues the level (triangulation.py:576-579): only the best-so-far bounds then
                const double bmean = (last_level || (LRSWAP && M > 2)) ? bw : fmin(bw, thr);
                const double bnd = bmean * (double)nkept * (1.0 + 1e-9);      // inf while nothing bounds (0 x inf = NaN: never out)
                double psum = 0.0;
                int idx = 0;
                for (; idx < C; ++idx) {
                    const int c = __builtin_amdgcn_readfirstlane((int)sPerm[idx]);
                    double x, y, w;
                    oobs.raw(c, x, y, w);
                    bool reg;
                    const double dd = camera_distance<UNDISTORT>(cams + c, q, x, y, reg);
                    psum += (((kept >> c) & 1u) && reg && dd == dd) ? dd : 0.0;   // what cannot be told here adds nothing
                    if ((idx & 3) == 3 && __all(!go || psum > bnd)) { ++idx; break; }
                }
                st_cams += (unsigned long long)idx * (unsigned long long)__popcll(__ballot(go));
                alive = go && !(psum > bnd);
            }
            double err = kInf;
            if (__any(alive)) {
                err = mean_error<T, UNDISTORT, 0>(cams, C, oobs, kept, q);
                st_cams += (unsigned long long)C * (unsigned long long)__popcll(__ballot(go));
            }
            // rank in itertools order (ties, reduction): only where a candidate can replace the lane's best
            uint32_t rt = kNone;
            bool have_rank = false;
            const bool may_win = alive && (brank == kNone || !(err > be));
            if (__any(may_win)) { rt = rank_subset(S, C, level, sBinom); have_rank = true; }
            if (may_win && better_candidate(err, rt, be, brank)) { be = err; bq0 = q[0]; bq1 = q[1]; bq2 = q[2]; brank = rt; bS = S; }
            if (LRSWAP && !skip_swap && __any(alive && err <= thr)) {
                skip_swap = true;
                if (lane == 0) atomicOr(plain_ok, 1u);
            }
            if (LRSWAP && M > 2 && !skip_swap) {
                // the swap candidate counts only if the level's plain minimum stays above the threshold, which no
                // single wave knows: every chunk evaluates it, the reduction decides (triangulation.py:509)
                double qs[3];
                swap_solve_from_base<T>(Nsw, nan_sw, sP, oobs, oobs_sw, o_valid, kept, M, go, qs);
                double es;
                if (prune) {
                    // a swap candidate counts only under the threshold unless this is the last level (:576-579); the
                    // cameras that disagree with the majority are the suspicious ones for the mirrored point as well
                    const double sw = wave_min_d(se);
                    es = swap_error_pruned<T, UNDISTORT>(cams, C, oobs_sw, kept, M, qs, sPerm, go, last_level ? sw : fmin(sw, thr));
                } else {
                    es = swap_error<T, UNDISTORT, 0>(cams, C, oobs_sw, kept, M, qs);
                }
                const bool swap_may_win = go && (srank == kNone || !(es > se));
                if (!have_rank && __any(swap_may_win)) rt = rank_subset(S, C, level, sBinom);
                if (swap_may_win && (srank == kNone || es < se || (es == se && rt < srank))) { se = es; sq0 = qs[0]; sq1 = qs[1]; sq2 = qs[2]; srank = rt; sS = S; }
            }
        }
        // wave argmin, lowest rank on ties (np.nanargmin / np.argmin)
        for (int off = 32; off > 0; off >>= 1) {
            const double oe = shfl_d(be, lane ^ off);
            const uint32_t orank = __shfl(brank, lane ^ off, 64);
            const bool take = better_candidate(oe, orank, be, brank);
            const double t0 = shfl_d(bq0, lane ^ off), t1 = shfl_d(bq1, lane ^ off), t2 = shfl_d(bq2, lane ^ off);
            const uint32_t tS = __shfl(bS, lane ^ off, 64);
            if (take) { be = oe; brank = orank; bq0 = t0; bq1 = t1; bq2 = t2; bS = tS; }
            if (LRSWAP) {
                const double xe = shfl_d(se, lane ^ off);
                const uint32_t xrank = __shfl(srank, lane ^ off, 64);
                const bool tk = (xrank != kNone) && (srank == kNone || xe < se || (xe == se && xrank < srank));
                const double s0 = shfl_d(sq0, lane ^ off), s1 = shfl_d(sq1, lane ^ off), s2 = shfl_d(sq2, lane ^ off);
                const uint32_t xS = __shfl(sS, lane ^ off, 64);
                if (tk) { se = xe; srank = xrank; sq0 = s0; sq1 = s1; sq2 = s2; sS = xS; }
            }
        }
        if (lane == 0) {
            if (a.stats && prune) atomicAdd(a.stats + 4, st_cams);
            P2sDeepPartial &p = d.partials[t];
            p.e = be; p.q[0] = bq0; p.q[1] = bq1; p.q[2] = bq2; p.rank = brank; p.S = bS;
            p.se = se; p.sq[0] = sq0; p.sq[1] = sq1; p.sq[2] = sq2; p.srank = srank; p.sS = sS;
        }
        wave_sync();                                        // the next ticket overwrites this wave's LDS region
    }
}

// ---- reduce: the level's result of every scheduled entry -------------------------------------------------------------------
template <bool LRSWAP>
__global__ void __launch_bounds__(64) p2s_deep_reduce_kernel(const P2sTriArgs a, const P2sDeepArgs d) {
    const int C = a.C;
    const uint32_t n = min(d.ctl[P2S_DEEP_N_ENTRIES], d.capacity);
    const int lane = threadIdx.x;
    const double thr = a.thr;
    const uint32_t allmask = (C == 32) ? 0xffffffffu : ((1u << C) - 1u);
    // counters of this wave's entries, added to the shared ones once at the end (one atomic per entry on one address was
    // most of this kernel's time)
    unsigned long long st_subsets = 0, st_passes = 0, st_capped = 0;
    uint32_t finished = 0;
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        P2sDeepEntry *e = entry_at(d, i);
        if (e->state != P2S_DEEP_SCHEDULED) continue;
        double be = kInf, se = kInf;
        uint32_t brank = kNone, srank = kNone, bt = 0, st = 0;         // bt / st: ticket holding the candidate
        for (uint32_t c = lane; c < e->n_chunks; c += 64) {
            const P2sDeepPartial &p = d.partials[e->first_ticket + c];
            if (better_candidate(p.e, p.rank, be, brank)) { be = p.e; brank = p.rank; bt = e->first_ticket + c; }
            if (LRSWAP && p.srank != kNone && (srank == kNone || p.se < se || (p.se == se && p.srank < srank))) { se = p.se; srank = p.srank; st = e->first_ticket + c; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double oe = shfl_d(be, lane ^ off);
            const uint32_t orank = __shfl(brank, lane ^ off, 64), ot = __shfl(bt, lane ^ off, 64);
            if (better_candidate(oe, orank, be, brank)) { be = oe; brank = orank; bt = ot; }
            if (LRSWAP) {
                const double xe = shfl_d(se, lane ^ off);
                const uint32_t xrank = __shfl(srank, lane ^ off, 64), xt = __shfl(st, lane ^ off, 64);
                if ((xrank != kNone) && (srank == kNone || xe < se || (xe == se && xrank < srank))) { se = xe; srank = xrank; st = xt; }
            }
        }
        if (lane != 0) continue;
        // the level's result, as in the search kernel (triangulation.py:500-505, 509-579)
        const uint32_t o_nan = e->nanmask, o_d = e->nanmask | e->zeromask, o_valid = allmask & ~o_d;
        const int oV = __popc(o_d);
        const int level = (int)e->level;
        const int M = C - oV - level;
        const P2sDeepPartial &pb = d.partials[bt];
        double l_err = be, l_q0 = pb.q[0], l_q1 = pb.q[1], l_q2 = pb.q[2];
        uint32_t l_mask = o_nan | pb.S;
        const uint32_t l_nexcl = (uint32_t)(oV + __popc(pb.S & o_valid));            // :436 counts NaN or zero
        if (LRSWAP && l_err > thr && M > 2 && srank != kNone && se < l_err) {        // :576-579, nb_cams_excluded NOT updated
            const P2sDeepPartial &ps = d.partials[st];
            l_err = se; l_q0 = ps.sq[0]; l_q1 = ps.sq[1]; l_q2 = ps.sq[2]; l_mask = o_nan | ps.sS;
        }
        e->err_min = l_err; e->Q[0] = l_q0; e->Q[1] = l_q1; e->Q[2] = l_q2; e->mask = l_mask; e->n_excl = l_nexcl;
        const bool more = (l_err > thr) && (level + 1 <= e->Lmax);
        const bool cont = more && (a.binom[C * 33 + level + 1] <= a.max_subsets);
        if (more && !cont) ++st_capped;                                              // stopped by the safety valve
        st_subsets += (unsigned long long)a.binom[C * 33 + level];
        st_passes += (unsigned long long)e->n_chunks * (P2S_DEEP_CHUNK / 64);
        if (cont) {
            e->level = (uint32_t)(level + 1);
            e->state = P2S_DEEP_WAITING;
        } else {                                                                     // triangulation.py:588-604
            const int64_t gu = a.block0 * a.K + e->unit;
            const bool fail = !(l_err <= thr);
            double *Qo = a.Q + gu * 3;
            Qo[0] = fail ? d_nan() : l_q0; Qo[1] = fail ? d_nan() : l_q1; Qo[2] = fail ? d_nan() : l_q2;
            a.err[gu] = fail ? __builtin_nanf("") : (float)l_err;
            a.n_excl[gu] = (uint8_t)l_nexcl;
            a.mask[gu] = l_mask;
            e->state = P2S_DEEP_DONE;
            ++finished;
        }
    }
    if (lane == 0) {
        if (finished) atomicSub(d.ctl + P2S_DEEP_PENDING, finished);
        if (a.stats) {
            if (st_capped) atomicAdd(a.stats + 3, st_capped);
            if (st_subsets) { atomicAdd(a.stats + 1, st_subsets); if (d.prune) atomicAdd(a.stats + 5, st_subsets); }
            if (st_passes) atomicAdd(a.stats + 2, st_passes);
        }
    }
}

template <typename T, bool U, bool L>
hipError_t launch_round(const P2sTriArgs &a, const P2sDeepArgs &d, int grid_eval, int lds, hipStream_t s) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&p2s_deep_eval_kernel<T, U, L>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(p2s_deep_plan_kernel, dim3(1), dim3(1024), 0, s, d, a.binom, a.C);
    hipLaunchKernelGGL((p2s_deep_eval_kernel<T, U, L>), dim3(grid_eval), dim3(256), lds, s, a, d);
    hipLaunchKernelGGL((p2s_deep_reduce_kernel<L>), dim3(3072), dim3(64), 0, s, a, d);   // a wave per entry, grid-stride: 12 waves per CU
    return hipGetLastError();
}

}  // namespace

hipError_t p2s_launch_deep_round(const P2sTriArgs &a, const P2sDeepArgs &d, int dtype, int grid_eval, int lds, hipStream_t s) {
    const bool U = a.undistort != 0, L = a.lr_swap != 0;
    if (dtype == 0) {
        if (U) return L ? launch_round<float, true, true>(a, d, grid_eval, lds, s) : launch_round<float, true, false>(a, d, grid_eval, lds, s);
        return L ? launch_round<float, false, true>(a, d, grid_eval, lds, s) : launch_round<float, false, false>(a, d, grid_eval, lds, s);
    }
    if (U) return L ? launch_round<double, true, true>(a, d, grid_eval, lds, s) : launch_round<double, true, false>(a, d, grid_eval, lds, s);
    return L ? launch_round<double, false, true>(a, d, grid_eval, lds, s) : launch_round<double, false, false>(a, d, grid_eval, lds, s);
}
