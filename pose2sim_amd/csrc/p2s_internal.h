// Internal structures shared by the C-ABI layer (p2s_api.hip) and the kernels.
#ifndef P2S_INTERNAL_H
#define P2S_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "p2s.h"

// One camera, device resident (computeP + retrieve_calib_params, common.py:254-324).
struct P2sCam {
    double P[12];          // projection matrix (optim_K based when undistorting)
    double fx, fy, cx, cy; // original intrinsics
    double ifx, ify;       // 1/fx, 1/fy (OpenCV multiplies by reciprocals)
    double k[5];           // k1 k2 p1 p2 k3
    double R[9];           // rotation matrix, world -> camera
    double T[3];
    double nk[9];          // optim_K
    double iK[9];          // inverse of the original K        (association rays)
    double center[3];      // -R^T T                            (association rays)
    float Pf[12];          // P rounded to float: the fp32 screen of the camera-subset search (p2s_tri_pool.hip)
    float pad_[4];
};

// Kernel diagnostics (phase stamps, memory-skeleton modes) exist only in a -DP2S_DIAG build (exp/README.md): in the
// shipped library the mode is the constant 0 and every branch on it is compiled out.
#ifdef P2S_DIAG
#define P2S_DEBUG_MODE(a) ((a).debug_mode)
#else
#define P2S_DEBUG_MODE(a) 0
#endif

// p2s_get_tri_stats: [0] units that entered the camera-subset search, [1] camera subsets evaluated (lane-evaluations of
// DLT + reprojection error beyond level 0), [2] evaluation passes (64-lane), [3] units whose search was cut short by
// P2S_MAX_SUBSETS_PER_LEVEL (they come back as not triangulated; the reference would have gone on), [4] per-camera errors
// computed for the candidates of the pruned passes, [5] those candidates (a part of [1]), [6] camera subsets looked at by
// the fp32 screen of p2s_tri_pool.hip (only its survivors are evaluated and counted in [1]), [7] screen passes (64-lane)
#define P2S_N_STATS 8
// the counters are kept in shards of their own 64-byte lines (a wave adds to shard blockIdx % P2S_STAT_SHARDS): one
// word takes ~88 atomics per microsecond, and 40 000 waves adding to ONE word cost 1.2 ms per launch
#define P2S_STAT_SHARDS 256
#define P2S_STAT_STRIDE 8        // unsigned long long per shard (64 bytes)

#define P2S_MAX_SUBSETS_PER_LEVEL (1u << 26)   // search kernel: deeper levels are not entered (see p2s_tri.hip)
// The work list is cut into shards (workgroup b appends to shard b % P2S_WL_SHARDS) so that the
// append counters do not serialise: one returning atomic per wave on ONE word caps near 90 per us.
#define P2S_WL_SHARDS 128
#define P2S_REC_HDR 16           // record header: u32 unit in chunk + padding to 16 bytes (records are 16-byte multiples)

struct P2sTriArgs {
    const void *xyl;             // whole tensor [n_blocks_total][C][K][3]
    const int32_t *swap_idx;
    double *Q;
    float *err;
    uint8_t *n_excl;
    uint32_t *mask;
    const P2sCam *cams;
    const uint32_t *binom;       // [33][33] binomial coefficients
    unsigned long long *stats;   // [P2S_N_STATS] counters of this context, added to with one atomic per wave (may be NULL)
    const uint16_t *sub_tab;     // fused kernel (C <= 16): every camera subset as a bit mask, by level, in itertools.combinations order
    const uint32_t *sub_off;     // [C + 2] start of level k in sub_tab
    uint32_t *wl_count;          // work list of this chunk: P2S_WL_SHARDS record counts, then P2S_WL_SHARDS job tickets (zeroed before kernel 1)
    unsigned char *wl_rec;       // records: {u32 unit id in chunk, pad to 16 B, T obs[C][3] (, T obs_swapped[C][3]), pad to 16 B}
    int64_t block0;              // first (frame, person) block of this chunk
    int64_t n_blocks;            // blocks in this chunk
    uint32_t wl_capacity;        // records per shard
    int32_t rec_bytes;
    int32_t K, C, FB;
    int32_t lds_binom_off, lds_rec_off;   // kernel 2 LDS layout: [P][binom][records]
    int32_t min_cams, undistort, lr_swap;
    unsigned char *deep_entries; // search kernel: units about to enter a level of more than deep_min_subsets subsets are
    uint32_t *deep_ctl;          //   exported here (NULL: every level is walked in the wave)
    uint32_t deep_capacity, deep_entry_bytes, deep_min_subsets;
    uint32_t max_subsets;        // search kernel: a level with more subsets is not entered (P2S_MAX_SUBSETS_PER_LEVEL unless tuned)
    uint32_t pool_pairs, pool_singles;   // p2s_tri_fused.hip: per XCD, workgroups that take two tiles, then workgroups that take one
    int32_t screen;              // p2s_tri_pool.hip: 1 = fp32 screen of the candidates, 0 = every candidate goes to the fp64 evaluation
    int32_t prune;               // exact pruning of the subset evaluations where a wave works on one unit (search kernel, deep rounds)
    int32_t job;                 // work-list records a search wave takes at a time (<= 64); sizes its LDS region
    int32_t debug_mode;          // 0 = normal; diagnostics only: 1 = stage + store (no compute), 2 = every tile reads tile 0
    double thr, lik_thr;
};

// ---- deep levels of the search, spread over the GPU (p2s_tri_deep.hip) ----------------------------------------------
#ifndef P2S_DEEP_CHUNK
#define P2S_DEEP_CHUNK 16384u            // consecutive subset ranks one wave evaluates per ticket (256 rounds of 64 lanes)
#endif
#define P2S_DEEP_MIN_SUBSETS 4096u       // a level with more subsets than this leaves the search kernel's wave (C(32, 3) = 4 960
                                         // and beyond: 7 % of the units of the 32-camera shard; the deep list holds 2^19 per
                                         // chunk of 4 M units, a unit that finds it full stays in its wave).  162 -> 142 ms on
                                         // the tenth-shard against 16 384 (1 024 the same, 256: 180)
#define P2S_DEEP_N_ENTRIES 0             // ctl words
#define P2S_DEEP_N_TICKETS 1
#define P2S_DEEP_TICKET 2
#define P2S_DEEP_PENDING 3
#define P2S_DEEP_WAITING 0u              // entry states
#define P2S_DEEP_SCHEDULED 1u
#define P2S_DEEP_DONE 2u
struct P2sDeepEntry {                    // followed by the unit's observations as in its work-list record
    uint32_t unit, level;                // unit id within the chunk; level to evaluate next
    int32_t Lmax;
    uint32_t nanmask, zeromask, mask, n_excl, state;
    uint32_t first_ticket, n_chunks, pad0, pad1;
    double err_min, Q[3];                // best of the last level (reporting only: a deeper level replaces it)
    double N[10];                        // level-0 normal matrix
};
struct P2sDeepPartial {                  // best plain and best swap candidate of one chunk
    double e, q[3], se, sq[3];
    uint32_t rank, S, srank, sS;
};
struct P2sDeepArgs {
    unsigned char *entries;
    uint32_t *ctl;
    uint32_t *sched_entry, *sched_chunk;
    P2sDeepPartial *partials;
    uint32_t capacity, max_tickets, entry_bytes, obs_bytes;
    uint32_t prune;                      // exact pruning of the evaluation (p2s_tri_deep.hip); 0: every camera of every candidate
};
hipError_t p2s_launch_deep_round(const struct P2sTriArgs &a, const P2sDeepArgs &d, int dtype, int grid_eval, int lds, hipStream_t s);

struct P2sTriLaunch {
    int grid0, threads0, lds0;   // level-0 (streaming) kernel
    int grid1, threads1, lds1;   // search kernel
    int force_tiled;             // diagnostics: use the LDS-tiled streaming kernel even when C <= 8
};

struct P2sAssocArgs {
    const int32_t *n_persons;   // [F][C]
    const int64_t *offsets;     // [F+1]
    const void *kpts;           // [rows][Kj][3]
    double *affinity;           // [F][Nmax][Nmax]
    const P2sCam *cams;
    int64_t n_frames;
    int32_t C, Kj, Nmax, max_iter;
    int32_t debug_mode;         // diagnostics only: 7 = per-frame phase timeline instead of the result (exp/assoc_trace.py)
    int32_t form;               // P2S_ASSOC_FORM_* (p2s_set_tuning: tests run both kernels on the same frames)
    unsigned long long *stats;  // sharded counters (frames, ADMM passes, Jacobi sweeps, fp64 operations) or NULL
    double recon_thr, min_affinity, w_rank, tol, w_sparse;
};

hipError_t p2s_launch_tri(const P2sTriArgs &a, int dtype, const P2sTriLaunch &g, hipStream_t s, hipStream_t side,
                          hipEvent_t k1_done);
// p2s_tri_fused.hip: streaming pass + in-wave subset search in one launch (pinhole, no L/R swap, C <= 16); up to 8
// cameras two tiles per wave, except for the last singles_pct % of every XCD's tiles
bool p2s_tri_fused_supports(int C, int dtype, int undistort, int lr_swap);
hipError_t p2s_launch_tri_fused(const P2sTriArgs &a, int dtype, int singles_pct, hipStream_t s);
// p2s_tri_pool.hip: persistent waves, failures pooled across tiles, fp32 screen + fp64 evaluation of the survivors
bool p2s_tri_pool_supports(int C, int dtype, int undistort, int lr_swap);
hipError_t p2s_launch_tri_pool(const P2sTriArgs &a, int dtype, int singles_pct, int tiles_per_wave, hipStream_t s);
struct P2sSingleArgs {
    const int32_t *n_persons;   // [F][C]
    const int64_t *offsets;     // [F+1]
    const void *tracked;        // [rows][3]
    int32_t *comb;              // [F][C]
    double *err;                // [F]
    double *Q;                  // [F][3]
    const P2sCam *cams;
    const uint32_t *binom;
    int64_t n_frames;
    int32_t C, min_cams;
    double thr, lik_thr;
};
hipError_t p2s_launch_single(const P2sSingleArgs &a, int dtype, hipStream_t s);

hipError_t p2s_launch_assoc(const P2sAssocArgs &a, int dtype, hipStream_t s);

// p2s_filter.hip
#define P2S_MAX_FILTER_ORDER 8
struct P2sFilterArgs {
    const double *in;            // [n_frames][n_cols]
    double *out;                 // [n_frames][n_cols]
    double *work;                // [n_frames + 2 padlen][n_cols] forward-pass output
    int64_t n_frames;
    int32_t n_cols, n_order, padlen;   // n_order = len(b) - 1
    double b[P2S_MAX_FILTER_ORDER + 1], a[P2S_MAX_FILTER_ORDER + 1], zi[P2S_MAX_FILTER_ORDER];
};
hipError_t p2s_launch_butter(const P2sFilterArgs &a, hipStream_t s);

// the window filters (Hampel, Gaussian, median) and the one-euro recurrence of filtering.py
struct P2sColFilterArgs {
    const double *in;            // [n_frames][n_cols]
    double *out;                 // [n_frames][n_cols]
    double *work;                // one-euro: forward pass [n_frames][n_cols]
    const double *w;             // Gaussian: 2 radius + 1 weights (device)
    int64_t n_frames;
    int32_t n_cols, kind, radius;
    double p[4];                 // Hampel: n_sigma; one-euro: dt, min_cutoff, beta, d_cutoff
};
hipError_t p2s_launch_col_filter(const P2sColFilterArgs &a, hipStream_t s);

struct P2sMetricsArgs {
    const double *xyz;           // [n_frames][n_markers][3]
    const int32_t *bones;        // [n_bones][2] (parent, child) marker indices
    double *bone_len;            // [n_bones][n_frames]
    double *bone_stats;          // [n_bones][3] mean, population sd, n_valid
    double *accel;               // [n_markers][n_frames - 2]
    int64_t *missing;            // [n_markers]
    int64_t n_frames;
    int32_t n_markers, n_bones;
};
hipError_t p2s_launch_trc_metrics(const P2sMetricsArgs &a, hipStream_t s);

#endif
