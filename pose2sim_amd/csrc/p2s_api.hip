// p2s_api.hip -- the extern "C" boundary declared in include/p2s.h.
//
// Owns the per-GPU context (stream, calibration, scratch) and validates operand shapes on the
// host before any kernel is launched.  No torch types, no exceptions across the ABI.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>

#include "p2s.h"
#include "p2s_internal.h"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? P2S_ERR_OOM : P2S_ERR_HIP, "%s failed: %s", #expr, \
                        hipGetErrorString(e_));                                                    \
    } while (0)

struct Scratch {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t n) {
        if (n <= bytes) return P2S_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) return fail(P2S_ERR_OOM, "hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
        bytes = n;
        return P2S_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
};

}  // namespace

struct p2s_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    P2sCam *d_cams = nullptr;
    uint32_t *d_binom = nullptr;
    int n_cams = 0;
    bool full_calib = false;     // K, dist, R, T, newK were provided
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t side_stream = nullptr;               // search kernels run here, beside the next chunk's streaming pass
    hipEvent_t ev_k1[2] = {nullptr, nullptr}, ev_k2[2] = {nullptr, nullptr};
    Scratch in, swap, q, err, nexcl, mask, aux0, aux1;
    Scratch wl_rec, wl_count;
    Scratch deep_entries, deep_ctl, deep_sched, deep_partials;   // deep levels of the search (p2s_tri_deep.hip)
    uint32_t deep_min_subsets = P2S_DEEP_MIN_SUBSETS;            // 0 = every level stays in the search kernel's wave
    unsigned long long *d_stats = nullptr;           // P2S_N_STATS counters (p2s_get_tri_stats)
    unsigned long long *d_assoc_stats = nullptr;     // 4 counters (p2s_get_assoc_stats)
    uint16_t *d_sub_tab = nullptr;                   // camera subsets by level (fused kernel), built with the calibration
    uint32_t *d_sub_off = nullptr;
    // p2s_set_tuning: experiments and tests only, never read from the environment
    int tri_path = P2S_TRI_PATH_AUTO;
    int force_tiled = 0, no_overlap = 0, job = 0;
    uint32_t max_subsets = P2S_MAX_SUBSETS_PER_LEVEL;
    int debug_mode = 0;                              // honoured by a -DP2S_DIAG build only
    int assoc_form = P2S_ASSOC_FORM_AUTO;
    int deep_prune = 1;                              // p2s_tri_deep.hip: exact pruning of the deep levels' evaluations
    int pool_singles_pct = 8;                        // p2s_tri_fused.hip: share of the tiles that the last workgroups take one at a time
    int screen = 1;                                  // p2s_tri_pool.hip: fp32 screen of the camera-subset candidates
    int pool_tiles = 5;                              // p2s_tri_pool.hip: tiles a wave streams before it searches their pooled failures (2..6)
};

namespace {

struct Geometry {
    int FB, threads, lds_bytes;
};

int gcd_i(int a, int b) { return b ? gcd_i(b, a % b) : a; }

// Tile geometry of the streaming kernel: FB consecutive (frame, person) blocks per workgroup.
// The tile start must stay 16-byte aligned for the dwordx4 staging loads, the lanes of a workgroup
// should be nearly all busy (FB*K close to a multiple of 64), and several workgroups should fit
// the CU's 160 KB LDS.
Geometry choose_geometry(int C, int K, int dtype) {
    const int elem = dtype == P2S_F32 ? 4 : 8;
    const long blk_bytes = (long)C * K * 3 * elem;
    const int step = 16 / gcd_i(16, (int)(blk_bytes % 16 == 0 ? 16 : blk_bytes % 16));
    Geometry best{};
    double best_score = -1.0;
    for (int FB = step; FB <= 4096; FB += step) {
        const long tile = FB * blk_bytes;
        const long lds = (tile + 15) / 16 * 16;
        if (lds > 150 * 1024) break;
        const long units = (long)FB * K;
        const int threads = (int)std::min<long>(256, (units + 63) / 64 * 64);
        const long passes = (units + threads - 1) / threads;
        const double eff = (double)units / (double)(passes * threads);
        const int wg_per_cu = (int)std::min<long>(8, (160 * 1024) / lds);
        const int waves = std::min(32, wg_per_cu * threads / 64);
        double score = eff * std::min(1.0, waves / 12.0);
        if (lds > 64 * 1024) score *= 0.8;
        if (passes > 1) score *= 0.9;
        if (score > best_score + 1e-9) {
            best_score = score;
            best.FB = FB;
            best.threads = threads;
            best.lds_bytes = (int)lds;
        }
    }
    if (best_score < 0) best.FB = 0;   // a single block does not fit: not supported
    return best;
}

constexpr int64_t kChunkUnits = 1 << 22;   // units per (level-0, search) kernel pair: bounds the work-list scratch

void fill_binom(uint32_t *b) {
    for (int n = 0; n < 33; ++n)
        for (int k = 0; k < 33; ++k) {
            unsigned long long v;
            if (k > n) v = 0;
            else if (k == 0 || k == n) v = 1;
            else v = (unsigned long long)b[(n - 1) * 33 + k - 1] + b[(n - 1) * 33 + k];
            b[n * 33 + k] = (uint32_t)std::min<unsigned long long>(v, 0xfffffffeull);
        }
}

int check_tri(p2s_ctx *ctx, int64_t n_blocks, int32_t K, int32_t dtype, const p2s_tri_params *p,
              const void *swap_idx) {
    if (!ctx) return fail(P2S_ERR_INVALID_ARG, "null context");
    if (ctx->n_cams <= 0) return fail(P2S_ERR_NO_CALIB, "p2s_set_calibration has not been called");
    if (!p) return fail(P2S_ERR_INVALID_ARG, "null params");
    if (n_blocks < 0 || K <= 0) return fail(P2S_ERR_INVALID_ARG, "bad shape: n_blocks=%lld K=%d", (long long)n_blocks, K);
    if (dtype != P2S_F32 && dtype != P2S_F64) return fail(P2S_ERR_INVALID_ARG, "dtype must be P2S_F32 or P2S_F64");
    if (p->min_cameras < 1) return fail(P2S_ERR_INVALID_ARG, "min_cameras must be >= 1 (got %d)", p->min_cameras);
    if (!(p->reproj_error_threshold == p->reproj_error_threshold))
        return fail(P2S_ERR_INVALID_ARG, "reproj_error_threshold is NaN");
    if (p->undistort_points && !ctx->full_calib)
        return fail(P2S_ERR_NO_CALIB, "undistort_points needs K, dist, R, T and optim_K in p2s_set_calibration");
    if (p->handle_lr_swap && !swap_idx) return fail(P2S_ERR_INVALID_ARG, "handle_lr_swap needs swap_idx");
    if (n_blocks * (int64_t)K >= (int64_t)1 << 40) return fail(P2S_ERR_INVALID_ARG, "too many units");
    return P2S_OK;
}

}  // namespace

// Error slot shared with the host-side translation units (p2s_ingest.cpp).
int p2s_set_error(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

extern "C" {

int p2s_version(void) { return 100; }

const char *p2s_last_error(void) { return g_last_error.c_str(); }

int p2s_device_count(int *count) {
    if (!count) return fail(P2S_ERR_INVALID_ARG, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *count = n;
    return P2S_OK;
}

int p2s_create(int device_id, p2s_ctx **out) {
    if (!out) return fail(P2S_ERR_INVALID_ARG, "null out");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(P2S_ERR_NO_DEVICE, "no HIP device visible: the triangulation engine has no CPU fallback");
    }
    if (device_id < 0 || device_id >= n) return fail(P2S_ERR_INVALID_ARG, "device %d out of range (%d devices)", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    p2s_ctx *c = new p2s_ctx();
    c->device = device_id;
    HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    HIP_TRY(hipMalloc((void **)&c->d_cams, sizeof(P2sCam) * P2S_MAX_CAMS));
    HIP_TRY(hipMalloc((void **)&c->d_binom, sizeof(uint32_t) * 33 * 33));
    std::vector<uint32_t> b(33 * 33);
    fill_binom(b.data());
    HIP_TRY(hipMemcpy(c->d_binom, b.data(), b.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc((void **)&c->d_stats, sizeof(unsigned long long) * P2S_STAT_SHARDS * P2S_STAT_STRIDE));
    HIP_TRY(hipMemset(c->d_stats, 0, sizeof(unsigned long long) * P2S_STAT_SHARDS * P2S_STAT_STRIDE));
    HIP_TRY(hipMalloc((void **)&c->d_assoc_stats, sizeof(unsigned long long) * P2S_STAT_SHARDS * P2S_STAT_STRIDE));
    HIP_TRY(hipMemset(c->d_assoc_stats, 0, sizeof(unsigned long long) * P2S_STAT_SHARDS * P2S_STAT_STRIDE));
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    HIP_TRY(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipEventCreateWithFlags(&c->ev_k1[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&c->ev_k2[i], hipEventDisableTiming));
    }
    *out = c;
    return P2S_OK;
}

int p2s_destroy(p2s_ctx *ctx) {
    if (!ctx) return P2S_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->in.release(); ctx->swap.release(); ctx->q.release(); ctx->err.release();
    ctx->nexcl.release(); ctx->mask.release(); ctx->aux0.release(); ctx->aux1.release();
    ctx->wl_rec.release(); ctx->wl_count.release();
    ctx->deep_entries.release(); ctx->deep_ctl.release(); ctx->deep_sched.release(); ctx->deep_partials.release();
    if (ctx->d_cams) (void)hipFree(ctx->d_cams);
    if (ctx->d_binom) (void)hipFree(ctx->d_binom);
    if (ctx->d_stats) (void)hipFree(ctx->d_stats);
    if (ctx->d_assoc_stats) (void)hipFree(ctx->d_assoc_stats);
    if (ctx->d_sub_tab) (void)hipFree(ctx->d_sub_tab);
    if (ctx->d_sub_off) (void)hipFree(ctx->d_sub_off);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->side_stream) { (void)hipStreamSynchronize(ctx->side_stream); (void)hipStreamDestroy(ctx->side_stream); }
    for (int i = 0; i < 2; ++i) {
        if (ctx->ev_k1[i]) (void)hipEventDestroy(ctx->ev_k1[i]);
        if (ctx->ev_k2[i]) (void)hipEventDestroy(ctx->ev_k2[i]);
    }
    delete ctx;
    return P2S_OK;
}

int p2s_set_stream(p2s_ctx *ctx, void *hip_stream) {
    if (!ctx) return fail(P2S_ERR_INVALID_ARG, "null context");
    ctx->stream = (hipStream_t)hip_stream;   // NULL = HIP's default stream
    return P2S_OK;
}

int p2s_synchronize(p2s_ctx *ctx) {
    if (!ctx) return fail(P2S_ERR_INVALID_ARG, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return P2S_OK;
}

int p2s_set_calibration(p2s_ctx *ctx, int32_t n_cams, const double *P, const double *Kmat, const double *dist,
                        const double *Rmat, const double *T, const double *newK) {
    if (!ctx) return fail(P2S_ERR_INVALID_ARG, "null context");
    if (n_cams < 1 || n_cams > P2S_MAX_CAMS) return fail(P2S_ERR_INVALID_ARG, "n_cams=%d outside [1, %d]", n_cams, P2S_MAX_CAMS);
    if (!P) return fail(P2S_ERR_INVALID_ARG, "null P");
    const bool full = Kmat && dist && Rmat && T && newK;
    if (!full && (Kmat || dist || Rmat || T || newK))
        return fail(P2S_ERR_INVALID_ARG, "K, dist, R, T and optim_K must be given together or all be NULL");
    std::vector<P2sCam> cams(P2S_MAX_CAMS);
    std::memset(cams.data(), 0, sizeof(P2sCam) * P2S_MAX_CAMS);
    for (int c = 0; c < n_cams; ++c) {
        P2sCam &cam = cams[c];
        std::memcpy(cam.P, P + 12 * c, sizeof cam.P);
        for (int i = 0; i < 12; ++i) cam.Pf[i] = (float)cam.P[i];
        if (!full) continue;
        const double *K = Kmat + 9 * c;
        cam.fx = K[0]; cam.fy = K[4]; cam.cx = K[2]; cam.cy = K[5];
        cam.ifx = 1.0 / cam.fx; cam.ify = 1.0 / cam.fy;
        std::memcpy(cam.k, dist + 5 * c, sizeof cam.k);
        std::memcpy(cam.R, Rmat + 9 * c, sizeof cam.R);
        std::memcpy(cam.T, T + 3 * c, sizeof cam.T);
        std::memcpy(cam.nk, newK + 9 * c, sizeof cam.nk);
        // inverse of K (general 3x3, as numpy.linalg.inv at common.py:282) by cofactors
        const double a = K[0], b = K[1], cc = K[2], d = K[3], e = K[4], f = K[5], g = K[6], h = K[7], i = K[8];
        const double det = a * (e * i - f * h) - b * (d * i - f * g) + cc * (d * h - e * g);
        const double id = 1.0 / det;
        cam.iK[0] = (e * i - f * h) * id; cam.iK[1] = (cc * h - b * i) * id; cam.iK[2] = (b * f - cc * e) * id;
        cam.iK[3] = (f * g - d * i) * id; cam.iK[4] = (a * i - cc * g) * id; cam.iK[5] = (cc * d - a * f) * id;
        cam.iK[6] = (d * h - e * g) * id; cam.iK[7] = (b * g - a * h) * id; cam.iK[8] = (a * e - b * d) * id;
        for (int r = 0; r < 3; ++r)
            cam.center[r] = -(cam.R[0 * 3 + r] * cam.T[0] + cam.R[1 * 3 + r] * cam.T[1] + cam.R[2 * 3 + r] * cam.T[2]);
    }
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpy(ctx->d_cams, cams.data(), sizeof(P2sCam) * P2S_MAX_CAMS, hipMemcpyHostToDevice));
    // every subset of the n_cams cameras as a bit mask, level by level in itertools.combinations order
    // (triangulation.py:411): what the fused kernel's lanes index by (level, rank); 2^C entries, C <= 16
    if (n_cams <= 16) {
        std::vector<uint16_t> tab;
        std::vector<uint32_t> off(n_cams + 2, 0);
        tab.reserve((size_t)1 << n_cams);
        for (int k = 0; k <= n_cams; ++k) {
            off[k] = (uint32_t)tab.size();
            std::vector<int> idx(k);
            for (int i = 0; i < k; ++i) idx[i] = i;
            for (;;) {
                uint32_t m = 0;
                for (int i = 0; i < k; ++i) m |= 1u << idx[i];
                tab.push_back((uint16_t)m);
                int i = k - 1;
                while (i >= 0 && idx[i] == n_cams - k + i) --i;
                if (i < 0) break;
                ++idx[i];
                for (int j = i + 1; j < k; ++j) idx[j] = idx[j - 1] + 1;
            }
        }
        off[n_cams + 1] = (uint32_t)tab.size();
        if (!ctx->d_sub_tab) HIP_TRY(hipMalloc((void **)&ctx->d_sub_tab, sizeof(uint16_t) << 16));
        if (!ctx->d_sub_off) HIP_TRY(hipMalloc((void **)&ctx->d_sub_off, sizeof(uint32_t) * 18));
        HIP_TRY(hipMemcpy(ctx->d_sub_tab, tab.data(), tab.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ctx->d_sub_off, off.data(), off.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    }
    ctx->n_cams = n_cams;
    ctx->full_calib = full;
    return P2S_OK;
}

int p2s_get_tri_stats(p2s_ctx *ctx, uint64_t *out, int32_t reset) {
    if (!ctx || !out) return fail(P2S_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->side_stream) HIP_TRY(hipStreamSynchronize(ctx->side_stream));
    std::vector<unsigned long long> h((size_t)P2S_STAT_SHARDS * P2S_STAT_STRIDE);
    HIP_TRY(hipMemcpy(h.data(), ctx->d_stats, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < P2S_N_STATS; ++i) {
        out[i] = 0;
        for (int sh = 0; sh < P2S_STAT_SHARDS; ++sh) out[i] += h[(size_t)sh * P2S_STAT_STRIDE + i];
    }
    if (reset) HIP_TRY(hipMemset(ctx->d_stats, 0, h.size() * sizeof(unsigned long long)));
    return P2S_OK;
}

int p2s_get_assoc_stats(p2s_ctx *ctx, uint64_t *out, int32_t reset) {
    if (!ctx || !out) return fail(P2S_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<unsigned long long> h((size_t)P2S_STAT_SHARDS * P2S_STAT_STRIDE);
    HIP_TRY(hipMemcpy(h.data(), ctx->d_assoc_stats, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) {
        out[i] = 0;
        for (int sh = 0; sh < P2S_STAT_SHARDS; ++sh) out[i] += h[(size_t)sh * P2S_STAT_STRIDE + i];
    }
    if (reset) HIP_TRY(hipMemset(ctx->d_assoc_stats, 0, h.size() * sizeof(unsigned long long)));
    return P2S_OK;
}

int p2s_set_tuning(p2s_ctx *ctx, int32_t key, int32_t value) {
    if (!ctx) return fail(P2S_ERR_INVALID_ARG, "null context");
    switch (key) {
    case P2S_TUNE_TRI_PATH:
        if (value != P2S_TRI_PATH_AUTO && value != P2S_TRI_PATH_WORKLIST && value != P2S_TRI_PATH_ONE_TILE &&
            value != P2S_TRI_PATH_POOLED && value != P2S_TRI_PATH_TWO_TILES)
            return fail(P2S_ERR_INVALID_ARG, "unknown triangulation path %d", value);
        ctx->tri_path = value;
        return P2S_OK;
    case P2S_TUNE_FORCE_TILED: ctx->force_tiled = value ? 1 : 0; return P2S_OK;
    case P2S_TUNE_NO_OVERLAP: ctx->no_overlap = value ? 1 : 0; return P2S_OK;
    case P2S_TUNE_SEARCH_JOB:
        if (value != 0 && (value < 8 || value > 64)) return fail(P2S_ERR_INVALID_ARG, "search job size %d outside [8, 64]", value);
        ctx->job = value;
        return P2S_OK;
    case P2S_TUNE_MAX_SUBSETS:
        if (value < 1) return fail(P2S_ERR_INVALID_ARG, "max subsets per level must be >= 1");
        ctx->max_subsets = (uint32_t)value;
        return P2S_OK;
    case P2S_TUNE_DEEP_MIN_SUBSETS:
        if (value < 0) return fail(P2S_ERR_INVALID_ARG, "deep-level threshold must be >= 0");
        ctx->deep_min_subsets = (uint32_t)value;
        return P2S_OK;
    case P2S_TUNE_DEEP_PRUNE: ctx->deep_prune = value ? 1 : 0; return P2S_OK;
    case P2S_TUNE_POOL_SINGLES_PCT:
        if (value < 0 || value > 100) return fail(P2S_ERR_INVALID_ARG, "percentage outside [0, 100]");
        ctx->pool_singles_pct = value;
        return P2S_OK;
    case P2S_TUNE_SCREEN: ctx->screen = value ? 1 : 0; return P2S_OK;
    case P2S_TUNE_POOL_TILES:
        if (value < 2 || value > 6) return fail(P2S_ERR_INVALID_ARG, "tiles per wave outside [2, 6]");
        ctx->pool_tiles = value;
        return P2S_OK;
    case P2S_TUNE_ASSOC_FORM:
        if (value != P2S_ASSOC_FORM_AUTO && value != P2S_ASSOC_FORM_GENERAL)
            return fail(P2S_ERR_INVALID_ARG, "unknown association kernel form %d", value);
        ctx->assoc_form = value;
        return P2S_OK;
    case P2S_TUNE_DIAG_MODE:
#ifdef P2S_DIAG
        ctx->debug_mode = value;
        return P2S_OK;
#else
        return fail(P2S_ERR_INVALID_ARG, "kernel diagnostics need a -DP2S_DIAG build of the library");
#endif
    default: return fail(P2S_ERR_INVALID_ARG, "unknown tuning key %d", key);
    }
}

int p2s_tri_geometry(int32_t n_cams, int32_t n_kpts, int32_t dtype, int32_t *blocks_per_tile, int32_t *threads,
                     int32_t *lds_bytes) {
    if (n_cams < 1 || n_cams > P2S_MAX_CAMS || n_kpts < 1 || (dtype != P2S_F32 && dtype != P2S_F64))
        return fail(P2S_ERR_INVALID_ARG, "bad geometry query");
    Geometry g = choose_geometry(n_cams, n_kpts, dtype);
    if (g.FB == 0) return fail(P2S_ERR_INVALID_ARG, "one block of C=%d x K=%d does not fit in LDS", n_cams, n_kpts);
    if (blocks_per_tile) *blocks_per_tile = g.FB;
    if (threads) *threads = g.threads;
    if (lds_bytes) *lds_bytes = g.lds_bytes;
    return P2S_OK;
}

int p2s_triangulate_device(p2s_ctx *ctx, int64_t n_blocks, int32_t n_kpts, int32_t dtype, const void *d_xyl,
                           const int32_t *d_swap_idx, const p2s_tri_params *params, double *d_Q, float *d_err,
                           uint8_t *d_n_excl, uint32_t *d_excl_mask) {
    int rc = check_tri(ctx, n_blocks, n_kpts, dtype, params, d_swap_idx);
    if (rc != P2S_OK) return rc;
    if (n_blocks == 0) return P2S_OK;
    if (!d_xyl || !d_Q || !d_err || !d_n_excl || !d_excl_mask) return fail(P2S_ERR_INVALID_ARG, "null device pointer");
    if (((uintptr_t)d_xyl & 15) != 0) return fail(P2S_ERR_INVALID_ARG, "xyl must be 16-byte aligned");
    const int C = ctx->n_cams;
    const int elem = dtype == P2S_F32 ? 4 : 8;
    if (ctx->tri_path != P2S_TRI_PATH_WORKLIST && !ctx->force_tiled &&
        p2s_tri_fused_supports(C, dtype, params->undistort_points, params->handle_lr_swap)) {
        // one launch per chunk: streaming pass + in-wave subset search (p2s_tri_fused.hip).  A chunk keeps the
        // kernel's 32-bit byte offsets below 2^31 and starts on a multiple of 16 blocks (16-byte result stores).
        P2sTriArgs a{};
        a.xyl = d_xyl;
        a.Q = d_Q; a.err = d_err; a.n_excl = d_n_excl; a.mask = d_excl_mask;
        a.cams = ctx->d_cams;
        a.sub_tab = ctx->d_sub_tab; a.sub_off = ctx->d_sub_off; a.binom = ctx->d_binom;
        a.stats = ctx->d_stats;
        a.K = n_kpts; a.C = C;
        a.min_cams = params->min_cameras;
        a.thr = params->reproj_error_threshold;
        a.lik_thr = params->likelihood_threshold;
        // the pooled kernel (persistent waves, fp32 screen) where it applies; P2S_TUNE_TRI_PATH picks the older forms
        const bool pooled = (ctx->tri_path == P2S_TRI_PATH_AUTO || ctx->tri_path == P2S_TRI_PATH_POOLED) &&
                            p2s_tri_pool_supports(C, dtype, params->undistort_points, params->handle_lr_swap);
        a.screen = ctx->screen;
        const int64_t blk_bytes = (int64_t)C * n_kpts * 3 * elem;
        if (blk_bytes > ((int64_t)1 << 26)) return fail(P2S_ERR_INVALID_ARG, "K=%d too large", n_kpts);
        const int64_t chunk_blocks = std::max<int64_t>(16, (((int64_t)1 << 31) / blk_bytes) / 16 * 16);
        HIP_TRY(hipSetDevice(ctx->device));
        for (int64_t b0 = 0; b0 < n_blocks; b0 += chunk_blocks) {
            a.block0 = b0;
            a.n_blocks = std::min<int64_t>(chunk_blocks, n_blocks - b0);
            if (pooled)
                HIP_TRY(p2s_launch_tri_pool(a, dtype, ctx->pool_singles_pct, ctx->pool_tiles, ctx->stream));
            else
                HIP_TRY(p2s_launch_tri_fused(a, dtype, ctx->tri_path == P2S_TRI_PATH_ONE_TILE ? 100 : ctx->pool_singles_pct, ctx->stream));
        }
        return P2S_OK;
    }
    Geometry g = choose_geometry(C, n_kpts, dtype);
    if (g.FB == 0) return fail(P2S_ERR_INVALID_ARG, "one block of C=%d x K=%d does not fit in LDS", C, n_kpts);
    const int rec_bytes = P2S_REC_HDR + (3 * C * elem * (params->handle_lr_swap ? 2 : 1) + 15) / 16 * 16;

    // chunks of whole tiles, at most kChunkUnits units each
    int64_t chunk_blocks = std::max<int64_t>(g.FB, (kChunkUnits / n_kpts) / g.FB * g.FB);
    chunk_blocks = std::min<int64_t>(chunk_blocks, (n_blocks + g.FB - 1) / g.FB * g.FB);
    const int64_t n_chunks = (n_blocks + chunk_blocks - 1) / chunk_blocks;
    const int64_t chunk_units = chunk_blocks * n_kpts;
    if (chunk_units > 0xffffffffLL / 2) return fail(P2S_ERR_INVALID_ARG, "K=%d too large", n_kpts);

    HIP_TRY(hipSetDevice(ctx->device));
    // the work list: P2S_WL_SHARDS shards, workgroup b of the streaming kernel appends to shard b % SHARDS;
    // two lists are used alternately by consecutive chunks; counters are zeroed on the stream first
    const int64_t tiles_per_chunk = chunk_blocks / g.FB;
    const int64_t shard_cap_tiled = (tiles_per_chunk + P2S_WL_SHARDS - 1) / P2S_WL_SHARDS * (int64_t)g.FB * n_kpts;
    const int64_t direct_wgs = (chunk_units + 255) / 256;     // the direct kernel appends per 256-unit workgroup
    const int64_t shard_cap_direct = (direct_wgs + P2S_WL_SHARDS - 1) / P2S_WL_SHARDS * 256;
    const int64_t shard_cap = std::max(shard_cap_tiled, shard_cap_direct);
    const size_t list_bytes = (size_t)P2S_WL_SHARDS * shard_cap * rec_bytes;
    if ((rc = ctx->wl_rec.ensure(2 * list_bytes)) != P2S_OK) return rc;
    if ((rc = ctx->wl_count.ensure((size_t)n_chunks * 2 * P2S_WL_SHARDS * sizeof(uint32_t))) != P2S_OK) return rc;
    HIP_TRY(hipMemsetAsync(ctx->wl_count.p, 0, (size_t)n_chunks * 2 * P2S_WL_SHARDS * sizeof(uint32_t), ctx->stream));

    // search kernel geometry: LDS = [P][binom][waves x 64 records]
    const int lds_binom_off = (C * 12 * 8 + 15) / 16 * 16;
    const int lds_rec_off = (lds_binom_off + 33 * 33 * 4 + 15) / 16 * 16;
    // records per search job: 40 (32..48 measure alike on cfg2) unless the records are large -- a wave's LDS region
    // (job x (record + state)) is kept near 9 KB so that 3 waves per SIMD stay resident; with 32 cameras and the
    // swapped copy a 40-record job took 35 KB and left less than one wave per SIMD (10 records there: 4.6 -> 2.0 s
    // together with the two-pass swap evaluation; deep searches also balance better with small jobs)
    int job = (int)std::max<int64_t>(8, std::min<int64_t>(40, (9 * 1024) / (int64_t)(rec_bytes + 96)));
    if (ctx->job) job = ctx->job;                                                        // p2s_set_tuning: kernel experiments only
    const int64_t fit = (40 * 1024) / (job * (int64_t)(rec_bytes + 96));
    const int wpb = fit >= 4 ? 4 : fit >= 2 ? 2 : 1;   // waves per search workgroup
    const int lds1 = lds_rec_off + wpb * job * (rec_bytes + 96);  // per wave: `job` records + `job` owner states
    if (lds1 > 160 * 1024) return fail(P2S_ERR_INVALID_ARG, "search records of C=%d do not fit in LDS", C);

    P2sTriArgs a{};
    a.xyl = d_xyl;
    a.swap_idx = d_swap_idx;
    a.Q = d_Q; a.err = d_err; a.n_excl = d_n_excl; a.mask = d_excl_mask;
    a.cams = ctx->d_cams;
    a.binom = ctx->d_binom;
    a.stats = ctx->d_stats;
    a.K = n_kpts; a.C = C; a.FB = g.FB;
    a.rec_bytes = rec_bytes;
    a.wl_capacity = (uint32_t)shard_cap;
    a.lds_binom_off = lds_binom_off; a.lds_rec_off = lds_rec_off;
    a.job = job;
    a.max_subsets = ctx->max_subsets;
    a.min_cams = params->min_cameras;
    a.undistort = params->undistort_points ? 1 : 0;
    a.lr_swap = params->handle_lr_swap ? 1 : 0;
    a.thr = params->reproj_error_threshold;
    a.lik_thr = params->likelihood_threshold;
    a.debug_mode = ctx->debug_mode;                                   // 0 unless a -DP2S_DIAG build was told otherwise
    a.prune = ctx->deep_prune;

    // Deep levels: can a level of this camera count exceed the threshold at all?  Then units about to enter one are
    // exported by the search kernel and finished by rounds of plan / eval / reduce over the whole GPU, chunk by chunk
    // (the host reads a 4-byte count per round, so these calls synchronise the stream).
    bool deep = false;
    P2sDeepArgs dargs{};
    constexpr uint32_t kDeepCapacity = 1u << 19, kDeepTickets = 1u << 20;
    if (ctx->deep_min_subsets > 0) {
        std::vector<uint32_t> b(33 * 33);
        fill_binom(b.data());
        for (int k = 1; k <= C - params->min_cameras; ++k) deep = deep || b[C * 33 + k] > ctx->deep_min_subsets;
    }
    if (deep) {
        if ((uint64_t)ctx->max_subsets > (uint64_t)kDeepTickets * P2S_DEEP_CHUNK)
            return fail(P2S_ERR_INVALID_ARG, "max subsets per level exceeds the deep-level ticket buffer");
        const uint32_t obs_bytes = (uint32_t)(rec_bytes - P2S_REC_HDR);
        const uint32_t entry_bytes = (uint32_t)((sizeof(P2sDeepEntry) + obs_bytes + 15) / 16 * 16);
        if ((rc = ctx->deep_entries.ensure((size_t)kDeepCapacity * entry_bytes)) != P2S_OK) return rc;
        if ((rc = ctx->deep_ctl.ensure(64)) != P2S_OK) return rc;
        if ((rc = ctx->deep_sched.ensure((size_t)kDeepTickets * 2 * sizeof(uint32_t))) != P2S_OK) return rc;
        if ((rc = ctx->deep_partials.ensure((size_t)kDeepTickets * sizeof(P2sDeepPartial))) != P2S_OK) return rc;
        dargs.prune = ctx->deep_prune ? 1u : 0u;
        dargs.entries = (unsigned char *)ctx->deep_entries.p;
        dargs.ctl = (uint32_t *)ctx->deep_ctl.p;
        dargs.sched_entry = (uint32_t *)ctx->deep_sched.p;
        dargs.sched_chunk = dargs.sched_entry + kDeepTickets;
        dargs.partials = (P2sDeepPartial *)ctx->deep_partials.p;
        dargs.capacity = kDeepCapacity; dargs.max_tickets = kDeepTickets;
        dargs.entry_bytes = entry_bytes; dargs.obs_bytes = obs_bytes;
        a.deep_entries = dargs.entries; a.deep_ctl = dargs.ctl;
        a.deep_capacity = kDeepCapacity; a.deep_entry_bytes = entry_bytes; a.deep_min_subsets = ctx->deep_min_subsets;
    }

    for (int64_t ch = 0; ch < n_chunks; ++ch) {
        a.block0 = ch * chunk_blocks;
        a.n_blocks = std::min<int64_t>(chunk_blocks, n_blocks - a.block0);
        a.wl_count = (uint32_t *)ctx->wl_count.p + ch * 2 * P2S_WL_SHARDS;
        a.wl_rec = (unsigned char *)ctx->wl_rec.p + (size_t)(ch & 1) * list_bytes;
        P2sTriLaunch L{};
        L.grid0 = (int)((a.n_blocks + g.FB - 1) / g.FB);
        L.threads0 = g.threads;
        L.lds0 = g.lds_bytes;
        // persistent search grid: what stays resident at 3 waves per SIMD (256 CUs x 12 waves), no
        // more than the chunk could ever need
        const int64_t need_waves = (a.n_blocks * n_kpts + job - 1) / job;
        const int64_t waves = std::max<int64_t>(wpb, std::min<int64_t>(3072, need_waves));
        L.grid1 = (int)((waves + wpb - 1) / wpb);
        L.threads1 = 64 * wpb;
        L.lds1 = lds1;
        L.force_tiled = ctx->force_tiled;                    // p2s_set_tuning: tests of the tiled kernel
        const int slot = (int)(ch & 1);
        const bool overlap = n_chunks > 1 && !ctx->no_overlap && !deep;
        if (deep) HIP_TRY(hipMemsetAsync(dargs.ctl, 0, 64, ctx->stream));
        hipStream_t side = overlap ? ctx->side_stream : ctx->stream;
        if (overlap && ch >= 2) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_k2[slot], 0));   // list `slot` is free again
        HIP_TRY(p2s_launch_tri(a, dtype, L, ctx->stream, side, ctx->ev_k1[slot]));
        if (overlap) HIP_TRY(hipEventRecord(ctx->ev_k2[slot], side));
        if (deep) {
            uint32_t pending = 0;
            HIP_TRY(hipMemcpyAsync(&pending, dargs.ctl + P2S_DEEP_N_ENTRIES, sizeof pending, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            const int deep_lds = lds_rec_off + 4 * (int)dargs.obs_bytes;
            while (pending > 0) {
                HIP_TRY(p2s_launch_deep_round(a, dargs, dtype, 256 * 3, deep_lds, ctx->stream));
                HIP_TRY(hipMemcpyAsync(&pending, dargs.ctl + P2S_DEEP_PENDING, sizeof pending, hipMemcpyDeviceToHost, ctx->stream));
                HIP_TRY(hipStreamSynchronize(ctx->stream));
            }
        }
    }
    if (n_chunks > 1 && !ctx->no_overlap && !deep) {   // join: the caller's stream sees every search finished
        HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_k2[(n_chunks - 1) & 1], 0));
        if (n_chunks > 1) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_k2[(n_chunks - 2) & 1], 0));
    }
    return P2S_OK;
}

int p2s_triangulate_host(p2s_ctx *ctx, int64_t n_blocks, int32_t n_kpts, int32_t dtype, const void *xyl,
                         const int32_t *swap_idx, const p2s_tri_params *params, double *Q, float *err,
                         uint8_t *n_excl, uint32_t *excl_mask) {
    int rc = check_tri(ctx, n_blocks, n_kpts, dtype, params, swap_idx);
    if (rc != P2S_OK) return rc;
    if (n_blocks == 0) return P2S_OK;
    if (!xyl || !Q || !err || !n_excl || !excl_mask) return fail(P2S_ERR_INVALID_ARG, "null host pointer");
    const int C = ctx->n_cams;
    const size_t elem = dtype == P2S_F32 ? 4 : 8;
    const size_t n_units = (size_t)n_blocks * n_kpts;
    const size_t in_bytes = (size_t)n_blocks * C * n_kpts * 3 * elem;
    HIP_TRY(hipSetDevice(ctx->device));
    if ((rc = ctx->in.ensure(in_bytes)) != P2S_OK) return rc;
    if ((rc = ctx->q.ensure(n_units * 24)) != P2S_OK) return rc;
    if ((rc = ctx->err.ensure(n_units * 4)) != P2S_OK) return rc;
    if ((rc = ctx->nexcl.ensure(n_units)) != P2S_OK) return rc;
    if ((rc = ctx->mask.ensure(n_units * 4)) != P2S_OK) return rc;
    const int32_t *d_swap = nullptr;
    if (swap_idx) {
        if ((rc = ctx->swap.ensure((size_t)n_kpts * 4)) != P2S_OK) return rc;
        HIP_TRY(hipMemcpyAsync(ctx->swap.p, swap_idx, (size_t)n_kpts * 4, hipMemcpyHostToDevice, ctx->stream));
        d_swap = (const int32_t *)ctx->swap.p;
    }
    HIP_TRY(hipMemcpyAsync(ctx->in.p, xyl, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = p2s_triangulate_device(ctx, n_blocks, n_kpts, dtype, ctx->in.p, d_swap, params, (double *)ctx->q.p,
                                (float *)ctx->err.p, (uint8_t *)ctx->nexcl.p, (uint32_t *)ctx->mask.p);
    if (rc != P2S_OK) return rc;
    HIP_TRY(hipMemcpyAsync(Q, ctx->q.p, n_units * 24, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(err, ctx->err.p, n_units * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(n_excl, ctx->nexcl.p, n_units, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(excl_mask, ctx->mask.p, n_units * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return P2S_OK;
}

static int check_assoc(p2s_ctx *ctx, int64_t n_frames, int32_t Kj, int32_t n_max, int32_t dtype,
                       const p2s_assoc_params *p) {
    if (!ctx) return fail(P2S_ERR_INVALID_ARG, "null context");
    if (ctx->n_cams <= 0 || !ctx->full_calib)
        return fail(P2S_ERR_NO_CALIB, "association needs K, R and T in p2s_set_calibration");
    if (!p) return fail(P2S_ERR_INVALID_ARG, "null params");
    if (n_frames < 0 || n_frames > 0x7fffffffLL || Kj <= 0) return fail(P2S_ERR_INVALID_ARG, "bad shape");
    if (n_max < 1 || n_max > P2S_MAX_PERSONS_TOTAL)
        return fail(P2S_ERR_INVALID_ARG, "n_max=%d outside [1, %d]", n_max, P2S_MAX_PERSONS_TOTAL);
    if (dtype != P2S_F32 && dtype != P2S_F64) return fail(P2S_ERR_INVALID_ARG, "dtype must be P2S_F32 or P2S_F64");
    if (!(p->reconstruction_error_threshold > 0)) return fail(P2S_ERR_INVALID_ARG, "reconstruction_error_threshold must be > 0");
    if (p->max_iter < 0) return fail(P2S_ERR_INVALID_ARG, "max_iter < 0");
    return P2S_OK;
}

int p2s_associate_device(p2s_ctx *ctx, int64_t n_frames, int32_t n_kpts_json, int32_t n_max, int32_t dtype,
                         const int32_t *d_n_persons, const int64_t *d_offsets, const void *d_kpts,
                         const p2s_assoc_params *params, double *d_affinity) {
    int rc = check_assoc(ctx, n_frames, n_kpts_json, n_max, dtype, params);
    if (rc != P2S_OK) return rc;
    if (n_frames == 0) return P2S_OK;
    if (!d_n_persons || !d_offsets || !d_kpts || !d_affinity) return fail(P2S_ERR_INVALID_ARG, "null device pointer");
    if (n_max & 1) return fail(P2S_ERR_INVALID_ARG, "n_max must be even (pad the affinity stride)");
    P2sAssocArgs a{};
    a.n_persons = d_n_persons; a.offsets = d_offsets; a.kpts = d_kpts; a.affinity = d_affinity;
    a.cams = ctx->d_cams;
    a.n_frames = n_frames; a.C = ctx->n_cams; a.Kj = n_kpts_json; a.Nmax = n_max;
    a.max_iter = params->max_iter;
    a.debug_mode = ctx->debug_mode;
    a.form = ctx->assoc_form;
    a.stats = ctx->d_assoc_stats;
    a.recon_thr = params->reconstruction_error_threshold; a.min_affinity = params->min_affinity;
    a.w_rank = params->w_rank; a.tol = params->tol; a.w_sparse = params->w_sparse;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(p2s_launch_assoc(a, dtype, ctx->stream));
    return P2S_OK;
}

int p2s_associate_host(p2s_ctx *ctx, int64_t n_frames, int32_t n_kpts_json, int32_t n_max, int32_t dtype,
                       const int32_t *n_persons, const int64_t *offsets, const void *kpts,
                       const p2s_assoc_params *params, double *affinity) {
    int rc = check_assoc(ctx, n_frames, n_kpts_json, n_max, dtype, params);
    if (rc != P2S_OK) return rc;
    if (n_frames == 0) return P2S_OK;
    if (!n_persons || !offsets || !affinity) return fail(P2S_ERR_INVALID_ARG, "null host pointer");
    const int C = ctx->n_cams;
    // operand shapes are checked on the host before anything is launched
    int64_t rows = 0;
    for (int64_t f = 0; f < n_frames; ++f) {
        if (offsets[f] != rows) return fail(P2S_ERR_INVALID_ARG, "offsets[%lld] does not match n_persons", (long long)f);
        int64_t nf = 0;
        for (int c = 0; c < C; ++c) {
            if (n_persons[f * C + c] < 0) return fail(P2S_ERR_INVALID_ARG, "negative person count");
            nf += n_persons[f * C + c];
        }
        if (nf > n_max) return fail(P2S_ERR_INVALID_ARG, "frame %lld has %lld detections > n_max=%d", (long long)f, (long long)nf, n_max);
        rows += nf;
    }
    if (offsets[n_frames] != rows) return fail(P2S_ERR_INVALID_ARG, "offsets[F] does not match n_persons");
    if (rows > 0 && !kpts) return fail(P2S_ERR_INVALID_ARG, "null kpts");
    const size_t elem = dtype == P2S_F32 ? 4 : 8;
    const size_t kp_bytes = std::max<size_t>(16, (size_t)rows * n_kpts_json * 3 * elem);
    const size_t aff_bytes = (size_t)n_frames * n_max * n_max * sizeof(double);
    HIP_TRY(hipSetDevice(ctx->device));
    if ((rc = ctx->in.ensure(kp_bytes)) != P2S_OK) return rc;
    if ((rc = ctx->aux0.ensure((size_t)n_frames * C * 4)) != P2S_OK) return rc;
    if ((rc = ctx->aux1.ensure((size_t)(n_frames + 1) * 8)) != P2S_OK) return rc;
    if ((rc = ctx->q.ensure(aff_bytes)) != P2S_OK) return rc;
    if (rows > 0) HIP_TRY(hipMemcpyAsync(ctx->in.p, kpts, (size_t)rows * n_kpts_json * 3 * elem, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->aux0.p, n_persons, (size_t)n_frames * C * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->aux1.p, offsets, (size_t)(n_frames + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = p2s_associate_device(ctx, n_frames, n_kpts_json, n_max, dtype, (const int32_t *)ctx->aux0.p,
                              (const int64_t *)ctx->aux1.p, ctx->in.p, params, (double *)ctx->q.p);
    if (rc != P2S_OK) return rc;
    HIP_TRY(hipMemcpyAsync(affinity, ctx->q.p, aff_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return P2S_OK;
}

static int check_single(p2s_ctx *ctx, int64_t n_frames, int32_t dtype, const p2s_single_params *p) {
    if (!ctx) return fail(P2S_ERR_INVALID_ARG, "null context");
    if (ctx->n_cams <= 0) return fail(P2S_ERR_NO_CALIB, "p2s_set_calibration has not been called");
    if (!p) return fail(P2S_ERR_INVALID_ARG, "null params");
    if (n_frames < 0 || n_frames > 0x7fffffffLL) return fail(P2S_ERR_INVALID_ARG, "bad frame count");
    if (dtype != P2S_F32 && dtype != P2S_F64) return fail(P2S_ERR_INVALID_ARG, "dtype must be P2S_F32 or P2S_F64");
    if (p->min_cameras < 1) return fail(P2S_ERR_INVALID_ARG, "min_cameras must be >= 1");
    return P2S_OK;
}

int p2s_associate_single_device(p2s_ctx *ctx, int64_t n_frames, int32_t dtype, const int32_t *d_n_persons,
                                const int64_t *d_offsets, const void *d_tracked, const p2s_single_params *params,
                                int32_t *d_comb, double *d_err, double *d_Q) {
    int rc = check_single(ctx, n_frames, dtype, params);
    if (rc != P2S_OK) return rc;
    if (n_frames == 0) return P2S_OK;
    if (!d_n_persons || !d_offsets || !d_tracked || !d_comb || !d_err || !d_Q)
        return fail(P2S_ERR_INVALID_ARG, "null device pointer");
    P2sSingleArgs a{};
    a.n_persons = d_n_persons; a.offsets = d_offsets; a.tracked = d_tracked;
    a.comb = d_comb; a.err = d_err; a.Q = d_Q;
    a.cams = ctx->d_cams; a.binom = ctx->d_binom;
    a.n_frames = n_frames; a.C = ctx->n_cams; a.min_cams = params->min_cameras;
    a.thr = params->reproj_error_threshold; a.lik_thr = params->likelihood_threshold;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(p2s_launch_single(a, dtype, ctx->stream));
    return P2S_OK;
}

int p2s_associate_single_host(p2s_ctx *ctx, int64_t n_frames, int32_t dtype, const int32_t *n_persons,
                              const int64_t *offsets, const void *tracked, const p2s_single_params *params,
                              int32_t *comb, double *err, double *Q) {
    int rc = check_single(ctx, n_frames, dtype, params);
    if (rc != P2S_OK) return rc;
    if (n_frames == 0) return P2S_OK;
    if (!n_persons || !offsets || !comb || !err || !Q) return fail(P2S_ERR_INVALID_ARG, "null host pointer");
    const int C = ctx->n_cams;
    int64_t rows = 0;
    for (int64_t f = 0; f < n_frames; ++f) {       // operand shapes are checked before anything is launched
        if (offsets[f] != rows) return fail(P2S_ERR_INVALID_ARG, "offsets[%lld] does not match n_persons", (long long)f);
        double prod = 1.0;
        for (int c = 0; c < C; ++c) {
            const int32_t n = n_persons[f * C + c];
            if (n < 0 || n > P2S_MAX_PERSONS_PER_CAM)
                return fail(P2S_ERR_INVALID_ARG, "frame %lld camera %d: %d persons outside [0, %d]", (long long)f, c, n, P2S_MAX_PERSONS_PER_CAM);
            rows += n;
            prod *= n > 0 ? n : 1;
        }
        if (prod > (double)P2S_MAX_COMBINATIONS)
            return fail(P2S_ERR_INVALID_ARG, "frame %lld: %.0f person combinations exceed %d", (long long)f, prod, P2S_MAX_COMBINATIONS);
        // worst case of the search (no combination ever gets under the threshold): every combination x every
        // subset of up to (cameras with detections - min_cameras) cameras switched off.  The reference would
        // run just as long; a frame that could keep one wave busy for minutes is refused instead.
        int present = 0;
        for (int c = 0; c < C; ++c) present += n_persons[f * C + c] > 0;
        double subsets = 0.0, binom = 1.0;
        for (int k = 0; k <= present - params->min_cameras; ++k) {
            subsets += binom;
            binom = binom * (present - k) / (k + 1);
        }
        if (prod * subsets > P2S_MAX_SINGLE_SEARCH)
            return fail(P2S_ERR_INVALID_ARG, "frame %lld: up to %.3g (combination, camera subset) evaluations exceed %.3g; raise "
                        "min_cameras_for_triangulation or reduce the detections", (long long)f, prod * subsets, (double)P2S_MAX_SINGLE_SEARCH);
    }
    if (offsets[n_frames] != rows) return fail(P2S_ERR_INVALID_ARG, "offsets[F] does not match n_persons");
    if (rows > 0 && !tracked) return fail(P2S_ERR_INVALID_ARG, "null tracked");
    const size_t elem = dtype == P2S_F32 ? 4 : 8;
    HIP_TRY(hipSetDevice(ctx->device));
    if ((rc = ctx->in.ensure(std::max<size_t>(16, (size_t)rows * 3 * elem))) != P2S_OK) return rc;
    if ((rc = ctx->aux0.ensure((size_t)n_frames * C * 4)) != P2S_OK) return rc;
    if ((rc = ctx->aux1.ensure((size_t)(n_frames + 1) * 8)) != P2S_OK) return rc;
    if ((rc = ctx->mask.ensure((size_t)n_frames * C * 4)) != P2S_OK) return rc;
    if ((rc = ctx->err.ensure((size_t)n_frames * 8)) != P2S_OK) return rc;
    if ((rc = ctx->q.ensure((size_t)n_frames * 24)) != P2S_OK) return rc;
    if (rows > 0) HIP_TRY(hipMemcpyAsync(ctx->in.p, tracked, (size_t)rows * 3 * elem, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->aux0.p, n_persons, (size_t)n_frames * C * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->aux1.p, offsets, (size_t)(n_frames + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    rc = p2s_associate_single_device(ctx, n_frames, dtype, (const int32_t *)ctx->aux0.p, (const int64_t *)ctx->aux1.p,
                                     ctx->in.p, params, (int32_t *)ctx->mask.p, (double *)ctx->err.p, (double *)ctx->q.p);
    if (rc != P2S_OK) return rc;
    HIP_TRY(hipMemcpyAsync(comb, ctx->mask.p, (size_t)n_frames * C * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(err, ctx->err.p, (size_t)n_frames * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(Q, ctx->q.p, (size_t)n_frames * 24, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return P2S_OK;
}

int p2s_butterworth_host(p2s_ctx *ctx, int64_t n_frames, int32_t n_cols, const double *data, int32_t n_coef,
                         const double *b, const double *a, const double *zi, double *out) {
    if (!ctx) return fail(P2S_ERR_INVALID_ARG, "null context");
    if (n_frames < 0 || n_cols < 0) return fail(P2S_ERR_INVALID_ARG, "bad shape: n_frames=%lld n_cols=%d", (long long)n_frames, n_cols);
    if (n_coef < 2 || n_coef > P2S_MAX_FILTER_ORDER + 1)
        return fail(P2S_ERR_INVALID_ARG, "filter with %d coefficients: supported 2..%d", n_coef, P2S_MAX_FILTER_ORDER + 1);
    if (n_frames == 0 || n_cols == 0) return P2S_OK;
    if (!data || !out || !b || !a || !zi) return fail(P2S_ERR_INVALID_ARG, "null pointer");
    if (!(a[0] == 1.0)) return fail(P2S_ERR_INVALID_ARG, "a[0] must be 1 (scipy.signal.butter normalises it)");
    P2sFilterArgs f{};
    f.n_frames = n_frames; f.n_cols = n_cols; f.n_order = n_coef - 1;
    f.padlen = 3 * n_coef;                                  // filtering.py:457
    for (int i = 0; i < n_coef; ++i) { f.b[i] = b[i]; f.a[i] = a[i]; }
    for (int i = 0; i < n_coef - 1; ++i) f.zi[i] = zi[i];
    const size_t bytes = (size_t)n_frames * n_cols * sizeof(double);
    const size_t wbytes = (size_t)(n_frames + 2 * f.padlen) * n_cols * sizeof(double);
    int rc;
    HIP_TRY(hipSetDevice(ctx->device));
    if ((rc = ctx->in.ensure(bytes)) != P2S_OK) return rc;
    if ((rc = ctx->q.ensure(bytes)) != P2S_OK) return rc;
    if ((rc = ctx->aux0.ensure(wbytes)) != P2S_OK) return rc;
    f.in = (const double *)ctx->in.p; f.out = (double *)ctx->q.p; f.work = (double *)ctx->aux0.p;
    HIP_TRY(hipMemcpyAsync(ctx->in.p, data, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(p2s_launch_butter(f, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, ctx->q.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return P2S_OK;
}

int p2s_filter_columns_host(p2s_ctx *ctx, int32_t kind, int64_t n_frames, int32_t n_cols, const double *data,
                            const double *params, int32_t n_params, double *out) {
    if (!ctx) return fail(P2S_ERR_INVALID_ARG, "null context");
    if (n_frames < 0 || n_cols < 0) return fail(P2S_ERR_INVALID_ARG, "bad shape: n_frames=%lld n_cols=%d", (long long)n_frames, n_cols);
    if (n_params < 0 || (n_params > 0 && !params)) return fail(P2S_ERR_INVALID_ARG, "null parameters");
    P2sColFilterArgs f{};
    f.kind = kind; f.n_frames = n_frames; f.n_cols = n_cols;
    switch (kind) {
    case P2S_FILTER_HAMPEL:
        if (n_params != 1) return fail(P2S_ERR_INVALID_ARG, "Hampel filter: params = {n_sigma}");
        f.p[0] = params[0];
        break;
    case P2S_FILTER_GAUSSIAN:
        if (n_params < 1 || n_params % 2 != 1 || n_params > 8191) return fail(P2S_ERR_INVALID_ARG, "Gaussian filter: params = 2 radius + 1 weights");
        f.radius = n_params / 2;
        break;
    case P2S_FILTER_MEDIAN: {
        if (n_params != 1) return fail(P2S_ERR_INVALID_ARG, "median filter: params = {kernel_size}");
        const int k = (int)params[0];
        if ((double)k != params[0] || k < 1 || k % 2 != 1 || k > 1023) return fail(P2S_ERR_INVALID_ARG, "median filter: kernel_size must be odd, 1..1023");
        f.radius = k / 2;
        break;
    }
    case P2S_FILTER_ONE_EURO:
        if (n_params != 4) return fail(P2S_ERR_INVALID_ARG, "one-euro filter: params = {dt, min_cutoff, beta, d_cutoff}");
        for (int i = 0; i < 4; ++i) f.p[i] = params[i];
        if (!(f.p[0] > 0.0)) return fail(P2S_ERR_INVALID_ARG, "one-euro filter: dt must be positive");
        break;
    case P2S_FILTER_KALMAN:
        if (n_params != 4) return fail(P2S_ERR_INVALID_ARG, "Kalman filter: params = {dt, measurement_noise, process_noise, smooth}");
        for (int i = 0; i < 4; ++i) f.p[i] = params[i];
        if (!(f.p[0] > 0.0)) return fail(P2S_ERR_INVALID_ARG, "Kalman filter: dt must be positive");
        break;
    default: return fail(P2S_ERR_INVALID_ARG, "unknown column filter %d", kind);
    }
    if (n_frames == 0 || n_cols == 0) return P2S_OK;
    if (!data || !out) return fail(P2S_ERR_INVALID_ARG, "null pointer");
    const size_t bytes = (size_t)n_frames * n_cols * sizeof(double);
    if (kind == P2S_FILTER_MEDIAN)
        for (size_t i = 0, n = (size_t)n_frames * n_cols; i < n; ++i)
            if (!(data[i] == data[i])) return fail(P2S_ERR_INVALID_ARG, "median filter: the data hold NaN (scipy.signal.medfilt's answer for them is not defined)");
    int rc;
    HIP_TRY(hipSetDevice(ctx->device));
    if ((rc = ctx->in.ensure(bytes)) != P2S_OK) return rc;
    if ((rc = ctx->q.ensure(bytes)) != P2S_OK) return rc;
    f.in = (const double *)ctx->in.p; f.out = (double *)ctx->q.p;
    if (kind == P2S_FILTER_ONE_EURO || kind == P2S_FILTER_KALMAN) {
        if ((rc = ctx->aux0.ensure(kind == P2S_FILTER_KALMAN ? 12 * bytes : bytes)) != P2S_OK) return rc;
        f.work = (double *)ctx->aux0.p;
    }
    if (kind == P2S_FILTER_GAUSSIAN) {
        if ((rc = ctx->aux1.ensure((size_t)n_params * sizeof(double))) != P2S_OK) return rc;
        HIP_TRY(hipMemcpyAsync(ctx->aux1.p, params, (size_t)n_params * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        f.w = (const double *)ctx->aux1.p;
    }
    HIP_TRY(hipMemcpyAsync(ctx->in.p, data, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(p2s_launch_col_filter(f, ctx->stream));
    HIP_TRY(hipMemcpyAsync(out, ctx->q.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return P2S_OK;
}

int p2s_trc_metrics_host(p2s_ctx *ctx, int64_t n_frames, int32_t n_markers, const double *xyz, int32_t n_bones,
                         const int32_t *bones, double *bone_len, double *bone_stats, double *accel, int64_t *missing) {
    if (!ctx) return fail(P2S_ERR_INVALID_ARG, "null context");
    if (n_frames < 0 || n_markers < 0 || n_bones < 0) return fail(P2S_ERR_INVALID_ARG, "bad shape");
    if (n_frames == 0 || (n_markers == 0 && n_bones == 0)) return P2S_OK;
    if (!xyz || (n_bones && (!bones || !bone_len || !bone_stats)) || (n_markers && (!accel || !missing)))
        return fail(P2S_ERR_INVALID_ARG, "null pointer");
    for (int i = 0; i < 2 * n_bones; ++i)
        if (bones[i] < 0 || bones[i] >= n_markers) return fail(P2S_ERR_INVALID_ARG, "bone %d names marker %d of %d", i / 2, bones[i], n_markers);
    const size_t xyz_b = (size_t)n_frames * n_markers * 3 * sizeof(double);
    const size_t len_b = std::max<size_t>(16, (size_t)n_bones * n_frames * sizeof(double));
    const size_t acc_b = std::max<size_t>(16, (size_t)n_markers * (n_frames > 2 ? n_frames - 2 : 0) * sizeof(double));
    int rc;
    HIP_TRY(hipSetDevice(ctx->device));
    if ((rc = ctx->in.ensure(xyz_b)) != P2S_OK) return rc;
    if ((rc = ctx->q.ensure(len_b)) != P2S_OK) return rc;
    if ((rc = ctx->aux0.ensure(acc_b)) != P2S_OK) return rc;
    if ((rc = ctx->aux1.ensure(std::max<size_t>(16, (size_t)n_bones * 8 + (size_t)n_bones * 24 + (size_t)n_markers * 8))) != P2S_OK) return rc;
    P2sMetricsArgs m{};
    m.xyz = (const double *)ctx->in.p;
    m.bone_len = (double *)ctx->q.p;
    m.accel = (double *)ctx->aux0.p;
    unsigned char *aux = (unsigned char *)ctx->aux1.p;
    m.bones = (const int32_t *)aux;
    m.bone_stats = (double *)(aux + (size_t)n_bones * 8);
    m.missing = (int64_t *)(aux + (size_t)n_bones * 8 + (size_t)n_bones * 24);
    m.n_frames = n_frames; m.n_markers = n_markers; m.n_bones = n_bones;
    HIP_TRY(hipMemcpyAsync(ctx->in.p, xyz, xyz_b, hipMemcpyHostToDevice, ctx->stream));
    if (n_bones) HIP_TRY(hipMemcpyAsync(aux, bones, (size_t)n_bones * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(p2s_launch_trc_metrics(m, ctx->stream));
    if (n_bones) {
        HIP_TRY(hipMemcpyAsync(bone_len, m.bone_len, (size_t)n_bones * n_frames * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(bone_stats, m.bone_stats, (size_t)n_bones * 24, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (n_markers) {
        if (n_frames > 2)
            HIP_TRY(hipMemcpyAsync(accel, m.accel, (size_t)n_markers * (n_frames - 2) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipMemcpyAsync(missing, m.missing, (size_t)n_markers * 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return P2S_OK;
}

int p2s_timing_begin(p2s_ctx *ctx) {
    if (!ctx) return fail(P2S_ERR_INVALID_ARG, "null context");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    return P2S_OK;
}

int p2s_timing_end(p2s_ctx *ctx, float *elapsed_ms) {
    if (!ctx || !elapsed_ms) return fail(P2S_ERR_INVALID_ARG, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return P2S_OK;
}

}  // extern "C"
