// Proposal extraction, first half (host side of the C-ABI, include/p2s.h): the per-detection rows of
// person_index_per_cam (personAssociation.py:512-527) for every frame of a trial on host threads.
//
// From the thresholded matchSVT matrix, for every detection (row) the best-matching detection of each camera:
// np.argmax of the camera's block, -1 when the block is empty or has no positive entry.  In the reference this is a
// Python double loop with one np.argmax per (row, camera): ~250 us per frame of 32 detections x 8 cameras, four
// fifths of the whole function.  The second half -- np.unique of the rows, np.argsort of their counts, the
// first-come filter -- stays in NumPy on purpose: np.argsort's order among equal counts is not specified (the
// AVX-512 sorting networks of NumPy 2.x are not stable even for 7 elements) and it decides which person comes
// first in the rewritten JSON files, so the caller runs the very same NumPy calls as the reference.
#include <atomic>
#include <cstdint>
#include <thread>
#include <vector>

#include "p2s.h"

int p2s_set_error(int code, const char *fmt, ...);   // p2s_api.hip

extern "C" int p2s_assoc_argmax_rows(int64_t n_frames, int32_t n_cams, int32_t n_max, const double *affinity,
                                     const int32_t *n_persons, int32_t n_threads, int32_t *rows) {
    if (n_frames < 0 || n_cams < 1 || n_cams > P2S_MAX_CAMS || n_max < 0 || (n_frames > 0 && (!affinity || !n_persons || !rows)))
        return p2s_set_error(P2S_ERR_INVALID_ARG, "bad arguments");
    for (int64_t f = 0; f < n_frames; ++f) {
        int64_t tot = 0;
        for (int c = 0; c < n_cams; ++c) {
            if (n_persons[f * n_cams + c] < 0) return p2s_set_error(P2S_ERR_INVALID_ARG, "negative person count");
            tot += n_persons[f * n_cams + c];
        }
        if (tot > n_max) return p2s_set_error(P2S_ERR_INVALID_ARG, "frame %lld holds more detections than n_max", (long long)f);
    }
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 32) nt = 32;
    if ((int64_t)nt > n_frames / 256 + 1) nt = (int)(n_frames / 256 + 1);
    std::atomic<int64_t> next{0};
    auto work = [&] {
        while (true) {
            const int64_t lo = next.fetch_add(256);
            if (lo >= n_frames) break;
            const int64_t hi = lo + 256 < n_frames ? lo + 256 : n_frames;
            for (int64_t f = lo; f < hi; ++f) {
                const double *aff = affinity + f * (int64_t)n_max * n_max;
                const int32_t *np_ = n_persons + f * n_cams;
                int32_t *out = rows + f * (int64_t)n_max * n_cams;
                int cum[P2S_MAX_CAMS + 1];
                cum[0] = 0;
                for (int c = 0; c < n_cams; ++c) cum[c + 1] = cum[c] + np_[c];
                const int N = cum[n_cams];
                for (int r = 0; r < N; ++r)
                    for (int c = 0; c < n_cams; ++c) {
                        int best = -1;
                        double bv = 0.0;
                        bool has_nan = false;
                        for (int j = cum[c]; j < cum[c + 1]; ++j) {
                            const double v = aff[(int64_t)r * n_max + j];
                            if (v != v) { has_nan = true; if (best < 0 || !(bv != bv)) { best = j - cum[c]; bv = v; } break; }   // np.argmax: first NaN wins
                            if (best < 0 || v > bv) { best = j - cum[c]; bv = v; }                                          // first maximum
                        }
                        // `max(block) > 0` is False for a NaN maximum as well
                        out[(int64_t)r * n_cams + c] = (best >= 0 && !has_nan && bv > 0.0) ? best : -1;
                    }
            }
        }
    };
    if (nt <= 1) work();
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t) pool.emplace_back(work);
        for (auto &th : pool) th.join();
    }
    return P2S_OK;
}


// ---- second half of person_index_per_cam (:528-549) around the one call whose order is not specified ------------------------
// np.unique(rows, axis=0, return_counts=True) is deterministic (the distinct rows in lexicographic order, with their
// multiplicities) and so is everything after np.argsort(counts)[::-1]: the first-come filter (a proposal that reuses, for
// some camera, a person of ANY proposal ranked before it is dropped) and the minimum number of cameras.  Both halves are
// done here for every frame at once; the caller makes the argsort call itself, on the very array the reference would pass.
template <typename F>
static void for_frames(int64_t n_frames, int32_t n_threads, F &&body) {
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if (nt > 32) nt = 32;
    if ((int64_t)nt > n_frames / 256 + 1) nt = (int)(n_frames / 256 + 1);
    std::atomic<int64_t> next{0};
    auto work = [&] {
        while (true) {
            const int64_t lo = next.fetch_add(256);
            if (lo >= n_frames) break;
            const int64_t hi = lo + 256 < n_frames ? lo + 256 : n_frames;
            for (int64_t f = lo; f < hi; ++f) body(f);
        }
    };
    if (nt <= 1) { work(); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) pool.emplace_back(work);
    for (auto &th : pool) th.join();
}

extern "C" int p2s_assoc_unique_rows(int64_t n_frames, int32_t n_cams, int32_t n_max, const int32_t *rows, const int32_t *n_rows,
                                     int32_t n_threads, int32_t *uniq, int64_t *counts, int32_t *n_uniq) {
    if (n_frames < 0 || n_cams < 1 || n_cams > P2S_MAX_CAMS || n_max < 0 || (n_frames > 0 && (!rows || !n_rows || !uniq || !counts || !n_uniq)))
        return p2s_set_error(P2S_ERR_INVALID_ARG, "bad arguments");
    for (int64_t f = 0; f < n_frames; ++f)
        if (n_rows[f] < 0 || n_rows[f] > n_max) return p2s_set_error(P2S_ERR_INVALID_ARG, "frame %lld: %d rows of at most %d", (long long)f, n_rows[f], n_max);
    for_frames(n_frames, n_threads, [&](int64_t f) {
        const int32_t *in = rows + f * (int64_t)n_max * n_cams;
        int32_t *out = uniq + f * (int64_t)n_max * n_cams;
        int64_t *cnt = counts + f * (int64_t)n_max;
        const int n = n_rows[f];
        int idx[256];                                                   // n_max <= 48 detections per frame (p2s_associate_*)
        std::vector<int> big;
        int *order = idx;
        if (n > 256) { big.resize(n); order = big.data(); }
        for (int i = 0; i < n; ++i) order[i] = i;
        auto less = [&](int a, int b) {
            for (int c = 0; c < n_cams; ++c) {
                const int32_t x = in[(int64_t)a * n_cams + c], y = in[(int64_t)b * n_cams + c];
                if (x != y) return x < y;
            }
            return false;
        };
        for (int i = 1; i < n; ++i) {                                   // insertion sort: n is a few dozen
            const int v = order[i];
            int j = i;
            while (j > 0 && less(v, order[j - 1])) { order[j] = order[j - 1]; --j; }
            order[j] = v;
        }
        int u = 0;
        for (int i = 0; i < n; ++i) {
            if (i > 0 && !less(order[i - 1], order[i])) { ++cnt[u - 1]; continue; }
            for (int c = 0; c < n_cams; ++c) out[(int64_t)u * n_cams + c] = in[(int64_t)order[i] * n_cams + c];
            cnt[u++] = 1;
        }
        n_uniq[f] = u;
    });
    return P2S_OK;
}

extern "C" int p2s_assoc_filter_rows(int64_t n_frames, int32_t n_cams, int32_t n_max, const int32_t *uniq, const int32_t *n_uniq,
                                     const int32_t *rank, int32_t min_cams, int32_t n_threads, int32_t *props, int32_t *n_props) {
    if (n_frames < 0 || n_cams < 1 || n_cams > P2S_MAX_CAMS || n_max < 0 || (n_frames > 0 && (!uniq || !n_uniq || !rank || !props || !n_props)))
        return p2s_set_error(P2S_ERR_INVALID_ARG, "bad arguments");
    for (int64_t f = 0; f < n_frames; ++f) {
        if (n_uniq[f] < 0 || n_uniq[f] > n_max) return p2s_set_error(P2S_ERR_INVALID_ARG, "frame %lld: bad row count", (long long)f);
        for (int i = 0; i < n_uniq[f]; ++i)
            if (rank[f * (int64_t)n_max + i] < 0 || rank[f * (int64_t)n_max + i] >= n_uniq[f])
                return p2s_set_error(P2S_ERR_INVALID_ARG, "frame %lld: rank out of range", (long long)f);
    }
    for_frames(n_frames, n_threads, [&](int64_t f) {
        const int32_t *in = uniq + f * (int64_t)n_max * n_cams;
        const int32_t *rk = rank + f * (int64_t)n_max;
        int32_t *out = props + f * (int64_t)n_max * n_cams;
        const int n = n_uniq[f];
        int kept = 0;
        for (int i = 0; i < n; ++i) {
            const int32_t *row = in + (int64_t)rk[i] * n_cams;
            bool reused = false;
            int seen_by = 0;
            for (int c = 0; c < n_cams; ++c) {
                if (row[c] < 0) continue;                               // NaN in the reference: equal to nothing
                ++seen_by;
                for (int k = 0; k < i && !reused; ++k) reused = in[(int64_t)rk[k] * n_cams + c] == row[c];
            }
            if (reused || seen_by < min_cams) continue;
            for (int c = 0; c < n_cams; ++c) out[(int64_t)kept * n_cams + c] = row[c];
            ++kept;
        }
        n_props[f] = kept;
    });
    return P2S_OK;
}
