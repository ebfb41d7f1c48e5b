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
