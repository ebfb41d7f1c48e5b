// p2s_filter.hip -- zero-phase Butterworth filtering of .trc coordinate columns and the trc_evaluate quality metrics
// on gfx950 (the first consumer of the triangulation's output).
//
// p2s_butter_kernel: filtering.py:437-471 (butterworth_filter_1d) for every column at once.  A column is cut into its
// runs of valid samples (not NaN, not 0); a run longer than padlen goes through scipy.signal.filtfilt's algorithm
// (odd extension by padlen samples at either end, forward pass of the IIR filter in direct form II transposed with
// the steady-state initial condition scaled by the first sample, backward pass likewise, extension dropped), every
// other sample is copied.  The recurrence is sequential in time, so the parallel axis is the column: one lane per
// column, consecutive lanes = consecutive columns of the row-major [frame][column] matrix, i.e. coalesced loads
// while the columns' runs coincide (they do after triangulate_all's gap filling).  Latency-bound on the recurrence
// (2 passes x ~6 dependent fp64 operations per sample): a few ms for 100k frames, whatever the number of columns up
// to the chip's 16k resident lanes; HBM traffic is 5 x 8 B per sample and irrelevant.
//
// p2s_trc_metrics_kernel: Utilities/trc_evaluate.py:114-228 -- per-frame bone lengths and second-difference
// magnitudes written out for the host's order statistics, sums for means and standard deviations, missing counts.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "p2s_internal.h"

namespace {

__device__ __forceinline__ bool sample_valid(double v) { return (v == v) && (v != 0.0); }

// One sample of scipy's lfilter (direct form II transposed, a[0] = 1): y = z[0] + b[0] x, then
// z[k] = z[k+1] + b[k+1] x - a[k+1] y.  Same operation order as scipy's C loop and no contraction, so that the result
// matches scipy.signal.filtfilt to rounding.
template <int N>
__device__ __forceinline__ double iir_step(const P2sFilterArgs &a, double (&z)[N], double x) {
#pragma clang fp contract(off)
    const double y = z[0] + a.b[0] * x;
#pragma unroll
    for (int k = 0; k < N - 1; ++k) z[k] = (z[k + 1] + x * a.b[k + 1]) - y * a.a[k + 1];
    z[N - 1] = x * a.b[N] - y * a.a[N];
    return y;
}

template <int N>
__global__ void __launch_bounds__(64) p2s_butter_kernel(const P2sFilterArgs a) {
    const int col = blockIdx.x * 64 + threadIdx.x;
    if (col >= a.n_cols) return;
    const int64_t F = a.n_frames, S = a.n_cols;
    const double *in = a.in + col;
    double *out = a.out + col;
    double *work = a.work + col;
    const int64_t pad = a.padlen;
    int64_t f = 0;
    while (f < F) {
        const double v = in[f * S];
        if (!sample_valid(v)) { out[f * S] = v; ++f; continue; }
        int64_t r = f + 1;                                     // the run [f, r) of valid samples
        while (r < F && sample_valid(in[r * S])) ++r;
        const int64_t L = r - f;
        if (L <= pad) {                                        // filtering.py:466: only runs longer than padlen
            for (int64_t i = f; i < r; ++i) out[i * S] = in[i * S];
            f = r;
            continue;
        }
        // scipy.signal.filtfilt(b, a, x) with its defaults: padtype 'odd', padlen = 3 max(len(a), len(b))
        const double *x = in + f * S;
        const double x0 = x[0], xl = x[(L - 1) * S];
        const int64_t E = L + 2 * pad;
        auto ext = [&](int64_t i) -> double {
            if (i < pad) return 2.0 * x0 - x[(pad - i) * S];
            if (i < pad + L) return x[(i - pad) * S];
            return 2.0 * xl - x[(L - 2 - (i - pad - L)) * S];
        };
        double z[N];
        const double e0 = ext(0);
#pragma unroll
        for (int k = 0; k < N; ++k) z[k] = a.zi[k] * e0;
        for (int64_t i = 0; i < E; ++i) work[i * S] = iir_step<N>(a, z, ext(i));
        const double y0 = work[(E - 1) * S];
#pragma unroll
        for (int k = 0; k < N; ++k) z[k] = a.zi[k] * y0;
        for (int64_t i = E - 1; i >= 0; --i) {
            const double y = iir_step<N>(a, z, work[i * S]);
            if (i >= pad && i < pad + L) out[(f + i - pad) * S] = y;
        }
        f = r;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The other column filters of filtering.py.  Hampel, Gaussian and median look at a fixed window of the INPUT around every
// sample: one thread per (frame, column) element, consecutive threads = consecutive columns of a row (coalesced).  The
// one-euro filter is a recurrence like the Butterworth one: one lane per column.

// hampel_filter (filtering.py:63-85): 7-sample window, median and median absolute deviation; the sample is replaced by
// the median when 0.6745 |x - median| / mad > n_sigma.  np.median of a window that holds a NaN is NaN, the comparison
// with it is false and the sample stays; the first and last window / 2 samples are never looked at.
__device__ __forceinline__ void sort2(double &a, double &b) { const double lo = fmin(a, b), hi = fmax(a, b); a = lo; b = hi; }

__device__ __forceinline__ double median7(double v0, double v1, double v2, double v3, double v4, double v5, double v6) {
    // the 4th smallest of 7 (no NaN among them) by a sorting network
    sort2(v0, v4); sort2(v1, v5); sort2(v2, v6); sort2(v0, v2); sort2(v1, v3); sort2(v4, v6); sort2(v2, v4); sort2(v3, v5);
    sort2(v0, v1); sort2(v2, v3); sort2(v4, v5); sort2(v1, v4); sort2(v3, v6); sort2(v1, v2); sort2(v3, v4); sort2(v5, v6);
    return v3;
}

__global__ void __launch_bounds__(256) p2s_hampel_kernel(const P2sColFilterArgs a) {
#pragma clang fp contract(off)
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = a.n_frames * a.n_cols;
    if (idx >= total) return;
    const int64_t S = a.n_cols, f = idx / S;
    const double x = a.in[idx];
    double y = x;
    if (f >= 3 && f < a.n_frames - 3) {
        const double *p = a.in + idx;
        const double w0 = p[-3 * S], w1 = p[-2 * S], w2 = p[-S], w4 = p[S], w5 = p[2 * S], w6 = p[3 * S];
        const bool any_nan = !(w0 == w0) || !(w1 == w1) || !(w2 == w2) || !(x == x) || !(w4 == w4) || !(w5 == w5) || !(w6 == w6);
        if (!any_nan) {
            const double med = median7(w0, w1, w2, x, w4, w5, w6);
            const double mad = median7(fabs(w0 - med), fabs(w1 - med), fabs(w2 - med), fabs(x - med), fabs(w4 - med), fabs(w5 - med),
                                       fabs(w6 - med));
            if (mad != 0.0) {
                const double z = 0.6745 * (x - med) / mad;
                if (fabs(z) > a.p[0]) y = med;
            }
        }
    }
    a.out[idx] = y;
}

// gaussian_filter_1d (filtering.py:513-529) = scipy.ndimage.correlate1d(col, weights, mode='reflect') with the weights of
// scipy's own _gaussian_kernel1d (computed by the host with the same call): centre tap first, then the pairs from the
// far end inwards, as scipy's loop for a symmetric kernel adds them.  A NaN spreads over its whole neighbourhood.
__global__ void __launch_bounds__(256) p2s_gauss_kernel(const P2sColFilterArgs a) {
#pragma clang fp contract(off)
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = a.n_frames * a.n_cols;
    if (idx >= total) return;
    const int64_t S = a.n_cols, F = a.n_frames, f = idx / S, c = idx - f * S;
    const int r = a.radius;
    auto at = [&](int64_t i) -> double {                       // 'reflect': d c b a | a b c d | d c b a
        while (i < 0 || i >= F) i = (i < 0) ? -i - 1 : 2 * F - 1 - i;
        return a.in[i * S + c];
    };
    double acc = a.in[idx] * a.w[r];
    for (int k = -r; k < 0; ++k) acc += (at(f + k) + at(f - k)) * a.w[k + r];
    a.out[idx] = acc;
}

// median_filter_1d (filtering.py:561-577) = scipy.signal.medfilt = ndimage.rank_filter(rank k / 2, mode='constant'): the
// window is padded with zeros beyond the ends.  The median is the window element with as many smaller ones as its rank
// allows (k is small: 3 to 15 in practice; windows are read through the cache).  Columns with NaN are refused by the host.
__global__ void __launch_bounds__(256) p2s_median_kernel(const P2sColFilterArgs a) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t total = a.n_frames * a.n_cols;
    if (idx >= total) return;
    const int64_t S = a.n_cols, F = a.n_frames, f = idx / S, c = idx - f * S;
    const int r = a.radius, want = r;                           // rank k / 2 of k = 2 r + 1
    auto at = [&](int64_t i) -> double { return (i < 0 || i >= F) ? 0.0 : a.in[i * S + c]; };
    double med = 0.0;
    for (int i = -r; i <= r; ++i) {
        const double v = at(f + i);
        int less = 0, eq = 0;
        for (int j = -r; j <= r; ++j) {
            const double u = at(f + j);
            less += (u < v) ? 1 : 0;
            eq += (u == v) ? 1 : 0;
        }
        if (less <= want && want < less + eq) med = v;
    }
    a.out[idx] = med;
}

// one_euro_filter_1d (filtering.py:87-160): every run of at least two samples that are not NaN goes through the adaptive
// first-order low-pass forwards, and the result through it again backwards; p = {dt, min_cutoff, beta, d_cutoff}.
__device__ __forceinline__ double one_euro_alpha(double dt, double cutoff) {
#pragma clang fp contract(off)
    const double r = 2 * 3.141592653589793 * cutoff * dt;
    return r / (r + 1);
}

__global__ void __launch_bounds__(64) p2s_one_euro_kernel(const P2sColFilterArgs a) {
#pragma clang fp contract(off)
    const int col = blockIdx.x * 64 + threadIdx.x;
    if (col >= a.n_cols) return;
    const int64_t F = a.n_frames, S = a.n_cols;
    const double *in = a.in + col;
    double *out = a.out + col;
    double *work = a.work + col;
    const double dt = a.p[0], min_cutoff = a.p[1], beta = a.p[2], d_cutoff = a.p[3];
    const double alpha_d = one_euro_alpha(dt, d_cutoff);
    int64_t f = 0;
    while (f < F) {
        const double v = in[f * S];
        if (!(v == v)) { out[f * S] = v; ++f; continue; }
        int64_t r = f + 1;
        while (r < F && (in[r * S] == in[r * S])) ++r;
        const int64_t L = r - f;
        if (L < 2) { out[f * S] = v; f = r; continue; }
        double x_prev = v, dx_prev = 0.0;
        work[f * S] = v;
        for (int64_t i = f + 1; i < r; ++i) {                  // forward pass
            const double x = in[i * S];
            const double dx = (x - x_prev) / dt;
            const double dx_hat = alpha_d * dx + (1 - alpha_d) * dx_prev;
            const double alpha = one_euro_alpha(dt, min_cutoff + beta * fabs(dx_hat));
            const double x_hat = alpha * x + (1 - alpha) * x_prev;
            work[i * S] = x_hat;
            x_prev = x_hat; dx_prev = dx_hat;
        }
        x_prev = work[(r - 1) * S]; dx_prev = 0.0;
        out[(r - 1) * S] = x_prev;
        for (int64_t i = r - 2; i >= f; --i) {                 // backward pass over the forward result
            const double x = work[i * S];
            const double dx = (x - x_prev) / dt;
            const double dx_hat = alpha_d * dx + (1 - alpha_d) * dx_prev;
            const double alpha = one_euro_alpha(dt, min_cutoff + beta * fabs(dx_hat));
            const double x_hat = alpha * x + (1 - alpha) * x_prev;
            out[i * S] = x_hat;
            x_prev = x_hat; dx_prev = dx_hat;
        }
        f = r;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// kalman_filter_1d (filtering.py:316-434): constant-acceleration Kalman filter of one coordinate (state position, velocity,
// acceleration; the measurement is the position) over every run of >= 4 samples that are neither NaN nor 0, then the
// Rauch-Tung-Striebel smoother.  The reference builds it from filterpy (KalmanFilter.batch_filter: predict, then update in
// Joseph form, per sample; rts_smoother) -- filterpy is not importable here, so this follows its published algorithm and
// is PARITY UNPINNED (checked against oracle/filtering_ref.py only).  One lane per column; the forward pass leaves its
// means and covariances (12 doubles per sample) in `work` for the smoother.
struct M3 { double m[9]; };
__device__ __forceinline__ M3 mul3(const M3 &A, const M3 &B) {
    M3 C;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
    return C;
}
__device__ __forceinline__ M3 transpose3(const M3 &A) {
    M3 T;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) T.m[3 * i + j] = A.m[3 * j + i];
    return T;
}
__device__ __forceinline__ M3 inverse3(const M3 &A) {
    const double *a = A.m;
    const double c00 = a[4] * a[8] - a[5] * a[7], c01 = a[5] * a[6] - a[3] * a[8], c02 = a[3] * a[7] - a[4] * a[6];
    const double id = 1.0 / (a[0] * c00 + a[1] * c01 + a[2] * c02);
    M3 R;
    R.m[0] = c00 * id; R.m[1] = (a[2] * a[7] - a[1] * a[8]) * id; R.m[2] = (a[1] * a[5] - a[2] * a[4]) * id;
    R.m[3] = c01 * id; R.m[4] = (a[0] * a[8] - a[2] * a[6]) * id; R.m[5] = (a[2] * a[3] - a[0] * a[5]) * id;
    R.m[6] = c02 * id; R.m[7] = (a[1] * a[6] - a[0] * a[7]) * id; R.m[8] = (a[0] * a[4] - a[1] * a[3]) * id;
    return R;
}

__global__ void __launch_bounds__(64) p2s_kalman_kernel(const P2sColFilterArgs a) {
    const int col = blockIdx.x * 64 + threadIdx.x;
    if (col >= a.n_cols) return;
    const int64_t F = a.n_frames, S = a.n_cols;
    const double *in = a.in + col;
    double *out = a.out + col;
    double *work = a.work + (size_t)col * 12;                          // [frame][col][12]
    const int64_t WS = S * 12;
    const double dt = a.p[0], meas = a.p[1], proc = a.p[2];
    const bool smooth = a.p[3] != 0.0;
    const M3 Fm{{1.0, dt, dt * dt / 2, 0.0, 1.0, dt, 0.0, 0.0, 1.0}};                              // :355-359
    const M3 Ft = transpose3(Fm);
    const double var = proc * proc, R = meas * meas;
    const M3 Q{{.25 * dt * dt * dt * dt * var, .5 * dt * dt * dt * var, .5 * dt * dt * var,        // Q_discrete_white_noise(3, dt, var)
                .5 * dt * dt * dt * var, dt * dt * var, dt * var,
                .5 * dt * dt * var, dt * var, var}};
    auto usable = [](double v) { return (v == v) && (v != 0.0); };     // :421
    int64_t f = 0;
    while (f < F) {
        const double v = in[f * S];
        if (!usable(v)) { out[f * S] = v; ++f; continue; }
        int64_t r = f + 1;
        while (r < F && usable(in[r * S])) ++r;
        if (r - f < 4) {                                               // :428: shorter runs stay as they are
            for (int64_t i = f; i < r; ++i) out[i * S] = in[i * S];
            f = r;
            continue;
        }
        // initial state from the first three samples (:343-351), covariance I * measurement_noise (:377)
        const double z0 = in[f * S], z1 = in[(f + 1) * S], z2 = in[(f + 2) * S];
        double x0 = z0, x1 = (z1 - z0) / dt, x2 = ((z2 - z1) / dt - (z1 - z0) / dt) / dt;
        M3 P{{meas, 0.0, 0.0, 0.0, meas, 0.0, 0.0, 0.0, meas}};
        for (int64_t i = f; i < r; ++i) {
            // predict: x = F x, P = F P F^T + Q
            const double p0 = x0 + dt * x1 + (dt * dt / 2) * x2, p1 = x1 + dt * x2, p2 = x2;
            M3 Pp = mul3(mul3(Fm, P), Ft);
#pragma unroll
            for (int k = 0; k < 9; ++k) Pp.m[k] += Q.m[k];
            // update with z (H = [1 0 0]): K = P H^T / (H P H^T + R), x += K (z - x0), P = (I - K H) P (I - K H)^T + K R K^T
            const double y = in[i * S] - p0;
            const double Sinv = 1.0 / (Pp.m[0] + R);
            const double k0 = Pp.m[0] * Sinv, k1 = Pp.m[3] * Sinv, k2 = Pp.m[6] * Sinv;
            x0 = p0 + k0 * y; x1 = p1 + k1 * y; x2 = p2 + k2 * y;
            const M3 IKH{{1.0 - k0, 0.0, 0.0, -k1, 1.0, 0.0, -k2, 0.0, 1.0}};
            P = mul3(mul3(IKH, Pp), transpose3(IKH));
            const double kk[3] = {k0, k1, k2};
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int w = 0; w < 3; ++w) P.m[3 * u + w] += kk[u] * R * kk[w];
            double *wk = work + i * WS;
            wk[0] = x0; wk[1] = x1; wk[2] = x2;
#pragma unroll
            for (int k = 0; k < 9; ++k) wk[3 + k] = P.m[k];
            if (!smooth) out[i * S] = x0;
        }
        if (smooth) {
            // rts_smoother: from the last sample backwards, x[k] += K (x[k+1] - F x[k]), P[k] += K (P[k+1] - Pp) K^T with
            // Pp = F P[k] F^T + Q and K = P[k] F^T Pp^-1
            double n0 = x0, n1 = x1, n2 = x2;                          // smoothed state and covariance of sample k + 1
            M3 Pn = P;
            out[(r - 1) * S] = n0;
            for (int64_t i = r - 2; i >= f; --i) {
                const double *wk = work + i * WS;
                const double c0 = wk[0], c1 = wk[1], c2 = wk[2];
                M3 Pk;
#pragma unroll
                for (int k = 0; k < 9; ++k) Pk.m[k] = wk[3 + k];
                M3 Pp = mul3(mul3(Fm, Pk), Ft);
#pragma unroll
                for (int k = 0; k < 9; ++k) Pp.m[k] += Q.m[k];
                const M3 K = mul3(mul3(Pk, Ft), inverse3(Pp));
                const double d0 = n0 - (c0 + dt * c1 + (dt * dt / 2) * c2), d1 = n1 - (c1 + dt * c2), d2 = n2 - c2;
                n0 = c0 + K.m[0] * d0 + K.m[1] * d1 + K.m[2] * d2;
                n1 = c1 + K.m[3] * d0 + K.m[4] * d1 + K.m[5] * d2;
                n2 = c2 + K.m[6] * d0 + K.m[7] * d1 + K.m[8] * d2;
                M3 D;
#pragma unroll
                for (int k = 0; k < 9; ++k) D.m[k] = Pn.m[k] - Pp.m[k];
                const M3 U = mul3(mul3(K, D), transpose3(K));
#pragma unroll
                for (int k = 0; k < 9; ++k) Pn.m[k] = Pk.m[k] + U.m[k];
                out[i * S] = n0;
            }
        }
        f = r;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// trc_evaluate: one workgroup per bone (blocks [0, n_bones)) or per marker (blocks [n_bones, n_bones + n_markers)).
__global__ void __launch_bounds__(256) p2s_trc_metrics_kernel(const P2sMetricsArgs a) {
    __shared__ double s_sum[256];
    __shared__ unsigned long long s_cnt[256];
    const int tid = threadIdx.x;
    const int64_t F = a.n_frames;
    const int K = a.n_markers;
    if ((int)blockIdx.x < a.n_bones) {
        // compute_bone_lengths (:114-156): |child - parent| per frame, 0 -> NaN; mean and population standard deviation
        // of the valid lengths (two passes over the frames: the deviations are taken from the final mean, as np.nanstd)
        const int bi = blockIdx.x;
        const int p = a.bones[2 * bi], c = a.bones[2 * bi + 1];
        double *len = a.bone_len + (int64_t)bi * F;
        double sum = 0.0;
        unsigned long long cnt = 0;
        for (int64_t f = tid; f < F; f += 256) {
            const double *P = a.xyz + (f * K + p) * 3, *C = a.xyz + (f * K + c) * 3;
            const double dx = C[0] - P[0], dy = C[1] - P[1], dz = C[2] - P[2];
            double l = sqrt(dx * dx + dy * dy + dz * dz);
            if (l == 0.0) l = __builtin_nan("");
            len[f] = l;
            if (l == l) { sum += l; ++cnt; }
        }
        s_sum[tid] = sum; s_cnt[tid] = cnt;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) { s_sum[tid] += s_sum[tid + o]; s_cnt[tid] += s_cnt[tid + o]; }
            __syncthreads();
        }
        const unsigned long long n = s_cnt[0];
        const double mean = n ? s_sum[0] / (double)n : __builtin_nan("");
        __syncthreads();
        double dev = 0.0;
        for (int64_t f = tid; f < F; f += 256) {
            const double l = len[f];
            if (l == l) dev += (l - mean) * (l - mean);
        }
        s_sum[tid] = dev;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) s_sum[tid] += s_sum[tid + o];
            __syncthreads();
        }
        if (tid == 0) {
            a.bone_stats[3 * bi + 0] = mean;
            a.bone_stats[3 * bi + 1] = n ? sqrt(s_sum[0] / (double)n) : __builtin_nan("");
            a.bone_stats[3 * bi + 2] = (double)n;
        }
        return;
    }
    // compute_smoothness (:159-207) and compute_missing_data (:210-238) of one marker
    const int m = blockIdx.x - a.n_bones;
    if (m >= K) return;
    double *acc = a.accel + (int64_t)m * (F > 2 ? F - 2 : 0);
    unsigned long long missing = 0;
    for (int64_t f = tid; f < F; f += 256) {
        const double *p0 = a.xyz + (f * K + m) * 3;
        if (!(p0[0] == p0[0]) || !(p0[1] == p0[1]) || !(p0[2] == p0[2])) ++missing;
        if (f + 2 < F) {
            const double *p1 = p0 + (int64_t)K * 3, *p2 = p1 + (int64_t)K * 3;
            const double ax = p2[0] - 2 * p1[0] + p0[0], ay = p2[1] - 2 * p1[1] + p0[1], az = p2[2] - 2 * p1[2] + p0[2];
            acc[f] = sqrt(ax * ax + ay * ay + az * az);
        }
    }
    s_cnt[tid] = missing;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) s_cnt[tid] += s_cnt[tid + o];
        __syncthreads();
    }
    if (tid == 0) a.missing[m] = (int64_t)s_cnt[0];
}

}  // namespace

hipError_t p2s_launch_butter(const P2sFilterArgs &a, hipStream_t s) {
    const unsigned grid = (unsigned)((a.n_cols + 63) / 64);
    switch (a.n_order) {
    case 1: hipLaunchKernelGGL((p2s_butter_kernel<1>), dim3(grid), dim3(64), 0, s, a); break;
    case 2: hipLaunchKernelGGL((p2s_butter_kernel<2>), dim3(grid), dim3(64), 0, s, a); break;
    case 3: hipLaunchKernelGGL((p2s_butter_kernel<3>), dim3(grid), dim3(64), 0, s, a); break;
    case 4: hipLaunchKernelGGL((p2s_butter_kernel<4>), dim3(grid), dim3(64), 0, s, a); break;
    case 5: hipLaunchKernelGGL((p2s_butter_kernel<5>), dim3(grid), dim3(64), 0, s, a); break;
    case 6: hipLaunchKernelGGL((p2s_butter_kernel<6>), dim3(grid), dim3(64), 0, s, a); break;
    case 7: hipLaunchKernelGGL((p2s_butter_kernel<7>), dim3(grid), dim3(64), 0, s, a); break;
    case 8: hipLaunchKernelGGL((p2s_butter_kernel<8>), dim3(grid), dim3(64), 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t p2s_launch_col_filter(const P2sColFilterArgs &a, hipStream_t s) {
    const int64_t total = a.n_frames * a.n_cols;
    if (total == 0) return hipSuccess;
    const unsigned grid_e = (unsigned)((total + 255) / 256), grid_c = (unsigned)((a.n_cols + 63) / 64);
    switch (a.kind) {
    case P2S_FILTER_HAMPEL: hipLaunchKernelGGL(p2s_hampel_kernel, dim3(grid_e), dim3(256), 0, s, a); break;
    case P2S_FILTER_GAUSSIAN: hipLaunchKernelGGL(p2s_gauss_kernel, dim3(grid_e), dim3(256), 0, s, a); break;
    case P2S_FILTER_MEDIAN: hipLaunchKernelGGL(p2s_median_kernel, dim3(grid_e), dim3(256), 0, s, a); break;
    case P2S_FILTER_ONE_EURO: hipLaunchKernelGGL(p2s_one_euro_kernel, dim3(grid_c), dim3(64), 0, s, a); break;
    case P2S_FILTER_KALMAN: hipLaunchKernelGGL(p2s_kalman_kernel, dim3(grid_c), dim3(64), 0, s, a); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t p2s_launch_trc_metrics(const P2sMetricsArgs &a, hipStream_t s) {
    const unsigned grid = (unsigned)(a.n_bones + a.n_markers);
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(p2s_trc_metrics_kernel, dim3(grid), dim3(256), 0, s, a);
    return hipGetLastError();
}
