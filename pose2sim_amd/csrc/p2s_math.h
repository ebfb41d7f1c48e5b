// fp64 reciprocal / square-root helpers shared by the kernels: hardware seed (v_rcp_f64, v_rsq_f64)
// plus Newton steps, ~1 ulp, about half the cost of the IEEE division / sqrt sequences.
#ifndef P2S_MATH_H
#define P2S_MATH_H

#include <hip/hip_runtime.h>

// 1/d: 0 -> inf, inf -> 0 and NaN pass through the seed unchanged.
__device__ __forceinline__ double p2s_rcp(double d) {
    const double r0 = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r0, 1.0);
    double r = fma(r0, e, r0);
    e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    return (e == e) ? r : r0;
}

// 1/sqrt(t) for 0 < t < inf.
__device__ __forceinline__ double p2s_rsqrt(double t) {
    double r = __builtin_amdgcn_rsq(t);
    double e = fma(-t * r, r, 1.0);
    r = fma(0.5 * r, e, r);
    e = fma(-t * r, r, 1.0);
    r = fma(0.5 * r, e, r);
    return r;
}

// sqrt(s), s >= 0 (0 and inf are returned as they are).
__device__ __forceinline__ double p2s_sqrt(double s) {
    const double y = __builtin_amdgcn_rsq(s);
    double g = s * y;
    double h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, s);
    g = fma(d, h, g);
    return (s == 0.0 || s == __builtin_huge_val()) ? s : g;
}

#endif
