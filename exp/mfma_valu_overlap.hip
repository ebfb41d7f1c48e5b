// Diagnostics: do fp64 MFMA (v_mfma_f64_16x16x4_f64) and fp64 VALU FMA share an execution resource on gfx950?
// Four kernels, every CU filled with 12 waves per CU (3 per SIMD) of 64 lanes:
//   valu  : N dependent-free v_fma_f64 per wave (8 independent chains)
//   mfma  : M v_mfma_f64_16x16x4_f64 per wave (4 independent accumulators)
//   both  : the same wave issues N FMAs and M MFMAs interleaved
//   split : odd waves run `valu`, even waves run `mfma`
// If the pipes are separate, both ~ max(valu, mfma); if shared, both ~ valu + mfma.
//   hipcc --offload-arch=gfx950 -O3 exp/mfma_valu_overlap.hip -o exp/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(64) k(double *out, int iters, double seed) {
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
    v4d acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = v4d{0, 0, 0, 0};
    const double x = seed * 0.5, y = seed * 0.25;
    const bool do_valu = MODE == 0 || MODE == 2 || (MODE == 3 && (blockIdx.x & 1));
    const bool do_mfma = MODE == 1 || MODE == 2 || (MODE == 3 && !(blockIdx.x & 1));
    for (int it = 0; it < iters; ++it) {
        if (do_valu) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __builtin_fma(a[i], x, y);      // 32 FMAs
        }
        if (do_mfma) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[i], 0, 0, 0);   // 4 MFMAs
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE>
float run(double *out, int grid, int iters) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, iters, 1.0000001);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, iters, 1.0000001);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int grid = 256 * 12, iters = 20000;
    double *out;
    (void)hipMalloc(&out, grid * 64 * sizeof(double));
    const float t0 = run<0>(out, grid, iters), t1 = run<1>(out, grid, iters), t2 = run<2>(out, grid, iters), t3 = run<3>(out, grid, iters);
    const double fmas = (double)grid * iters * 32, mfmas = (double)grid * iters * 4;
    printf("valu : %.3f ms  -> %.2f cycles/FMA/SIMD at 2.4 GHz (%.1f TF/s)\n", t0, t0 * 1e-3 * 2.4e9 / (fmas / 1024), fmas * 128 / (t0 * 1e-3) / 1e12);
    printf("mfma : %.3f ms  -> %.2f cycles/MFMA/SIMD (%.1f TF/s)\n", t1, t1 * 1e-3 * 2.4e9 / (mfmas / 1024), mfmas * 2048 / (t1 * 1e-3) / 1e12);
    printf("both : %.3f ms  (sum %.3f, max %.3f)\n", t2, t0 + t1, t0 > t1 ? t0 : t1);
    printf("split: %.3f ms  (half the waves each: sum/2 %.3f, max/2 %.3f)\n", t3, (t0 + t1) / 2, (t0 > t1 ? t0 : t1) / 2);
    return 0;
}
