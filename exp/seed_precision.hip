// Diagnostics: relative error of the v_rcp_f64 / v_rsq_f64 hardware seeds on gfx950, and of the seeds after one
// Newton step (decides how many steps p2s_tri_dev.h's fast_rsqrt / fast_rcp need).
//   hipcc --offload-arch=gfx950 -O3 exp/seed_precision.hip -o gpurun_out/seed_precision && gpurun_out/seed_precision
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void probe(const double *x, double *rcp0, double *rcp1, double *rsq0, double *rsq1, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double d = x[i];
    double r = __builtin_amdgcn_rcp(d);
    rcp0[i] = r;
    rcp1[i] = fma(r, fma(-d, r, 1.0), r);
    double s = __builtin_amdgcn_rsq(d);
    rsq0[i] = s;
    double e = fma(-d * s, s, 1.0);
    rsq1[i] = fma(0.5 * s, e, s);
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n);
    unsigned long long st = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        const double u = (double)(st >> 11) / 9007199254740992.0;
        x[i] = std::ldexp(1.0 + u, (int)(st % 80) - 40);
    }
    double *dx, *a, *b, *c, *d;
    hipMalloc(&dx, n * 8); hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&c, n * 8); hipMalloc(&d, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, dx, a, b, c, d, n);
    std::vector<double> ha(n), hb(n), hc(n), hd(n);
    hipMemcpy(ha.data(), a, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), b, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hc.data(), c, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hd.data(), d, n * 8, hipMemcpyDeviceToHost);
    double m[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const long double t = x[i];
        const long double r = 1.0L / t, s = 1.0L / sqrtl(t);
        m[0] = fmax(m[0], (double)fabsl((ha[i] - r) / r)); m[1] = fmax(m[1], (double)fabsl((hb[i] - r) / r));
        m[2] = fmax(m[2], (double)fabsl((hc[i] - s) / s)); m[3] = fmax(m[3], (double)fabsl((hd[i] - s) / s));
    }
    printf("max rel err: rcp seed %.3e  rcp+1 Newton %.3e  rsq seed %.3e  rsq+1 Newton %.3e\n", m[0], m[1], m[2], m[3]);
    return 0;
}
