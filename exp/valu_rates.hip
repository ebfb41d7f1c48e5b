// Diagnostics: issue cost of the vector instruction classes the triangulation kernels are made of, on gfx950.
// Every CU is filled with WPS waves per SIMD; each wave issues ITER x 32 independent instructions of one class and
// stamps s_memtime around them.  Printed: cycles per instruction and SIMD (= wave cycles / (WPS x instructions)) and the
// wall-clock rate.  Question behind it (round 3): is a 32-bit VALU instruction cheaper than a 64-bit one when several
// waves share the SIMD, i.e. would an fp32 / packed-fp32 screening pass cost less than the fp64 evaluation it replaces?
//   hipcc --offload-arch=gfx950 -O3 exp/valu_rates.hip -o exp/bin/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// 64-bit operands live in d[8] (v pairs), 32-bit in f[8]; 2-wide packed in p[8] (v pairs)
#define I_FMA64(i)   asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dx), "v"(dy));
#define I_FMA64S(i)  asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "s"(sx), "v"(dy));
#define I_MUL64(i)   asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dx));
#define I_ADD64(i)   asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dy));
#define I_FMA32(i)   asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fx), "v"(fy));
#define I_MUL32(i)   asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fx));
#define I_PKFMA32(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dx), "v"(dy));
#define I_PKMUL32(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(dx));
#define I_PKADD32(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(dy));
#define I_MOV32(i)   asm volatile("v_mov_b32 %0, %1" : "=v"(f[i]) : "v"(fx));
#define I_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(fx) : );
#define I_CVT6432(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
#define I_CVT3264(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
#define I_RCP64(i)   asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
#define I_RSQ64(i)   asm volatile("v_rsq_f64 %0, %0" : "+v"(d[i]));
#define I_RCP32(i)   asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
#define I_CMP64(i)   asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(dx) : "vcc");
#define I_CMP32(i)   asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[i]), "v"(fx) : "vcc");
#define I_DPP(i)     asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f[i]));
#define I_AND32(i)   asm volatile("v_and_b32 %0, %0, %1" : "+v"(f[i]) : "v"(fx));
#define I_MAX64(i)   asm volatile("v_max_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dx));
#define I_BPERM(i)   asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(f[i]) : "v"(addr));
#define I_CNDE64(i)  asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fx), "s"(smask));
#define I_CNDVCC64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(f[i]) : "v"(fx));
#define I_CNDSDWA(i) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(f[i]) : "v"(fx));
#define I_CND0(i)    asm volatile("v_cndmask_b32_e64 %0, 0, %0, %1" : "+v"(f[i]) : "s"(smask));
#define I_CNDMIX(i)  asm volatile("v_cndmask_b32_e64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %1, %1, %4, %5" : "+v"(f[i]), "+v"(d[i]) : "v"(fx), "s"(smask), "v"(dx), "v"(dy));
#define I_FMA4(i)    asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dx), "v"(dy));
#define I_MOV64(i)   asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(dx));
#define I_PKMOV(i)   asm volatile("v_pk_mov_b32 %0, %1, %1" : "=v"(d[i]) : "v"(dx));
#define I_ADDU32(i)  asm volatile("v_add_u32 %0, %0, %1" : "+v"(f[i]) : "v"(fx));
#define I_LSHL64(i)  asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(d[i]));
#define I_BFE(i)     asm volatile("v_bfe_u32 %0, %0, 1, 3" : "+v"(f[i]));
#define I_RDLANE(i)  asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(f[i]) : "s20");
#define I_RDFIRST(i) asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(f[i]) : "s20");
#define I_CMPS64(i)  asm volatile("v_cmp_lt_f64_e64 s[20:21], %0, %1" : : "v"(d[i]), "v"(dx) : "s20", "s21");
#define I_CMPCND(i)  asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(d[i]) : "v"(dx), "v"(f[i]), "v"(fx) : "vcc");
#define I_DSRD64(i)  asm volatile("ds_read_b64 %0, %1" : "=v"(d[i]) : "v"(addr)); if (i == 7) asm volatile("s_waitcnt lgkmcnt(0)");
#define I_DSRD128(i) asm volatile("ds_read_b128 %0, %1" : "=v"(q4[i & 1]) : "v"(addr)); if (i == 7) asm volatile("s_waitcnt lgkmcnt(0)");
#define I_MAX32(i)   asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fx));
#define I_MED3(i)    asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fx), "v"(fy));
#define I_CNDPAIR(i) asm volatile("v_cndmask_b32 %0, %0, %3, vcc\n v_cndmask_b32 %1, %1, %3, vcc\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %2, %2, %4, %5" : "+v"(f[i]), "+v"(f[(i + 4) & 7]), "+v"(d[i]) : "v"(fx), "v"(dx), "v"(dy));
#define I_CNDALT(i)  asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fx));
#define I_CNDNODEP(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(f[i]) : "v"(fx), "v"(fy));
#define I_CNDE64ND(i) asm volatile("v_cndmask_b32_e64 %0, %1, %2, vcc" : "=v"(f[i]) : "v"(fx), "v"(fy));
#define I_FMAMIX(i)  asm volatile("v_fma_mix_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fx), "v"(fy));

#define KERNEL(NAME, INSTR)                                                                           \
    __global__ void __launch_bounds__(64) k_##NAME(double *out, unsigned long long *cyc, int iters,    \
                                                   double seed) {                                      \
        __shared__ double lds_buf[256];                                                                \
        lds_buf[threadIdx.x] = seed; lds_buf[threadIdx.x + 64] = seed; __syncthreads();                \
        double d[8];                                                                                   \
        float f[8];                                                                                    \
        for (int i = 0; i < 8; ++i) { d[i] = seed + i + threadIdx.x; f[i] = (float)d[i]; }            \
        double dx = seed * 0.999, dy = seed * 1e-3;                                                    \
        float fx = (float)dx, fy = (float)dy;                                                          \
        double sx = __builtin_amdgcn_readfirstlane((int)iters) * 1e-9 + 0.999;                         \
        int addr = ((threadIdx.x + 1) & 63) * 16;                                                      \
        unsigned long long smask = __builtin_amdgcn_read_exec() ^ (0x5555ull * (unsigned)iters);       \
        smask = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(smask >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)smask); \
        typedef float v4f __attribute__((ext_vector_type(4)));                                         \
        v4f q4[2]; q4[0] = v4f{0, 0, 0, 0}; q4[1] = q4[0]; (void)q4; (void)smask;                       \
        (void)sx; (void)fx; (void)fy; (void)dx; (void)dy; (void)addr;                                  \
        const unsigned long long t0 = __builtin_readcyclecounter();                                    \
        for (int it = 0; it < iters; ++it) {                                                           \
            REP8(INSTR) REP8(INSTR) REP8(INSTR) REP8(INSTR)                                            \
        }                                                                                              \
        const unsigned long long t1 = __builtin_readcyclecounter();                                    \
        double s = 0;                                                                                  \
        for (int i = 0; i < 8; ++i) s += d[i] + f[i];                                                  \
        out[blockIdx.x * 64 + threadIdx.x] = s;                                                        \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                               \
    }

KERNEL(fma64, I_FMA64)
KERNEL(fma64s, I_FMA64S)
KERNEL(mul64, I_MUL64)
KERNEL(add64, I_ADD64)
KERNEL(max64, I_MAX64)
KERNEL(fma32, I_FMA32)
KERNEL(mul32, I_MUL32)
KERNEL(pkfma32, I_PKFMA32)
KERNEL(pkmul32, I_PKMUL32)
KERNEL(pkadd32, I_PKADD32)
KERNEL(mov32, I_MOV32)
KERNEL(cndmask, I_CNDMASK)
KERNEL(cvt6432, I_CVT6432)
KERNEL(cvt3264, I_CVT3264)
KERNEL(rcp64, I_RCP64)
KERNEL(rsq64, I_RSQ64)
KERNEL(rcp32, I_RCP32)
KERNEL(cmp64, I_CMP64)
KERNEL(cmp32, I_CMP32)
KERNEL(dpp, I_DPP)
KERNEL(and32, I_AND32)
KERNEL(bperm, I_BPERM)
KERNEL(fmamix, I_FMAMIX)
KERNEL(cndpair, I_CNDPAIR)
KERNEL(cndalt, I_CNDALT)
KERNEL(cndnodep, I_CNDNODEP)
KERNEL(cnde64nd, I_CNDE64ND)
KERNEL(cnde64, I_CNDE64)
KERNEL(cndvcc64, I_CNDVCC64)
KERNEL(cndsdwa, I_CNDSDWA)
KERNEL(cnd0, I_CND0)
KERNEL(cndmix, I_CNDMIX)
KERNEL(fma4, I_FMA4)
KERNEL(mov64, I_MOV64)
KERNEL(pkmov, I_PKMOV)
KERNEL(addu32, I_ADDU32)
KERNEL(lshl64, I_LSHL64)
KERNEL(bfe, I_BFE)
KERNEL(rdlane, I_RDLANE)
KERNEL(rdfirst, I_RDFIRST)
KERNEL(cmps64, I_CMPS64)
KERNEL(cmpcnd, I_CMPCND)
KERNEL(dsrd64, I_DSRD64)
KERNEL(dsrd128, I_DSRD128)
KERNEL(max32, I_MAX32)
KERNEL(med3, I_MED3)

typedef void (*kern_t)(double *, unsigned long long *, int, double);

static void run(const char *name, kern_t k, double *out, unsigned long long *cyc, unsigned long long *hcyc, int wps, int iters) {
    const int grid = 256 * 4 * wps;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, out, cyc, iters, 1.0000001);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, out, cyc, iters, 1.0000001);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(hcyc, cyc, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < grid; ++i) mean += (double)hcyc[i];
    mean /= grid;
    const double n = (double)iters * 32;
    // s_memtime ticks at 100 MHz on gfx950 (constant-rate counter): convert through the wall time instead
    const double wall_cyc_per_inst = ms * 1e-3 * 2.4e9 / (n * wps);
    printf("%-9s wps %d : %8.3f ms  %6.2f cyc/inst/SIMD at 2.4 GHz  (%.2f at 2.0)   ticks/wave %.0f\n", name, wps, ms,
           wall_cyc_per_inst, wall_cyc_per_inst * 2.0 / 2.4, mean);
}

int main(int argc, char **argv) {
    const int iters = 4000;
    double *out;
    unsigned long long *cyc, *hcyc;
    (void)hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(double));
    (void)hipMalloc(&cyc, 256 * 4 * 8 * sizeof(unsigned long long));
    hcyc = (unsigned long long *)malloc(256 * 4 * 8 * sizeof(unsigned long long));
    struct { const char *n; kern_t k; } ks[] = {
        {"fma64", k_fma64}, {"fma64s", k_fma64s}, {"mul64", k_mul64}, {"add64", k_add64}, {"max64", k_max64},
        {"fma32", k_fma32}, {"mul32", k_mul32}, {"fmamix", k_fmamix}, {"pkfma32", k_pkfma32}, {"pkmul32", k_pkmul32},
        {"pkadd32", k_pkadd32}, {"mov32", k_mov32}, {"cndmask", k_cndmask}, {"and32", k_and32}, {"cvt6432", k_cvt6432},
        {"cvt3264", k_cvt3264}, {"rcp64", k_rcp64}, {"rsq64", k_rsq64}, {"rcp32", k_rcp32}, {"cmp64", k_cmp64},
        {"cmp32", k_cmp32}, {"dpp", k_dpp}, {"bperm", k_bperm},
        {"cnde64", k_cnde64}, {"cndvcc64", k_cndvcc64}, {"cndsdwa", k_cndsdwa}, {"cnd0", k_cnd0}, {"cndmix", k_cndmix}, {"fma4", k_fma4},
        {"mov64", k_mov64}, {"pkmov", k_pkmov}, {"addu32", k_addu32}, {"lshl64", k_lshl64}, {"bfe", k_bfe}, {"rdlane", k_rdlane},
        {"rdfirst", k_rdfirst}, {"cmps64", k_cmps64}, {"cmpcnd", k_cmpcnd}, {"dsrd64", k_dsrd64}, {"dsrd128", k_dsrd128},
        {"max32", k_max32}, {"med3", k_med3},
        {"cndpair", k_cndpair}, {"cndalt", k_cndalt}, {"cndnodep", k_cndnodep}, {"cnde64nd", k_cnde64nd}};
    for (auto &e : ks) {
        if (argc > 1 && strcmp(argv[1], "new") == 0) { if (&e - ks < 42) continue; } else if (argc > 1 && strcmp(argv[1], e.n) != 0) continue;
        for (int wps : {1, 2, 3, 4}) run(e.n, e.k, out, cyc, hcyc, wps, iters);
    }
    return 0;
}
