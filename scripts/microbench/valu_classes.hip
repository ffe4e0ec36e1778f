// Calibration of the vector-issue model (round 6, profiles/r06_issue_model.txt): for every instruction kind the hot loops of
// the path are made of, (a) its issue cost in SIMD cycles per wave64 instruction with every SIMD holding 8 (or 4) resident
// waves of independent instructions, and (b) -- when run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 ...` --
// which of the SQ instruction-class counters it is counted under (one kernel NAME per kind: the counter rows name it).
//   hipcc --offload-arch=gfx950 -O3 valu_classes.hip -o valu_classes && ./valu_classes [waves per SIMD = 8]
// Clocks: wall time by HIP events; the shader clock by the waves' own counters (s_memtime ticks per s_memrealtime tick of
// 100 MHz), so "cycles" are SHADER cycles at the clock the chip really ran this load at, and the clock is printed beside them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 1024;
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16_(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define REP64(X) REP16_(X) REP16_(X) REP16_(X) REP16_(X)

#define KERNEL(NAME, BODY)                                                                                                   \
    __global__ __launch_bounds__(256) void NAME(float* out, float seed, unsigned long long* cyc) {                           \
        extern __shared__ float pin_lds[];                                                                                   \
        if (seed == 12345.0f) pin_lds[threadIdx.x] = seed;                                                                   \
        float a[16]; f2 p[16]; unsigned u[16];                                                                               \
        for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; p[i] = (f2){a[i], a[i] + 0.5f}; u[i] = threadIdx.x * 7u + i; } \
        float s = seed; f2 ps = {seed, seed}; unsigned us = threadIdx.x | 3u;                                                \
        const unsigned long long r0 = wall_clock64();                                                                        \
        const unsigned long long c0 = __builtin_readcyclecounter();                                                          \
        for (int it = 0; it < ITER; ++it) { REP64(BODY) }                                                                    \
        const unsigned long long c1 = __builtin_readcyclecounter();                                                          \
        const unsigned long long r1 = wall_clock64();                                                                        \
        if ((threadIdx.x & 63) == 0) { cyc[(blockIdx.x * 4 + threadIdx.x / 64) * 2] = c1 - c0; cyc[(blockIdx.x * 4 + threadIdx.x / 64) * 2 + 1] = r1 - r0; } \
        float r = 0;                                                                                                         \
        for (int i = 0; i < 16; ++i) r += a[i] + p[i].x + p[i].y + (float)u[i];                                              \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                                             \
    }

#define B_FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(s), "v"(a[(i + 1) & 15]));
#define B_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %1, %0, %1" : "+v"(p[i]) : "v"(ps));
#define B_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(ps));
#define B_PKADD(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(ps));
#define B_MUL(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(s));
#define B_ADD(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(s));
#define B_SUB(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(s));
#define B_MIN(i) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a[i]) : "v"(s));
#define B_MAX(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "v"(s));
#define B_MED3(i) asm volatile("v_med3_f32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(s), "v"(a[(i + 1) & 15]));
#define B_EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
#define B_RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define B_SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define B_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(s));
#define B_CNDS(i) asm volatile("v_cndmask_b32 %0, %1, %0, s[20:21]" : "+v"(a[i]) : "v"(s) : "s20", "s21");
#define B_CMPF(i) asm volatile("v_cmp_lt_f32 s[20:21], %1, %0" : : "v"(a[i]), "v"(s) : "s20", "s21");
#define B_CMPU(i) asm volatile("v_cmp_lt_u32 s[20:21], %1, %0" : : "v"(u[i]), "v"(us) : "s20", "s21");
#define B_ADDU(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(u[i]) : "v"(us));
#define B_AND(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(u[i]) : "v"(us));
#define B_LSHL(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[i]));
#define B_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(u[i]));
#define B_MAD24(i) asm volatile("v_mad_u32_u24 %0, %1, %0, %1" : "+v"(u[i]) : "v"(us));
#define B_MULLO(i) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(u[i]) : "v"(us));
#define B_ADD3(i) asm volatile("v_add3_u32 %0, %1, %0, %1" : "+v"(u[i]) : "v"(us));
#define B_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(u[i]) : "v"(us));
#define B_CVTFU(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
#define B_CVTIF(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
#define B_FLOOR(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
#define B_RDLANE(i) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a[i]) : "s20");
#define B_DPPMAX(i) asm volatile("v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(u[i]));
#define B_ANDABS(i) asm volatile("v_max_f32 %0, |%1|, |%0|" : "+v"(a[i]) : "v"(s));

KERNEL(k_v_fma_f32, B_FMA) KERNEL(k_v_pk_fma_f32, B_PKFMA) KERNEL(k_v_pk_mul_f32, B_PKMUL) KERNEL(k_v_pk_add_f32, B_PKADD)
KERNEL(k_v_mul_f32, B_MUL) KERNEL(k_v_add_f32, B_ADD) KERNEL(k_v_sub_f32, B_SUB) KERNEL(k_v_min_f32, B_MIN) KERNEL(k_v_max_f32, B_MAX)
KERNEL(k_v_med3_f32, B_MED3) KERNEL(k_v_exp_f32, B_EXP) KERNEL(k_v_rcp_f32, B_RCP) KERNEL(k_v_sqrt_f32, B_SQRT) KERNEL(k_v_mov_b32, B_MOV)
KERNEL(k_v_cndmask_b32_sgpr, B_CNDS) KERNEL(k_v_cmp_lt_f32_sgpr, B_CMPF) KERNEL(k_v_cmp_lt_u32_sgpr, B_CMPU) KERNEL(k_v_add_u32, B_ADDU)
KERNEL(k_v_and_b32, B_AND) KERNEL(k_v_lshlrev_b32, B_LSHL) KERNEL(k_v_bfe_u32, B_BFE) KERNEL(k_v_mad_u32_u24, B_MAD24)
KERNEL(k_v_mul_lo_u32, B_MULLO) KERNEL(k_v_add3_u32, B_ADD3) KERNEL(k_v_lshl_add_u32, B_LSHLADD) KERNEL(k_v_cvt_f32_u32, B_CVTFU)
KERNEL(k_v_cvt_i32_f32, B_CVTIF) KERNEL(k_v_floor_f32, B_FLOOR) KERNEL(k_v_readlane_b32, B_RDLANE) KERNEL(k_v_max_u32_dpp, B_DPPMAX)
KERNEL(k_v_max_f32_abs, B_ANDABS)

typedef void (*kern_t)(float*, float, unsigned long long*);

static int run(const char* name, kern_t fn, int wps, float* out, unsigned long long* cyc, unsigned long long* h) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int blocks = 256 * wps;
    const size_t lds = (size_t)160 * 1024 / wps - (wps == 1 ? 0 : 256);        // wps workgroups fill a CU's LDS: pinned placement
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    fn<<<blocks, 256, lds>>>(out, 0.001f, cyc);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    fn<<<blocks, 256, lds>>>(out, 0.001f, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h, cyc, (size_t)blocks * 4 * 16, hipMemcpyDeviceToHost));
    double cs = 0, rs = 0;
    for (int b = 0; b < blocks * 4; ++b) { cs += (double)h[2 * b]; rs += (double)h[2 * b + 1]; }
    const double mhz = cs / rs * 100.0;                                          // shader ticks per 100 MHz tick
    const double inst_per_simd = (double)ITER * 64 * wps;
    printf("%-24s waves/SIMD %d  %8.3f ms  shader clock %6.0f MHz  %5.2f shader cycles per wave-instruction  (%5.2f at a nominal 2.4 GHz)\n",
           name, wps, ms, mhz, ms * 1e-3 * mhz * 1e6 / inst_per_simd, ms * 1e-3 * 2.4e9 / inst_per_simd);
    return 0;
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 8;
    float* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    unsigned long long* cyc; CK(hipMalloc(&cyc, 256 * 8 * 4 * 16));
    unsigned long long* h = (unsigned long long*)malloc(256 * 8 * 4 * 16);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("# %s, %d CUs, clockRate %d kHz; 64 instructions (16 independent chains) x %d iterations per wave, %d waves per SIMD (256-thread workgroups pinned by LDS)\n",
           prop.name, prop.multiProcessorCount, prop.clockRate, ITER, wps);
#define RUN(K) run(&#K[2], K, wps, out, cyc, h);
    RUN(k_v_fma_f32) RUN(k_v_pk_fma_f32) RUN(k_v_pk_mul_f32) RUN(k_v_pk_add_f32) RUN(k_v_mul_f32) RUN(k_v_add_f32) RUN(k_v_sub_f32)
    RUN(k_v_min_f32) RUN(k_v_max_f32) RUN(k_v_max_f32_abs) RUN(k_v_med3_f32) RUN(k_v_exp_f32) RUN(k_v_rcp_f32) RUN(k_v_sqrt_f32) RUN(k_v_mov_b32)
    RUN(k_v_cndmask_b32_sgpr) RUN(k_v_cmp_lt_f32_sgpr) RUN(k_v_cmp_lt_u32_sgpr) RUN(k_v_add_u32) RUN(k_v_and_b32) RUN(k_v_lshlrev_b32)
    RUN(k_v_bfe_u32) RUN(k_v_mad_u32_u24) RUN(k_v_mul_lo_u32) RUN(k_v_add3_u32) RUN(k_v_lshl_add_u32) RUN(k_v_cvt_f32_u32)
    RUN(k_v_cvt_i32_f32) RUN(k_v_floor_f32) RUN(k_v_readlane_b32) RUN(k_v_max_u32_dpp)
    return 0;
}
