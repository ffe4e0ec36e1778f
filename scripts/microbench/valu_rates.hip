// Microbenchmark: issue cost (cycles per wave64 instruction) of the VALU instructions the compositor is made of.
// hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
// Each kernel runs ITER iterations of 64 independent-by-16 instructions per wave; waves/SIMD = 1, 2, 4, 8.
// Placement is pinned: 256-thread workgroups (one wave per SIMD of a CU) that each reserve 1/wps of the CU's LDS, 256 x wps
// of them -- every CU holds exactly wps workgroups, every SIMD exactly wps waves, all resident for the whole kernel (round
// 5's first version launched one-wave workgroups and let the dispatcher place them: uneven SIMD loads inflated the figures).
// Two clocks per row: (a) wall time (HIP events) x an ASSUMED 2.4 GHz, and (b) the shader's own cycle counter (s_memtime,
// one tick per shader cycle on gfx950: MI355X_MICROARCH.md "Per-instruction cycle constants") read by every wave around
// its loop -- (b) does not depend on what the chip clocks at under this load; (b) / wall time = the effective clock.
// Output kept under profiles/ (round 5: profiles/r05_valu_rates.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITER = 1024;
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP16_(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define REP16(X) REP16_(X) REP16_(X) REP16_(X) REP16_(X)      // 64 instructions per loop trip: the loop's own scalar instructions are 3 in 67

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float seed, unsigned long long* cyc) {
    extern __shared__ float pin_lds[];
    if (seed == 12345.0f) pin_lds[threadIdx.x] = seed;      // (never true: keeps the reservation)
    float a[16]; f2 p[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; p[i] = (f2){a[i], a[i] + 0.5f}; }
    float s = seed; f2 ps = {seed, seed};
    unsigned long long m = 0;
    const unsigned long long c0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
        if (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %0, %1" : "+v"(a[i]) : "v"(s));
            REP16(X)
#undef X
        } else if (KIND == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %0, %1" : "+v"(p[i]) : "v"(ps));
            REP16(X)
#undef X
        } else if (KIND == 2) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (KIND == 3) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(s));
            REP16(X)
#undef X
        } else if (KIND == 4) {
#define X(i) asm volatile("v_cmp_gt_f32 vcc, %1, %0\n v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[i]) : "v"(s) : "vcc");
            REP16(X)
#undef X
        } else if (KIND == 5) {
#define X(i) asm volatile("v_readlane_b32 s20, %0, 3\n v_add_f32 %0, s20, %0" : "+v"(a[i]) : : "s20");
            REP16(X)
#undef X
        } else if (KIND == 6) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(ps));
            REP16(X)
#undef X
        } else if (KIND == 7) {
#define X(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(s));
            REP16(X)
#undef X
        } else if (KIND == 8) {   // fma with an SGPR operand
#define X(i) asm volatile("v_fma_f32 %0, s20, %0, %1" : "+v"(a[i]) : "v"(s) : "s20");
            REP16(X)
#undef X
        } else if (KIND == 9) {   // v_min + v_cmp (writes sgpr pair)
#define X(i) asm volatile("v_min_f32 %0, %1, %0\n v_cmp_lt_f32 s[20:21], %1, %0" : "+v"(a[i]) : "v"(s) : "s20", "s21");
            REP16(X)
#undef X
        } else if (KIND == 10) {
#define X(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(s));
            REP16(X)
#undef X
        } else if (KIND == 11) {  // compare alone, result to an SGPR pair
#define X(i) asm volatile("v_cmp_lt_f32 s[20:21], %1, %0" : : "v"(a[i]), "v"(s) : "s20", "s21");
            REP16(X)
#undef X
        } else if (KIND == 12) {  // three distinct VGPR sources
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(s), "v"(a[(i + 1) & 15]));
            REP16(X)
#undef X
        } else if (KIND == 13) {  // DPP row shift + add (the scans of the binning walks)
#define X(i) asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (KIND == 14) {  // v_rcp_f32 (transcendental unit)
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            REP16(X)
#undef X
        } else if (KIND == 15) {  // v_cndmask alone
#define X(i) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(s) : "vcc");
            REP16(X)
#undef X
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = c1 - c0;
    float r = 0;
    for (int i = 0; i < 16; ++i) r += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r + (float)m;
}

template <int KIND>
int run(const char* name, int per_inst, float* out, unsigned long long* cyc, unsigned long long* h_cyc) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;
        const size_t lds = (size_t)160 * 1024 / wps - (wps == 1 ? 0 : 256);        // wps workgroups fill a CU's LDS
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        k<KIND><<<blocks, 256, lds>>>(out, 0.001f, cyc);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        k<KIND><<<blocks, 256, lds>>>(out, 0.001f, cyc);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h_cyc, cyc, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost));
        double mean_cyc = 0;
        for (int b = 0; b < blocks * 4; ++b) mean_cyc += (double)h_cyc[b];
        mean_cyc /= blocks * 4;
        const double inst_per_simd = (double)ITER * 64 * per_inst * wps;
        // a wave's loop lasts mean_cyc shader cycles while its SIMD issues the instructions of all wps co-resident waves
        printf("%-28s waves/SIMD %d: %.3f ms  -> %.2f cycles per wave-instruction at an assumed 2.4 GHz | %.2f by the shader "
               "cycle counter (waves busy %.0f %% of the kernel at 2.4 GHz)\n", name, wps, ms, ms * 1e-3 * 2.4e9 / inst_per_simd,
               mean_cyc / inst_per_simd, 100.0 * mean_cyc / (ms * 1e-3 * 2.4e9));
    }
    return 0;
}

int main() {
    float* out; CK(hipMalloc(&out, 256 * 4 * 8 * 64 * 4));
    unsigned long long* cyc; CK(hipMalloc(&cyc, 256 * 4 * 8 * 8));
    unsigned long long* h = (unsigned long long*)malloc(256 * 4 * 8 * 8);
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("# %s, %d CUs, clockRate %d kHz; 64 instructions (16 independent chains) x %d iterations per wave, 256-thread workgroups pinned by LDS\n",
           prop.name, prop.multiProcessorCount, prop.clockRate, ITER);
    run<0>("v_fma_f32", 1, out, cyc, h);
    run<12>("v_fma_f32 3 vgpr sources", 1, out, cyc, h);
    run<8>("v_fma_f32 sgpr operand", 1, out, cyc, h);
    run<1>("v_pk_fma_f32", 1, out, cyc, h);
    run<7>("v_mul_f32", 1, out, cyc, h);
    run<10>("v_add_f32", 1, out, cyc, h);
    run<6>("v_pk_mul_f32", 1, out, cyc, h);
    run<2>("v_exp_f32", 1, out, cyc, h);
    run<14>("v_rcp_f32", 1, out, cyc, h);
    run<3>("v_mov_b32", 1, out, cyc, h);
    run<11>("v_cmp_lt_f32 -> sgpr pair", 1, out, cyc, h);
    run<15>("v_cndmask_b32", 1, out, cyc, h);
    run<4>("v_cmp+v_cndmask (2 inst)", 2, out, cyc, h);
    run<9>("v_min+v_cmp->sgpr (2 inst)", 2, out, cyc, h);
    run<5>("v_readlane+v_add (2 inst)", 2, out, cyc, h);
    run<13>("v_add_u32_dpp row_shr:1", 1, out, cyc, h);
    return 0;
}
