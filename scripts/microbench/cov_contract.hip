// Microbenchmark behind DESIGN.md section 2's decision: the EWA covariance contractions of the preprocess
//     T = J W (2x3 . 3x3),   U = T Sigma (2x3 . 3x3),   cov2D = U T^T (2x2)
// as (a) the VALU fmaf chains the kernel uses -- one Gaussian per lane -- and (b) f32 matrix-core instructions,
// v_mfma_f32_4x4x1_16b_f32: sixteen independent 4x4 blocks per instruction, one Gaussian per 4-lane quad.
//
// (b) is laid out so that NO lane shuffle is needed between the three products (each is computed in the transposed
// form whose D fragment is exactly the next product's A or B fragment):
//     D1 = W^T J^T   -> lane j, register i = T[j][i]
//     D2 = Sigma T^T -> lane j, register i = U[j][i]          (A = row i of Sigma, B = D1 registers)
//     D3 = U T^T     -> lane j, register i = cov[i][j]        (A = D2 registers,   B = D1 registers)
// Both forms are checked BITWISE against each other (f32 MFMA == k-ordered fmaf chain, MI355X_MICROARCH.md), then timed
// on register-resident data: the issue cost of the arithmetic alone, no memory traffic.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off cov_contract.hip -o cov_contract && ./cov_contract
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 2048;

struct Gauss { float j00, j02, j11, j12; float S[6]; };   // Jacobian entries and the symmetric 3D covariance

__device__ __forceinline__ float dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
    return fmaf(a2, b2, fmaf(a1, b1, a0 * b0));
}

// (a) one Gaussian per lane; W = the view's rotation (wave-uniform, in SGPRs)
__device__ __forceinline__ void cov_valu(const Gauss& g, const float* W, float& cxx, float& cxy, float& cyy) {
    // T = J W with J = [[j00, 0, j02], [0, j11, j12]], all three k terms spelled out (what the MFMA evaluates)
    float T0[3], T1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        T0[c] = dot3(g.j00, W[0 * 3 + c], 0.0f, W[1 * 3 + c], g.j02, W[2 * 3 + c]);
        T1[c] = dot3(0.0f, W[0 * 3 + c], g.j11, W[1 * 3 + c], g.j12, W[2 * 3 + c]);
    }
    const float S[3][3] = {{g.S[0], g.S[1], g.S[2]}, {g.S[1], g.S[3], g.S[4]}, {g.S[2], g.S[4], g.S[5]}};
    float U0[3], U1[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {   // U[j][i] = sum_k Sigma[i][k] T[j][k]
        U0[c] = dot3(S[c][0], T0[0], S[c][1], T0[1], S[c][2], T0[2]);
        U1[c] = dot3(S[c][0], T1[0], S[c][1], T1[1], S[c][2], T1[2]);
    }
    cxx = dot3(U0[0], T0[0], U0[1], T0[1], U0[2], T0[2]);
    cxy = dot3(U0[0], T1[0], U0[1], T1[1], U0[2], T1[2]);
    cyy = dot3(U1[0], T1[0], U1[1], T1[1], U1[2], T1[2]);
}

// (b) one Gaussian per quad; lane q = lane & 3 plays row / column q of every 4x4 block
__device__ __forceinline__ void cov_mfma(const float Wcol[3] /* W[k][q] */, const float Jrow[3] /* J[q][k], 0 for q > 1 */,
                                         const float Srow[3] /* Sigma[q][k], 0 for q = 3 */, f4& D3) {
    f4 D1 = {0.f, 0.f, 0.f, 0.f}, D2 = D1;
    D3 = D1;
    // D1[i][j] = sum_k W^T[i][k] J^T[k][j] = sum_k W[k][i] J[j][k]:  A: lane i gives W[k][i];  B: lane j gives J[j][k]
#pragma unroll
    for (int k = 0; k < 3; ++k) D1 = __builtin_amdgcn_mfma_f32_4x4x1f32(Wcol[k], Jrow[k], D1, 0, 0, 0);
    // D2[i][j] = sum_k Sigma[i][k] T[j][k]:  A: lane i gives Sigma[i][k];  B: lane j gives T[j][k] = its D1 register k
    D2 = __builtin_amdgcn_mfma_f32_4x4x1f32(Srow[0], D1[0], D2, 0, 0, 0);
    D2 = __builtin_amdgcn_mfma_f32_4x4x1f32(Srow[1], D1[1], D2, 0, 0, 0);
    D2 = __builtin_amdgcn_mfma_f32_4x4x1f32(Srow[2], D1[2], D2, 0, 0, 0);
    // D3[i][j] = sum_k U[i][k] T[j][k]:  A: lane i gives U[i][k] = its D2 register k;  B: lane j gives its D1 register k
    D3 = __builtin_amdgcn_mfma_f32_4x4x1f32(D2[0], D1[0], D3, 0, 0, 0);
    D3 = __builtin_amdgcn_mfma_f32_4x4x1f32(D2[1], D1[1], D3, 0, 0, 0);
    D3 = __builtin_amdgcn_mfma_f32_4x4x1f32(D2[2], D1[2], D3, 0, 0, 0);
}

// ---- correctness: both forms on the same Gaussians, results to memory
__global__ void check_valu(const Gauss* __restrict__ g, const float* __restrict__ W, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float Wl[9];
    for (int k = 0; k < 9; ++k) Wl[k] = W[k];
    cov_valu(g[i], Wl, out[3 * i], out[3 * i + 1], out[3 * i + 2]);
}

__device__ __forceinline__ void quad_operands(const Gauss& g, const float* W, int q, float Wcol[3], float Jrow[3], float Srow[3]) {
    const float J[4][3] = {{g.j00, 0.f, g.j02}, {0.f, g.j11, g.j12}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    const float S[4][3] = {{g.S[0], g.S[1], g.S[2]}, {g.S[1], g.S[3], g.S[4]}, {g.S[2], g.S[4], g.S[5]}, {0.f, 0.f, 0.f}};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Wcol[k] = q < 3 ? W[k * 3 + q] : 0.f;
        Jrow[k] = J[q][k];
        Srow[k] = S[q][k];
    }
}

__global__ void check_mfma(const Gauss* __restrict__ g, const float* __restrict__ W, float* __restrict__ out, int n) {
    const int lane = threadIdx.x & 63, q = lane & 3;
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 2;      // Gaussian of this quad
    float Wl[9];
    for (int k = 0; k < 9; ++k) Wl[k] = W[k];
    Gauss gg;
    memset(&gg, 0, sizeof(gg));
    if (i < n) gg = g[i];
    float Wcol[3], Jrow[3], Srow[3];
    quad_operands(gg, Wl, q, Wcol, Jrow, Srow);
    f4 D3;
    cov_mfma(Wcol, Jrow, Srow, D3);
    // lane j register i = cov[i][j]
    if (i < n) {
        if (q == 0) out[3 * i] = D3[0];
        if (q == 1) { out[3 * i + 1] = D3[0]; out[3 * i + 2] = D3[1]; }
    }
}

// ---- timing: ITER dependent rounds on register data (outputs nudged back into the inputs)
__global__ __launch_bounds__(256) void time_valu(const Gauss* __restrict__ g, const float* __restrict__ W, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float Wl[9];
    for (int k = 0; k < 9; ++k) Wl[k] = W[k];
    Gauss gg = g[i & 1023];
    float acc = 0.f;
    for (int it = 0; it < ITER; ++it) {
        float a, b, c;
        cov_valu(gg, Wl, a, b, c);
        gg.j02 = fmaf(1.0e-9f, a, gg.j02);      // keeps the chain dependent; one extra fma per round in both forms
        acc += b + c;
    }
    out[i] = acc;
}

__global__ __launch_bounds__(256) void time_mfma(const Gauss* __restrict__ g, const float* __restrict__ W, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, q = threadIdx.x & 3;
    float Wl[9];
    for (int k = 0; k < 9; ++k) Wl[k] = W[k];
    const Gauss gg = g[(t >> 2) & 1023];
    float Wcol[3], Jrow[3], Srow[3];
    quad_operands(gg, Wl, q, Wcol, Jrow, Srow);
    float acc = 0.f;
    for (int it = 0; it < ITER; ++it) {
        f4 D3;
        cov_mfma(Wcol, Jrow, Srow, D3);
        Jrow[2] = fmaf(1.0e-9f, D3[0], Jrow[2]);
        acc += D3[1];
    }
    out[t] = acc;
}

int main() {
    const int n = 1 << 16;
    std::vector<Gauss> h(n);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
    for (auto& g : h) {
        g.j00 = 900.f + 200.f * rnd(); g.j11 = 900.f + 200.f * rnd(); g.j02 = -300.f + 600.f * rnd(); g.j12 = -300.f + 600.f * rnd();
        float M[3][3];
        for (auto& r : M) for (auto& v : r) v = 0.02f * (rnd() - 0.5f);
        int k = 0;
        for (int a = 0; a < 3; ++a) for (int b = a; b < 3; ++b) g.S[k++] = M[a][0] * M[b][0] + M[a][1] * M[b][1] + M[a][2] * M[b][2];
    }
    const float Wh[9] = {0.36f, 0.48f, -0.8f, -0.8f, 0.6f, 0.0f, 0.48f, 0.64f, 0.6f};   // asymmetric rotation
    Gauss* dg; float *dW, *o1, *o2;
    CK(hipMalloc(&dg, n * sizeof(Gauss))); CK(hipMalloc(&dW, sizeof(Wh)));
    CK(hipMalloc(&o1, (size_t)n * 3 * 4)); CK(hipMalloc(&o2, (size_t)n * 3 * 4));
    CK(hipMemcpy(dg, h.data(), n * sizeof(Gauss), hipMemcpyHostToDevice)); CK(hipMemcpy(dW, Wh, sizeof(Wh), hipMemcpyHostToDevice));
    check_valu<<<n / 256, 256>>>(dg, dW, o1, n);
    check_mfma<<<n * 4 / 256, 256>>>(dg, dW, o2, n);
    CK(hipDeviceSynchronize());
    std::vector<float> r1((size_t)n * 3), r2((size_t)n * 3);
    CK(hipMemcpy(r1.data(), o1, r1.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r2.data(), o2, r2.size() * 4, hipMemcpyDeviceToHost));
    size_t diff = 0;
    for (size_t i = 0; i < r1.size(); ++i) diff += memcmp(&r1[i], &r2[i], 4) != 0;
    printf("bitwise check, %d Gaussians x 3 outputs: %zu differ (cov[0] of Gaussian 0: valu %.9g  mfma %.9g)\n", n, diff, r1[0], r2[0]);

    // timing: the same number of Gaussians per launch in both forms; one wave handles 64 (valu) or 16 (mfma) of them
    const int gauss = 256 * 1024 * 8;
    float* ot; CK(hipMalloc(&ot, (size_t)gauss * 4 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_v = 0, ms_m = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0)); time_valu<<<gauss / 256, 256>>>(dg, dW, ot); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_v, e0, e1));
        CK(hipEventRecord(e0)); time_mfma<<<gauss * 4 / 256, 256>>>(dg, dW, ot); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms_m, e0, e1));
    }
    const double evals = (double)gauss * ITER;
    printf("VALU fmaf chains, 1 Gaussian / lane : %8.3f ms  -> %7.1f G contractions/s\n", ms_v, evals / ms_v * 1e-6);
    printf("MFMA 4x4x1 x16,   1 Gaussian / quad : %8.3f ms  -> %7.1f G contractions/s  (%.2fx the VALU time)\n", ms_m,
           evals / ms_m * 1e-6, ms_m / ms_v);
    return diff != 0;
}
