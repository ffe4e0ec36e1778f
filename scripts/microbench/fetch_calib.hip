// Calibration of rocprofv3's FETCH_SIZE on gfx950 for this project's access patterns (MI355X_MICROARCH.md, HBM section:
// "calibrate on a known byte count in your own access pattern").
//   stream_k : wide coalesced read, 16 B per lane, BYTES = n * 16
//   gather_k : 48-byte records fetched by a random permutation index (three 16-B loads per lane), BYTES = n * 48 + n * 4
// hipcc --offload-arch=gfx950 -O3 fetch_calib.hip -o fetch_calib ; rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void stream_k(const float4* __restrict__ a, size_t n, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = a[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void gather_k(const float4* __restrict__ rec, const uint32_t* __restrict__ idx, size_t n, float* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4* r = rec + (size_t)idx[i] * 3;
        const float4 a = r[0], b = r[1], c = r[2];
        acc += a.x + b.y + c.z;
    }
    if (acc == 123.456f) out[0] = acc;
}
int main() {
    const size_t n_stream = (size_t)1 << 26;        // 64 Mi float4 = 1 GiB
    const size_t n_rec = (size_t)1 << 24;           // 16 Mi records = 768 MiB
    float4 *a, *rec; uint32_t* idx; float* out;
    CK(hipMalloc(&a, n_stream * 16)); CK(hipMalloc(&rec, n_rec * 48)); CK(hipMalloc(&idx, n_rec * 4)); CK(hipMalloc(&out, 4));
    CK(hipMemset(a, 1, n_stream * 16)); CK(hipMemset(rec, 1, n_rec * 48));
    std::vector<uint32_t> h(n_rec); std::iota(h.begin(), h.end(), 0u);
    std::mt19937 rng(1); std::shuffle(h.begin(), h.end(), rng);
    CK(hipMemcpy(idx, h.data(), n_rec * 4, hipMemcpyHostToDevice));
    stream_k<<<4096, 256>>>(a, n_stream, out);
    gather_k<<<4096, 256>>>(rec, idx, n_rec, out);
    // locally coherent gather (what the Morton-ordered scene gives): index = i with small shuffles inside blocks of 256
    for (size_t b = 0; b + 256 <= n_rec; b += 256) std::shuffle(h.begin() + b, h.begin() + b + 256, rng);
    std::iota(h.begin(), h.end(), 0u);
    for (size_t b = 0; b + 256 <= n_rec; b += 256) std::shuffle(h.begin() + b, h.begin() + b + 256, rng);
    CK(hipMemcpy(idx, h.data(), n_rec * 4, hipMemcpyHostToDevice));
    gather_k<<<4096, 256>>>(rec, idx, n_rec, out);
    CK(hipDeviceSynchronize());
    printf("expected bytes: stream %zu  gather (records+indices) %zu\n", n_stream * 16, n_rec * 52);
    return 0;
}
