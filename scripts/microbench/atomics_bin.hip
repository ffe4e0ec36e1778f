// Microbenchmark: cost of binning ~6 M (Gaussian,tile) instances into 2500 tile buckets with global atomics.
// hipcc --offload-arch=gfx950 -O3 atomics_bin.hip -o atomics_bin && ./atomics_bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void count_k(int n, const int4* __restrict__ rects, unsigned* __restrict__ cnt, int gx) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 r = rects[i];
    for (int y = r.y; y < r.w; ++y)
        for (int x = r.x; x < r.z; ++x) atomicAdd(&cnt[y * gx + x], 1u);
}
__global__ void scatter_k(int n, const int4* __restrict__ rects, const unsigned* __restrict__ base,
                          unsigned* __restrict__ cursor, uint2* __restrict__ bucket, int gx) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int4 r = rects[i];
    for (int y = r.y; y < r.w; ++y)
        for (int x = r.x; x < r.z; ++x) {
            int t = y * gx + x;
            unsigned slot = atomicAdd(&cursor[t], 1u);
            bucket[base[t] + slot] = make_uint2(0x3f800000u + i, (unsigned)i);
        }
}
// LDS-privatised count: each workgroup accumulates a chunk of Gaussians into an LDS histogram, then flushes non-zero bins
__global__ void count_lds_k(int n, const int4* __restrict__ rects, unsigned* __restrict__ cnt, int gx, int tiles, int per_block) {
    extern __shared__ unsigned h[];
    for (int t = threadIdx.x; t < tiles; t += blockDim.x) h[t] = 0;
    __syncthreads();
    int begin = blockIdx.x * per_block, end = min(n, begin + per_block);
    for (int i = begin + threadIdx.x; i < end; i += blockDim.x) {
        int4 r = rects[i];
        for (int y = r.y; y < r.w; ++y)
            for (int x = r.x; x < r.z; ++x) atomicAdd(&h[y * gx + x], 1u);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < tiles; t += blockDim.x) if (h[t]) atomicAdd(&cnt[t], h[t]);
}
int main() {
    const int n = 1000000, gx = 50, gy = 50, tiles = gx * gy;
    std::mt19937 rng(1);
    std::vector<int4> rects(n);
    long total = 0;
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& r : rects) {
        float cx = (rng() % 80000) / 100.f, cy = (rng() % 80000) / 100.f, rad = 4.f + fabsf(nd(rng)) * 10.f;
        int x0 = std::max(0, std::min(gx, (int)((cx - rad) / 16))), x1 = std::max(0, std::min(gx, (int)((cx + rad + 15) / 16)));
        int y0 = std::max(0, std::min(gy, (int)((cy - rad) / 16))), y1 = std::max(0, std::min(gy, (int)((cy + rad + 15) / 16)));
        r = make_int4(x0, y0, x1, y1);
        total += (long)(x1 - x0) * (y1 - y0);
    }
    printf("instances %ld (%.2f per Gaussian)\n", total, (double)total / n);
    int4* d_r; unsigned *d_cnt, *d_base, *d_cur; uint2* d_b;
    CK(hipMalloc(&d_r, n * sizeof(int4))); CK(hipMalloc(&d_cnt, tiles * 4)); CK(hipMalloc(&d_base, tiles * 4));
    CK(hipMalloc(&d_cur, tiles * 4)); CK(hipMalloc(&d_b, total * 8));
    CK(hipMemcpy(d_r, rects.data(), n * sizeof(int4), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(d_cnt, 0, tiles * 4));
        CK(hipEventRecord(e0)); count_k<<<(n + 255) / 256, 256>>>(n, d_r, d_cnt, gx); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("count (global atomics, no return): %.1f us  -> %.2f G atomics/s\n", ms * 1e3, total / ms / 1e6);
    }
    std::vector<unsigned> cnt(tiles), base(tiles);
    CK(hipMemcpy(cnt.data(), d_cnt, tiles * 4, hipMemcpyDeviceToHost));
    unsigned acc = 0; for (int t = 0; t < tiles; ++t) { base[t] = acc; acc += cnt[t]; }
    printf("check total %u\n", acc);
    CK(hipMemcpy(d_base, base.data(), tiles * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(d_cur, 0, tiles * 4));
        CK(hipEventRecord(e0)); scatter_k<<<(n + 255) / 256, 256>>>(n, d_r, d_base, d_cur, d_b, gx); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("scatter (returning atomics + 8 B store): %.1f us -> %.2f G/s\n", ms * 1e3, total / ms / 1e6);
    }
    for (int per_block : {2048, 8192, 32768}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(d_cnt, 0, tiles * 4));
            int blocks = (n + per_block - 1) / per_block;
            CK(hipEventRecord(e0)); count_lds_k<<<blocks, 1024, tiles * 4>>>(n, d_r, d_cnt, gx, tiles, per_block); CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("count LDS-privatised per_block=%d blocks=%d: %.1f us\n", per_block, blocks, ms * 1e3);
        }
    }
    return 0;
}
