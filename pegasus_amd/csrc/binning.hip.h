// binning.hip.h -- offset scan, (tile,depth) instance emission, tile ranges.
// SURVEY.md section 8a rows a6 (scan), a7 (emission), a9 (ranges).  Integer work: bit-exact.
#pragma once
#include "pgr_common.h"
#include "preprocess.hip.h"

namespace pgr {

// counters[0] = total instances (uint32), counters[1] = overflow flag
constexpr int SCAN_THREADS = 1024;

// One workgroup turns block_sums[n_blocks] into an EXCLUSIVE prefix (in place) and publishes
// the grand total.  n_blocks = ceil(N/256) <= ~20 k even at 5 M Gaussians: a single CU sweeps
// it in a few microseconds, cheaper than a multi-launch scan.
__global__ __launch_bounds__(SCAN_THREADS) void scan_block_sums_kernel(uint32_t* __restrict__ block_sums, int n_blocks,
                                                                       uint32_t* __restrict__ counters,
                                                                       uint32_t max_instances) {
    __shared__ uint32_t wave_tot[SCAN_THREADS / WAVE];
    __shared__ uint32_t carry_s;
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    unsigned long long total64 = 0;
    for (int base = 0; base < n_blocks; base += SCAN_THREADS) {
        const int idx = base + threadIdx.x;
        const uint32_t v = idx < n_blocks ? block_sums[idx] : 0u;
        // inclusive wave scan
        uint32_t s = v;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const uint32_t t = __shfl_up(s, d, WAVE);
            if (lane >= d) s += t;
        }
        if (lane == WAVE - 1) wave_tot[wid] = s;
        __syncthreads();
        uint32_t wave_prefix = 0;
        for (int w = 0; w < wid; ++w) wave_prefix += wave_tot[w];
        const uint32_t carry = carry_s;
        if (idx < n_blocks) block_sums[idx] = carry + wave_prefix + s - v;
        __syncthreads();
        if (threadIdx.x == SCAN_THREADS - 1) carry_s = carry + wave_prefix + s;
        __syncthreads();
    }
    (void)total64;
    if (threadIdx.x == 0) {
        counters[0] = carry_s;
        counters[1] = carry_s > max_instances ? 1u : 0u;
    }
}

// Per workgroup: local exclusive scan of tiles_touched + block prefix -> inclusive offsets[],
// then every Gaussian writes its (tile<<32 | depth bits, idx) pairs, tile row-major.
__global__ __launch_bounds__(PRE_BLOCK) void emit_kernel(int n, const CameraDev* __restrict__ camp,
                                                         const float2* __restrict__ xy, const float* __restrict__ depth,
                                                         const int32_t* __restrict__ radii,
                                                         const uint32_t* __restrict__ tiles_touched,
                                                         const uint32_t* __restrict__ block_prefix,
                                                         const uint32_t* __restrict__ counters,
                                                         uint32_t* __restrict__ offsets, uint64_t* __restrict__ keys,
                                                         uint32_t* __restrict__ vals) {
    __shared__ uint32_t wave_tot[PRE_BLOCK / WAVE];
    const int i = blockIdx.x * PRE_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    const uint32_t tt = i < n ? tiles_touched[i] : 0u;
    uint32_t s = tt;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        const uint32_t t = __shfl_up(s, d, WAVE);
        if (lane >= d) s += t;
    }
    if (lane == WAVE - 1) wave_tot[wid] = s;
    __syncthreads();
    uint32_t prefix = block_prefix[blockIdx.x];
    for (int w = 0; w < wid; ++w) prefix += wave_tot[w];
    const uint32_t incl = prefix + s;
    if (i < n) offsets[i] = incl;
    if (counters[1]) return;  // overflow: the host reports it, nothing may be written past the buffers
    if (tt == 0) return;

    const CameraDev& cam = *camp;
    const float2 p = xy[i];
    const TileRect r = tile_rect(p.x, p.y, radii[i], cam.grid_x, cam.grid_y);
    const uint64_t dbits = (uint64_t)__float_as_uint(depth[i]);
    uint32_t off = incl - tt;
    for (int y = r.miny; y < r.maxy; ++y)
        for (int x = r.minx; x < r.maxx; ++x) {
            keys[off] = ((uint64_t)(uint32_t)(y * cam.grid_x + x) << 32) | dbits;
            vals[off] = (uint32_t)i;
            ++off;
        }
}

// ranges[t] = [start,end) of tile t in the sorted key list; ranges is zero-filled beforehand.
__global__ void tile_ranges_kernel(const uint32_t* __restrict__ counters, const uint64_t* __restrict__ keys,
                                   uint2* __restrict__ ranges) {
    const uint32_t total = counters[0];
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= total) return;
    const uint32_t t = (uint32_t)(keys[k] >> 32);
    if (k == 0 || (uint32_t)(keys[k - 1] >> 32) != t) ranges[t].x = k;
    if (k == total - 1 || (uint32_t)(keys[k + 1] >> 32) != t) ranges[t].y = k + 1;
}

}  // namespace pgr
