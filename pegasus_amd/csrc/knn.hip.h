// knn.hip.h -- mean squared distance to the 3 nearest neighbours of every point (SURVEY.md section 8f row 4:
// simple_knn._C.distCUDA2, used once per scene by GaussianModel.create_from_pcd to size the initial splats,
// /root/reference/src/gs/gaussian_model.py:25,147).  Not on the render path.
//
// Exact 3-NN through a uniform grid built on the device (no host round trip): bounding box (ordered-int atomics) ->
// cell histogram -> scan -> counting-sort of the points by cell -> per point, shells of cells of growing Chebyshev
// radius until the third-best distance is inside the shell already covered.
#pragma once
#include "pgr_common.h"

namespace pgr {

constexpr int KNN_MAX_GRID = 128;                  // cells per axis (2 M cells at most)

struct KnnGrid {                                   // device-resident header (64 B)
    uint32_t lo[3], hi[3];                         // bounding box as order-preserving uint encodings of the floats
    float cell, inv_cell;
    int32_t dim[3];
    int32_t pad[5];
};

__device__ __forceinline__ uint32_t float_order(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float order_float(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ void knn_bbox_kernel(int n, const float* __restrict__ xyz, KnnGrid* __restrict__ g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const uint32_t u = float_order(xyz[3 * i + a]);
        atomicMin(&g->lo[a], u);
        atomicMax(&g->hi[a], u);
    }
}

// one thread: cubic cells sized so that the longest axis has `target` cells
__global__ void knn_grid_kernel(KnnGrid* __restrict__ g, int target) {
    float ext = 0.f, e[3];
    for (int a = 0; a < 3; ++a) { e[a] = order_float(g->hi[a]) - order_float(g->lo[a]); ext = fmaxf(ext, e[a]); }
    const float cell = ext > 0.f ? ext / (float)target : 1.0f;
    g->cell = cell;
    g->inv_cell = 1.0f / cell;
    for (int a = 0; a < 3; ++a) g->dim[a] = min(target, max(1, (int)(e[a] / cell) + 1));
}

__device__ __forceinline__ int3 knn_cell_of(const KnnGrid& g, float x, float y, float z) {
    int3 c;
    c.x = min(g.dim[0] - 1, max(0, (int)((x - order_float(g.lo[0])) * g.inv_cell)));
    c.y = min(g.dim[1] - 1, max(0, (int)((y - order_float(g.lo[1])) * g.inv_cell)));
    c.z = min(g.dim[2] - 1, max(0, (int)((z - order_float(g.lo[2])) * g.inv_cell)));
    return c;
}
__device__ __forceinline__ int knn_cell_index(const KnnGrid& g, int3 c) { return (c.z * g.dim[1] + c.y) * g.dim[0] + c.x; }

__global__ void knn_count_kernel(int n, const float* __restrict__ xyz, const KnnGrid* __restrict__ g,
                                 uint32_t* __restrict__ cell_count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    atomicAdd(&cell_count[knn_cell_index(*g, knn_cell_of(*g, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]))], 1u);
}

// one workgroup: exclusive scan of the cell counts (start[cells] = n)
__global__ __launch_bounds__(1024) void knn_scan_kernel(const KnnGrid* __restrict__ g, const uint32_t* __restrict__ cell_count,
                                                        uint32_t* __restrict__ cell_start) {
    __shared__ uint32_t wave_tot[1024 / WAVE];
    __shared__ uint32_t carry_s;
    const int cells = g->dim[0] * g->dim[1] * g->dim[2];
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < cells; base += 1024) {
        const int idx = base + threadIdx.x;
        const uint32_t v = idx < cells ? cell_count[idx] : 0u;
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
        if (idx < cells) cell_start[idx] = carry + wave_prefix + s - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wave_prefix + s;
        __syncthreads();
    }
    if (threadIdx.x == 0) cell_start[cells] = carry_s;
}

// counting sort: sorted[slot] = (x, y, z, original index); cursors start as copies of cell_start
__global__ void knn_scatter_kernel(int n, const float* __restrict__ xyz, const KnnGrid* __restrict__ g,
                                   uint32_t* __restrict__ cursor, float4* __restrict__ sorted) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    const uint32_t slot = atomicAdd(&cursor[knn_cell_index(*g, knn_cell_of(*g, x, y, z))], 1u);
    sorted[slot] = make_float4(x, y, z, __int_as_float(i));
}

__device__ __forceinline__ void knn_insert(float (&best)[3], float d2) {
    if (d2 < best[2]) {
        if (d2 < best[1]) {
            best[2] = best[1];
            if (d2 < best[0]) { best[1] = best[0]; best[0] = d2; } else best[1] = d2;
        } else {
            best[2] = d2;
        }
    }
}

// one thread per point, in cell order (neighbouring threads search the same cells)
__global__ void knn_search_kernel(int n, const KnnGrid* __restrict__ gp, const uint32_t* __restrict__ cell_start,
                                  const float4* __restrict__ sorted, float* __restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const KnnGrid g = *gp;
    const float4 p = sorted[s];
    const int3 c = knn_cell_of(g, p.x, p.y, p.z);
    float best[3] = {3.402823466e38f, 3.402823466e38f, 3.402823466e38f};
    const int rmax = max(g.dim[0], max(g.dim[1], g.dim[2]));
    for (int r = 0; r <= rmax; ++r) {
        const int z0 = max(0, c.z - r), z1 = min(g.dim[2] - 1, c.z + r);
        const int y0 = max(0, c.y - r), y1 = min(g.dim[1] - 1, c.y + r);
        const int x0 = max(0, c.x - r), x1 = min(g.dim[0] - 1, c.x + r);
        for (int z = z0; z <= z1; ++z)
            for (int y = y0; y <= y1; ++y) {
                const bool face = abs(z - c.z) == r || abs(y - c.y) == r;
                for (int x = x0; x <= x1; ++x) {
                    if (!face && abs(x - c.x) != r) continue;            // interior of the cube: visited at a smaller r
                    const int ci = (z * g.dim[1] + y) * g.dim[0] + x;
                    const uint32_t a = cell_start[ci], b = cell_start[ci + 1];
                    for (uint32_t j = a; j < b; ++j) {
                        if ((int)j == s) continue;
                        const float4 q = sorted[j];
                        const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
                        knn_insert(best, dx * dx + dy * dy + dz * dz);
                    }
                }
            }
        // everything not visited yet is farther than r cells away along some axis
        const float reach = (float)r * g.cell * 0.99999f;
        if (best[2] <= reach * reach) break;
    }
    out[__float_as_int(p.w)] = (best[0] + best[1] + best[2]) / 3.0f;
}

}  // namespace pgr
