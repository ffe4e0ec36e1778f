// composite.hip.h -- front-to-back alpha compositing of one 16x16 tile per workgroup.
// SURVEY.md section 8a row a10.  Same operation order as oracle/pgr_oracle.c composite_tile();
// the only non-bit-exact step is exp (v_exp_f32 here, glibc expf in the oracle).
//
// Bound: VALU + transcendental issue (not HBM, not MFMA): per (pixel, list entry) ~20 VALU ops and
// one v_exp_f32.  The tile's list is staged through LDS in batches so each entry's 44 B is
// fetched from HBM/L2 once per tile and then broadcast-read by all four waves.
#pragma once
#include "pgr_common.h"

namespace pgr {

constexpr int COMP_THREADS = TILE * TILE;  // 256 = 4 waves; wave w owns pixel rows 4w..4w+3

struct CompOut {
    float* color;        // [3,H,W]
    float* depth;        // [H,W]
    float* final_T;      // optional
    uint32_t* n_contrib; // optional
};

__global__ __launch_bounds__(COMP_THREADS) void composite_kernel(const CameraDev* __restrict__ camp,
                                                                 const uint2* __restrict__ ranges,
                                                                 const uint32_t* __restrict__ gauss_sorted,
                                                                 const float2* __restrict__ xy,
                                                                 const float4* __restrict__ conic_opacity,
                                                                 const float4* __restrict__ rgbd, CompOut o) {
    const CameraDev& cam = *camp;
    const int W = cam.width, H = cam.height;
    const int tile = blockIdx.x;
    const int tile_x = tile % cam.grid_x, tile_y = tile / cam.grid_x;
    const int lx = threadIdx.x & (TILE - 1), ly = threadIdx.x / TILE;
    const int px = tile_x * TILE + lx, py = tile_y * TILE + ly;
    const bool inside = px < W && py < H;
    const float pxf = (float)px, pyf = (float)py;

    const uint2 range = ranges[tile];
    const int rounds = (int)((range.y - range.x + COMP_THREADS - 1) / COMP_THREADS);
    int todo = (int)(range.y - range.x);

    __shared__ float4 s_a[COMP_THREADS];  // x, y, hx, ny
    __shared__ float4 s_b[COMP_THREADS];  // hz, opacity, r, g
    __shared__ float2 s_c[COMP_THREADS];  // b, depth

    bool done = !inside;
    float T = 1.0f, Cr = 0.0f, Cg = 0.0f, Cb = 0.0f, D = 0.0f;
    uint32_t contributor = 0, last = 0;

    for (int r = 0; r < rounds; ++r, todo -= COMP_THREADS) {
        if (__syncthreads_and(done)) break;
        const int progress = r * COMP_THREADS + threadIdx.x;
        if (range.x + progress < range.y) {
            const uint32_t g = gauss_sorted[range.x + progress];
            const float2 p = xy[g];
            const float4 co = conic_opacity[g];
            const float4 cd = rgbd[g];
            s_a[threadIdx.x] = make_float4(p.x, p.y, -0.5f * co.x, -co.y);
            s_b[threadIdx.x] = make_float4(-0.5f * co.z, co.w, cd.x, cd.y);
            s_c[threadIdx.x] = make_float2(cd.z, cd.w);
        }
        __syncthreads();
        const int cnt = todo < COMP_THREADS ? todo : COMP_THREADS;
        for (int j = 0; !done && j < cnt; ++j) {
            ++contributor;
            const float4 a = s_a[j];
            const float dx = a.x - pxf, dy = a.y - pyf;
            const float4 b = s_b[j];
            const float power = fmaf(dx, fmaf(a.z, dx, a.w * dy), (b.x * dy) * dy);
            if (power > 0.0f) continue;
            const float e = __builtin_amdgcn_exp2f(power * 1.4426950408889634f);
            const float alpha = fminf(ALPHA_MAX, b.y * e);
            if (alpha < ALPHA_MIN) continue;
            const float test_T = fmaf(-alpha, T, T);
            if (test_T < T_EPS) {
                done = true;
                continue;
            }
            const float2 c = s_c[j];
            const float w = alpha * T;
            Cr = fmaf(b.z, w, Cr);
            Cg = fmaf(b.w, w, Cg);
            Cb = fmaf(c.x, w, Cb);
            D = fmaf(c.y, w, D);
            T = test_T;
            last = contributor;
        }
    }

    if (inside) {
        const size_t P = (size_t)W * H, pix = (size_t)py * W + px;
        o.color[0 * P + pix] = fmaf(T, cam.bg[0], Cr);
        o.color[1 * P + pix] = fmaf(T, cam.bg[1], Cg);
        o.color[2 * P + pix] = fmaf(T, cam.bg[2], Cb);
        o.depth[pix] = D;
        if (o.final_T) o.final_T[pix] = T;
        if (o.n_contrib) o.n_contrib[pix] = last;
    }
}

// ---- frame post-processing ---------------------------------------

// masks[k,p] = || img[:,p] - colors[k] ||_2 <= thr   (reference: src/gs/render.py:60-63,89-93)
__global__ void color_masks_kernel(const float* __restrict__ img, size_t P, const float* __restrict__ colors, int k,
                                   float thr, uint8_t* __restrict__ masks) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float r = img[p], g = img[P + p], b = img[2 * P + p];
    for (int c = 0; c < k; ++c) {
        const float d0 = r - colors[3 * c], d1 = g - colors[3 * c + 1], d2 = b - colors[3 * c + 2];
        const float dist = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
        masks[(size_t)c * P + p] = dist <= thr ? 1 : 0;
    }
}

// rgb uint8 HWC (wraps like numpy's astype on x86), depth uint16 millimetres (pegasus.py:347,355)
__global__ void quantize_kernel(const float* __restrict__ img, const float* __restrict__ depth, size_t P,
                                uint8_t* __restrict__ rgb_hwc, uint16_t* __restrict__ depth_mm) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    if (img && rgb_hwc) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = img[(size_t)c * P + p] * 255.0f;
            v = fminf(fmaxf(v, -2147483520.0f), 2147483520.0f);
            rgb_hwc[3 * p + c] = (uint8_t)((int)v & 0xFF);
        }
    }
    if (depth && depth_mm) {
        float v = depth[p] * 1000.0f;
        v = fminf(fmaxf(v, -2147483520.0f), 2147483520.0f);
        depth_mm[p] = (uint16_t)((int)v & 0xFFFF);
    }
}

}  // namespace pgr
