// pgr_common.h -- shared device/host definitions for libpegasus_raster (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pegasus_raster.h"

namespace pgr {

constexpr int TILE = PGR_TILE_SIZE;          // 16x16 pixel tiles (part of the contract: radii/tiles_touched)
constexpr int WAVE = 64;                     // gfx950 wavefront
constexpr float NEAR_Z = 0.2f;
constexpr float LOWPASS = 0.3f;
constexpr float ALPHA_MAX = 0.99f;
constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float T_EPS = 0.0001f;

// Per-view constants, resident in HBM so that no host round trip is needed to read the
// caller's device-side camera tensors.  256 B, read through the scalar cache.
struct alignas(16) CameraDev {
    float view[16];
    float proj[16];
    float campos[3];
    float tanfovx;
    float bg[3];
    float tanfovy;
    float focal_x, focal_y;
    int32_t width, height;
    int32_t grid_x, grid_y;
    int32_t pad[2];
    float pad2[16];
};
static_assert(sizeof(CameraDev) == 256, "CameraDev must be 256 B");

// Workspace carve-up (all offsets multiples of 256 B).
struct Layout {
    size_t cam, counters, splats, radii, rects, crects, rel, ranges, bucket, alt,
        gauss_sorted, total;
    int32_t tiles, grid_x, grid_y;
    int32_t n_blocks;
    int32_t n_chunks;
};

// Work ordering: list-length classes (256, log-spaced); see composite.hip.h "work order".
constexpr int ORDER_CLASSES = 256;

__device__ __forceinline__ int length_class(uint32_t len) {
    if (len == 0) return 0;
    const int msb = 31 - __clz((int)len);
    const uint32_t frac = msb >= 3 ? (len >> (msb - 3)) & 7u : (len << (3 - msb)) & 7u;
    return msb * 8 + (int)frac + 1;   // 1..256 -> clamp below
}

constexpr int PRE_BLOCK = 256;               // Gaussians per preprocess workgroup

}  // namespace pgr
