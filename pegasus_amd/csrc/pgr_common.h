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

// Global-memory accessors.  Pointers that reach a kernel through a table in memory (per-view slices, output images)
// are generic, and the compiler addresses them with FLAT instructions: those also count on lgkmcnt, so every LDS wait
// behind one stalls until the flat access has cleared the texture-address unit (a scattered 8-byte store: 64 lines).
// These helpers say "global" (address space 1): global_load / global_store count on vmcnt only.
#define PGR_GLOBAL __attribute__((address_space(1)))
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t gload(const uint32_t* p) { return *(const PGR_GLOBAL uint32_t*)p; }
__device__ __forceinline__ uint8_t gload(const uint8_t* p) { return *(const PGR_GLOBAL uint8_t*)p; }
__device__ __forceinline__ int32_t gload(const int32_t* p) { return *(const PGR_GLOBAL int32_t*)p; }
__device__ __forceinline__ float gload(const float* p) { return *(const PGR_GLOBAL float*)p; }
__device__ __forceinline__ uint64_t gload(const uint64_t* p) { return *(const PGR_GLOBAL uint64_t*)p; }
__device__ __forceinline__ uint2 gload(const uint2* p) { const u32x2_t v = *(const PGR_GLOBAL u32x2_t*)p; return make_uint2(v.x, v.y); }
__device__ __forceinline__ uint4 gload(const uint4* p) { const u32x4_t v = *(const PGR_GLOBAL u32x4_t*)p; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ float2 gload(const float2* p) { const f32x2_t v = *(const PGR_GLOBAL f32x2_t*)p; return make_float2(v.x, v.y); }
__device__ __forceinline__ float4 gload(const float4* p) { const f32x4_t v = *(const PGR_GLOBAL f32x4_t*)p; return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ f32x4_t gload_quad(const float4* p) { return *(const PGR_GLOBAL f32x4_t*)p; }   // as one register tuple
__device__ __forceinline__ void gstore(uint32_t* p, uint32_t v) { *(PGR_GLOBAL uint32_t*)p = v; }
__device__ __forceinline__ void gstore(float* p, float v) { *(PGR_GLOBAL float*)p = v; }
__device__ __forceinline__ void gstore(uint8_t* p, uint8_t v) { *(PGR_GLOBAL uint8_t*)p = v; }
__device__ __forceinline__ void gstore(uint16_t* p, uint16_t v) { *(PGR_GLOBAL uint16_t*)p = v; }
__device__ __forceinline__ void gstore(uint64_t* p, uint64_t v) { *(PGR_GLOBAL uint64_t*)p = v; }
__device__ __forceinline__ void gstore(uint2* p, uint2 v) { *(PGR_GLOBAL u32x2_t*)p = u32x2_t{v.x, v.y}; }
__device__ __forceinline__ void gstore(float4* p, float4 v) { *(PGR_GLOBAL f32x4_t*)p = f32x4_t{v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ uint32_t gatomic_add(uint32_t* p, uint32_t v) {
    return __hip_atomic_fetch_add((PGR_GLOBAL uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t gatomic_max(uint32_t* p, uint32_t v) {
    return __hip_atomic_fetch_max((PGR_GLOBAL uint32_t*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t gatomic_load(const uint32_t* p) {     // sees other workgroups' agent-scope atomics
    return __hip_atomic_load((const PGR_GLOBAL uint32_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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
    int32_t depth_mode;                      // PgrDepthMode
    int32_t pad[1];
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
    int64_t max_instances;
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
