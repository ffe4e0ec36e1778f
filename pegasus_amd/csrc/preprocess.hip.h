// preprocess.hip.h -- per-Gaussian projection, EWA covariance, SH colour, tile rectangle.
//
// Replaces the per-Gaussian stage of the reference's missing CUDA extension (SURVEY.md
// section 8a row a5; call sites /root/reference/src/gs/render.py:16, /root/reference/pegasus.py:271).
// Arithmetic contract: identical operation order to oracle/pgr_oracle.c (fp32, explicit fmaf,
// translation unit compiled with -ffp-contract=off) => every output of this kernel is bit-exact
// against the oracle.
//
// HBM-bound: reads 12 B (xyz) for every Gaussian, +28 B (scale, rotation) for the ones in front of
// the near plane, +196 B (opacity, 48 SH floats) for the ones whose tile rectangle is non-empty;
// writes 4+8 B (radius, packed rectangle) for every Gaussian and 44 B for survivors.
#pragma once
#include "blockcull.hip.h"
#include "pgr_common.h"

namespace pgr {

__device__ __forceinline__ float dot3_chain(float a0, float b0, float a1, float b1, float a2, float b2) {
    return fmaf(a2, b2, fmaf(a1, b1, a0 * b0));
}

__device__ __forceinline__ int clamp_trunc(float f, int hi) {
    if (!(f > 0.0f)) return 0;
    if (f >= (float)hi) return hi;
    return (int)f;
}

struct TileRect { int minx, miny, maxx, maxy; };

__device__ __forceinline__ TileRect tile_rect(float px, float py, int radius, int grid_x, int grid_y) {
    const float rf = (float)radius;
    TileRect r;
    r.minx = clamp_trunc((px - rf) / (float)TILE, grid_x);
    r.miny = clamp_trunc((py - rf) / (float)TILE, grid_y);
    r.maxx = clamp_trunc((px + rf + (float)(TILE - 1)) / (float)TILE, grid_x);
    r.maxy = clamp_trunc((py + rf + (float)(TILE - 1)) / (float)TILE, grid_y);
    return r;
}

__device__ __forceinline__ void cov3d_from_scale_rot(float s0, float s1, float s2, float mod, float4 q, float cov[6]) {
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    float R[3][3];
    R[0][0] = 1.0f - 2.0f * (y * y + z * z);
    R[0][1] = 2.0f * (x * y - r * z);
    R[0][2] = 2.0f * (x * z + r * y);
    R[1][0] = 2.0f * (x * y + r * z);
    R[1][1] = 1.0f - 2.0f * (x * x + z * z);
    R[1][2] = 2.0f * (y * z - r * x);
    R[2][0] = 2.0f * (x * z - r * y);
    R[2][1] = 2.0f * (y * z + r * x);
    R[2][2] = 1.0f - 2.0f * (x * x + y * y);
    const float sx = mod * s0, sy = mod * s1, sz = mod * s2;
    float M[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        M[i][0] = R[i][0] * sx;
        M[i][1] = R[i][1] * sy;
        M[i][2] = R[i][2] * sz;
    }
    cov[0] = dot3_chain(M[0][0], M[0][0], M[0][1], M[0][1], M[0][2], M[0][2]);
    cov[1] = dot3_chain(M[0][0], M[1][0], M[0][1], M[1][1], M[0][2], M[1][2]);
    cov[2] = dot3_chain(M[0][0], M[2][0], M[0][1], M[2][1], M[0][2], M[2][2]);
    cov[3] = dot3_chain(M[1][0], M[1][0], M[1][1], M[1][1], M[1][2], M[1][2]);
    cov[4] = dot3_chain(M[1][0], M[2][0], M[1][1], M[2][1], M[1][2], M[2][2]);
    cov[5] = dot3_chain(M[2][0], M[2][0], M[2][1], M[2][1], M[2][2], M[2][2]);
}

template <int DEG>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float b[16]) {
    constexpr float C0 = 0.28209479177387814f;
    constexpr float C1 = 0.4886025119029199f;
    constexpr float C2_0 = 1.0925484305920792f, C2_1 = -1.0925484305920792f, C2_2 = 0.31539156525252005f,
                    C2_3 = -1.0925484305920792f, C2_4 = 0.5462742152960396f;
    constexpr float C3_0 = -0.5900435899266435f, C3_1 = 2.890611442640554f, C3_2 = -0.4570457994644658f,
                    C3_3 = 0.3731763325901154f, C3_4 = -0.4570457994644658f, C3_5 = 1.445305721320277f,
                    C3_6 = -0.5900435899266435f;
    b[0] = C0;
    if constexpr (DEG > 0) {
        b[1] = -(C1 * y);
        b[2] = C1 * z;
        b[3] = -(C1 * x);
    }
    if constexpr (DEG > 1) {
        const float xx = x * x, yy = y * y, zz = z * z;
        const float xy = x * y, yz = y * z, xz = x * z;
        b[4] = C2_0 * xy;
        b[5] = C2_1 * yz;
        b[6] = C2_2 * (2.0f * zz - xx - yy);
        b[7] = C2_3 * xz;
        b[8] = C2_4 * (xx - yy);
        if constexpr (DEG > 2) {
            b[9] = C3_0 * y * (3.0f * xx - yy);
            b[10] = C3_1 * xy * z;
            b[11] = C3_2 * y * (4.0f * zz - xx - yy);
            b[12] = C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
            b[13] = C3_4 * x * (4.0f * zz - xx - yy);
            b[14] = C3_5 * z * (xx - yy);
            b[15] = C3_6 * x * (xx - 3.0f * yy);
        }
    }
}

// SH -> RGB for one Gaussian, coefficients read straight from HBM (12 B per coefficient).
template <int DEG>
__device__ __forceinline__ float3 sh_to_rgb(const float* __restrict__ sh, float dx, float dy, float dz) {
    float b[16];
    sh_basis<DEG>(dx, dy, dz, b);
    constexpr int NC = (DEG + 1) * (DEG + 1);
    float acc[3];
    // coefficient-major, RGB-minor: 3*NC contiguous floats; 16-B aligned when the stride is 16
    acc[0] = b[0] * sh[0];
    acc[1] = b[0] * sh[1];
    acc[2] = b[0] * sh[2];
#pragma unroll
    for (int k = 1; k < NC; ++k) {
        acc[0] = fmaf(b[k], sh[3 * k + 0], acc[0]);
        acc[1] = fmaf(b[k], sh[3 * k + 1], acc[1]);
        acc[2] = fmaf(b[k], sh[3 * k + 2], acc[2]);
    }
    return make_float3(fmaxf(acc[0] + 0.5f, 0.0f), fmaxf(acc[1] + 0.5f, 0.0f), fmaxf(acc[2] + 0.5f, 0.0f));
}

// Candidate rectangle for the binning walk: the 3-sigma rectangle clipped to the axis-aligned box of the
// ellipse {q <= tau''} where tau'' bounds from above every threshold the tight-list predicate
// (cull.hip.h) can apply to a tile of this splat; tiles outside it are rejected by the predicate anyway, so the
// lists are unchanged -- there are just fewer candidates to test (anisotropic and faint splats shrink most).
__device__ __forceinline__ uint2 candidate_rect(const TileRect& r, float mx, float my, float c_xx, float c_yy,
                                                float con_x, float con_y, float con_z, float opacity, int radius) {
    if (opacity < ALPHA_MIN) return make_uint2(0u, 0u);                 // never listed
    const uint2 full = make_uint2((uint32_t)r.minx | ((uint32_t)r.miny << 16), (uint32_t)r.maxx | ((uint32_t)r.maxy << 16));
    if (!(con_x > 0.0f) || !(con_z > 0.0f)) return full;                // degenerate conic: the predicate keeps all
    const float t = 255.0f * opacity;
    const uint32_t bits = __float_as_uint(t);
    const float e = (float)((int)((bits >> 23) & 0xffu) - 127);
    const float m = __uint_as_float((bits & 0x007fffffu) | 0x3f800000u);
    const float tau = 1.3862944f * (e + (m - 1.0f) + 0.0861f);
    const float D = (float)radius + 2.0f * (float)TILE;                  // no pixel of the rectangle is farther than this
    const float M = (con_x + 2.0f * fabsf(con_y) + con_z) * D * D;
    const float tau2 = 1.001f * (tau + 0.00001f * M + 0.01f) + 0.001f;
    const float ex = sqrtf(tau2 * c_xx) * 1.001f + 1.0f, ey = sqrtf(tau2 * c_yy) * 1.001f + 1.0f;
    if (!(ex < 1.0e9f) || !(ey < 1.0e9f)) return full;
    // tiles whose pixel-centre span [16t, 16t+15] meets [m - e, m + e]
    const int minx = max(r.minx, (int)floorf((mx - ex - (float)(TILE - 1)) / (float)TILE) + 0);
    const int miny = max(r.miny, (int)floorf((my - ey - (float)(TILE - 1)) / (float)TILE) + 0);
    const int maxx = min(r.maxx, (int)floorf((mx + ex) / (float)TILE) + 1);
    const int maxy = min(r.maxy, (int)floorf((my + ey) / (float)TILE) + 1);
    if (maxx <= minx || maxy <= miny) return make_uint2(0u, 0u);
    return make_uint2((uint32_t)minx | ((uint32_t)miny << 16), (uint32_t)maxx | ((uint32_t)maxy << 16));
}
// (Round 6 built a per-tile-ROW refinement of this rectangle -- parallelogram slices of the ellipse, 21 % fewer candidates,
// lists unchanged -- and measured it a net loss: computing the intervals costs the preprocess what the walks save.
// profiles/r06_rows_ab.txt; code: git log -S row_code -- pegasus_amd/csrc/preprocess.hip.h)

// Per-Gaussian, per-view record the later stages gather: ONE 48-byte row instead of four arrays, so a gather
// touches 1-2 64-byte sectors instead of 3-4 (PMC: the compositor fetched 2.3x its algorithmic bytes with the
// split layout).  q0 = (x, y, A, B), q1 = (C, opacity, B/C, B/A), q2 = (r, g, b, depth): the
// binning walks need q0 and q1 only, the compositor parks q2 as it is.
constexpr int SPLAT_F4 = 3;

struct PreOut {
    float4* splats;          // [n, SPLAT_F4]
    uint2* rects;            // packed tile rectangle (4 x uint16: minx,miny,maxx,maxy), all zero when culled; NULL = not wanted
    uint2* crects;           // candidate rectangle for binning: rects clipped to the alpha >= 1/255 ellipse's box
    int32_t* radii;          // NULL = not wanted
};

// SH coefficients of one Gaussian held in registers across the views of a batch.
struct ShRegs { float v[48]; };

template <int DEG>
__device__ __forceinline__ float3 sh_regs_to_rgb(const ShRegs& sh, float dx, float dy, float dz) {
    float b[16];
    sh_basis<DEG>(dx, dy, dz, b);
    constexpr int NC = (DEG + 1) * (DEG + 1);
    float acc[3];
    acc[0] = b[0] * sh.v[0];
    acc[1] = b[0] * sh.v[1];
    acc[2] = b[0] * sh.v[2];
#pragma unroll
    for (int k = 1; k < NC; ++k) {
        acc[0] = fmaf(b[k], sh.v[3 * k + 0], acc[0]);
        acc[1] = fmaf(b[k], sh.v[3 * k + 1], acc[1]);
        acc[2] = fmaf(b[k], sh.v[3 * k + 2], acc[2]);
    }
    return make_float3(fmaxf(acc[0] + 0.5f, 0.0f), fmaxf(acc[1] + 0.5f, 0.0f), fmaxf(acc[2] + 0.5f, 0.0f));
}

template <int DEG>
__device__ __forceinline__ void load_sh(ShRegs& sh, const float* __restrict__ p, bool vec4) {
    constexpr int NF = 3 * (DEG + 1) * (DEG + 1);
    if (vec4 && NF % 4 == 0) {                   // 16-B aligned rows (stride 16 coefficients): 12 x dwordx4 at degree 3
#pragma unroll
        for (int k = 0; k < NF / 4; ++k) {
            const float4 q = reinterpret_cast<const float4*>(p)[k];
            sh.v[4 * k + 0] = q.x; sh.v[4 * k + 1] = q.y; sh.v[4 * k + 2] = q.z; sh.v[4 * k + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < NF; ++k) sh.v[k] = p[k];
    }
}

// The split layout (PgrScene::shs_rest): coefficient 0 from the [n,1,3] array, the others from the [n,stride-1,3] one.
// Rows of 3 and 3 (stride - 1) floats are only 4-B aligned -- which is all a global load asks for: the first coefficient is
// one 12-B load, the others arrive as 16-B quads (eleven + one word at degree 3: as many loads as the concatenated layout's
// twelve).  What the split layout costs is not the load count (one 12-B load per coefficient measured the same on one box)
// but the rows themselves: 180 + 12 B straddle the cache lines that rows of 192 B sit in -- a single view's preprocess takes
// 92 us against 83 (gaussian_renderer.py keeps the concatenation for a model that is rendered again and again).
typedef float f32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
template <int DEG>
__device__ __forceinline__ void load_sh_split(ShRegs& sh, const float* __restrict__ dc, const float* __restrict__ rest) {
    constexpr int NR = 3 * ((DEG + 1) * (DEG + 1) - 1);      // floats of the higher coefficients
    const f32x3_a4 c0 = *reinterpret_cast<const f32x3_a4*>(dc);
    sh.v[0] = c0.x; sh.v[1] = c0.y; sh.v[2] = c0.z;
#pragma unroll
    for (int k = 0; k < NR / 4; ++k) {
        const f32x4_a4 q = *reinterpret_cast<const f32x4_a4*>(rest + 4 * k);
        sh.v[3 + 4 * k + 0] = q.x; sh.v[3 + 4 * k + 1] = q.y; sh.v[3 + 4 * k + 2] = q.z; sh.v[3 + 4 * k + 3] = q.w;
    }
#pragma unroll
    for (int k = NR / 4 * 4; k < NR; ++k) sh.v[3 + k] = rest[k];
}

// One thread per Gaussian, ALL views of the batch: the 236 B of scene data are read from HBM once per
// batch instead of once per view (the view-independent 3D covariance is also built once), which turns the
// kernel from read-bound (~356 MB/view) into write-bound (12 N + 44 V per view).  Per-view arithmetic is
// unchanged, so the outputs are bit-identical to the single-view form and to the oracle.
//
// POSED (dynamic scenes, include/pegasus_raster.h PgrPosedObjects): every view carries its own rigid pose per object
// and a Gaussian of object k is placed on the fly -- position and orientation with the arithmetic of
// compose_object_kernel, colour from its own SH coefficients evaluated in the object's frame (direction R^T d) --
// so a batch of TIME STEPS runs like a batch of cameras: no composed copy of the scene is written (236 B read +
// 236 B written per object Gaussian and step) and no SH band rotation is evaluated.
constexpr int POSE_STRIDE = 20;      // floats per pose: R[9] row-major, t[3], center[3], q[4] (w,x,y,z), pad
struct PosedDev {
    const int32_t* object_id;        // [n] 0 = not posed
    const float* poses;              // [n_views, k, POSE_STRIDE]
    int32_t k;
};

// LAYERED (pgr_forward_layers_async: silhouettes): Gaussian i belongs to image layer_id[i] of the view (0: to none, it is
// dropped).  Everything per Gaussian is computed as if its layer were rendered alone; only the tile ROWS of its rectangles
// are moved down by (layer - 1) x grid_y, so that the binning sees one tall image of n_layers x grid_y tile rows and
// builds per-(tile, layer) lists.
struct LayerDev {
    const int32_t* layer_id;         // [n]
    int32_t n_layers;
};

// vis_out != NULL: hierarchical culling (blockcull.hip.h).  The wave first bounds its 64 Gaussians, then lane l decides
// for view 64 c + l whether the whole block is certainly culled there (one ballot = 64 views): such views are skipped by
// the whole wave (radii / rectangles, where the caller wants them, are zero; the candidate rectangle is not written:
// the binning walk consults the same bits, stored as vis_out[group * vis_words + view / 32]).
// SPLIT_SH: the coefficients come from two arrays (PgrScene::shs_rest) -- its own instantiation, so that the combined layout's
// kernels are the ones they were (a run-time branch in front of the SH loads cost the batch preprocess 11 %).
template <int DEG, bool POSED, bool LAYERED = false, bool SPLIT_SH = false>
__global__ __launch_bounds__(PRE_BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void preprocess_batch_kernel(PgrScene sc, const CameraDev* __restrict__ cams,
                                                                     const PreOut* __restrict__ outs, int n_views,
                                                                     PosedDev posed, uint32_t* __restrict__ vis_out,
                                                                     int vis_words, LayerDev layers = LayerDev{nullptr, 0},
                                                                     int views_per_block = 0x7fffffff) {
    const int i = blockIdx.x * PRE_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & (WAVE - 1);
    if (i - lane >= sc.n) return;            // the whole wave is past the end
    const bool real = i < sc.n;              // lanes past the end stay for the wave-wide steps and do nothing else
    const int group = __builtin_amdgcn_readfirstlane(i / WAVE);
    BlockBounds bounds;
    if (vis_out) bounds = wave_block_bounds(sc, i, real, POSED ? posed.object_id : nullptr);
    unsigned long long vis_mask = ~0ull;
    const float bx = real ? sc.means3d[3 * i + 0] : 0.f, by = real ? sc.means3d[3 * i + 1] : 0.f,
                bz = real ? sc.means3d[3 * i + 2] : 0.f;
    const int oid = POSED && real ? posed.object_id[i] : 0;
    int layer = 0;                           // LAYERED: 1-based image layer of this Gaussian, 0 = in none
    if (LAYERED && real) {
        layer = layers.layer_id[i];
        if (layer < 0 || layer > layers.n_layers) layer = 0;
    }
    float cov[6];
    bool have_cov = false, have_sh = false;
    ShRegs sh;
    const bool vec4 = (sc.sh_stride * 3) % 4 == 0 && (reinterpret_cast<uintptr_t>(sc.shs) & 15u) == 0;

    // gridDim.y > 1 (small scenes: fewer waves than the chip holds): the views are dealt to blockIdx.y in groups of
    // views_per_block -- a wave walks its views one after the other, and 3 000 waves doing 32 views each leave the SIMDs idle
    // behind their dependency chains (vis_out is NULL then: its words hold 32 views)
    const int v_begin = (int)blockIdx.y * views_per_block;
    const int v_end = (int)min((long long)n_views, (long long)v_begin + views_per_block);
    for (int v = v_begin; v < v_end; ++v) {
        const CameraDev& cam = cams[v];
        const PreOut& o = outs[v];
        if (vis_out) {
            if ((v & (WAVE - 1)) == 0) {
                const int view = v + lane;
                vis_mask = __ballot(view < n_views && !block_is_culled(bounds, cams[view]));
                if (lane == 0) {
                    vis_out[(size_t)group * vis_words + (v >> 5)] = (uint32_t)vis_mask;
                    if ((v >> 5) + 1 < vis_words) vis_out[(size_t)group * vis_words + (v >> 5) + 1] = (uint32_t)(vis_mask >> 32);
                }
            }
            if (!((vis_mask >> (v & (WAVE - 1))) & 1ull)) {
                if (real && o.radii) o.radii[i] = 0;
                if (real && o.rects) o.rects[i] = make_uint2(0u, 0u);
                continue;
            }
        }
        if (!real) continue;                 // (rejoins the wave at the next view's ballot)
        if (LAYERED && layer == 0) {         // in no layer: as if culled
            if (o.radii) o.radii[i] = 0;
            if (o.rects) o.rects[i] = make_uint2(0u, 0u);
            o.crects[i] = make_uint2(0u, 0u);
            continue;
        }
        int radius = 0;
        uint2 rect = make_uint2(0u, 0u), crect = make_uint2(0u, 0u);
        // The view matrix and the camera scalars are fetched ONCE per iteration, here: left to the compiler, every field
        // is loaded where it is first used -- five groups of s_load, each followed by a full s_waitcnt, i.e. five exposed
        // scalar-cache round trips per view and wave in a kernel that runs 4 waves per SIMD (VALU busy 74 %).  The empty
        // asm statements pin the loads (the optimiser would sink them back into the branches).  The projection matrix
        // stays where it is used (one more group, survivors of the near cull only): pinning it too needs more SGPRs than
        // a wave has, and the spills cost a wave of occupancy.
        float vm[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) vm[k] = cam.view[k];
        const float* pm = cam.proj;
        float cam_cx = cam.campos[0], cam_cy = cam.campos[1], cam_cz = cam.campos[2];
        float cam_tanx = cam.tanfovx, cam_tany = cam.tanfovy, cam_fx = cam.focal_x, cam_fy = cam.focal_y;
        int cam_w = cam.width, cam_h = cam.height, cam_gx = cam.grid_x, cam_gy = cam.grid_y;
        float4* out_splats = o.splats;           // (the view's output pointers ride along in the same round trip)
        uint2* out_crects = o.crects;
        asm volatile("" : "+s"(out_splats), "+s"(out_crects));
#pragma unroll
        for (int k = 0; k < 15; ++k)
            if ((k & 3) != 3) asm volatile("" : "+s"(vm[k]));
        asm volatile("" : "+s"(cam_cx), "+s"(cam_cy), "+s"(cam_cz), "+s"(cam_tanx), "+s"(cam_tany), "+s"(cam_fx), "+s"(cam_fy));
        asm volatile("" : "+s"(cam_w), "+s"(cam_h), "+s"(cam_gx), "+s"(cam_gy));
        float px = bx, py = by, pz = bz;
        const float* P = nullptr;
        if (POSED && oid > 0) {
            P = posed.poses + ((size_t)v * posed.k + (size_t)(oid - 1)) * POSE_STRIDE;
            const float dx = bx - P[12], dy = by - P[13], dz = bz - P[14];
            px = fmaf(P[2], dz, fmaf(P[1], dy, P[0] * dx)) + P[12] + P[9];
            py = fmaf(P[5], dz, fmaf(P[4], dy, P[3] * dx)) + P[13] + P[10];
            pz = fmaf(P[8], dz, fmaf(P[7], dy, P[6] * dx)) + P[14] + P[11];
        }
        float tx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
        float ty = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
        const float tz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
        if (tz > NEAR_Z) {
            const float hx = pm[0] * px + pm[4] * py + pm[8] * pz + pm[12];
            const float hy = pm[1] * px + pm[5] * py + pm[9] * pz + pm[13];
            const float hw = pm[3] * px + pm[7] * py + pm[11] * pz + pm[15];
            const float p_w = 1.0f / (hw + 0.0000001f);
            const float ndc_x = hx * p_w, ndc_y = hy * p_w;
            if (POSED && P) {            // orientation follows the pose: q' = q_R (x) normalise(q), per view
                const float4 q = reinterpret_cast<const float4*>(sc.rotations)[i];
                const float inv = 1.0f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
                const float w = q.x * inv, x = q.y * inv, y = q.z * inv, z = q.w * inv;
                const float a = P[15], b = P[16], c = P[17], d = P[18];
                const float4 qp = make_float4(a * w - b * x - c * y - d * z, a * x + b * w + c * z - d * y,
                                              a * y - b * z + c * w + d * x, a * z + b * y - c * x + d * w);
                cov3d_from_scale_rot(sc.scales[3 * i + 0], sc.scales[3 * i + 1], sc.scales[3 * i + 2],
                                     sc.scale_modifier, qp, cov);
            } else if (!have_cov) {
                if (sc.cov3d_precomp) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) cov[k] = sc.cov3d_precomp[6 * (size_t)i + k];
                } else {
                    const float4 q = reinterpret_cast<const float4*>(sc.rotations)[i];
                    cov3d_from_scale_rot(sc.scales[3 * i + 0], sc.scales[3 * i + 1], sc.scales[3 * i + 2],
                                         sc.scale_modifier, q, cov);
                }
                have_cov = true;
            }
            const float limx = 1.3f * cam_tanx, limy = 1.3f * cam_tany;
            const float txtz = tx / tz, tytz = ty / tz;
            tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
            ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
            const float j00 = cam_fx / tz;
            const float j02 = -(cam_fx * tx) / (tz * tz);
            const float j11 = cam_fy / tz;
            const float j12 = -(cam_fy * ty) / (tz * tz);
            float T0[3], T1[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                T0[c] = fmaf(j02, vm[4 * c + 2], j00 * vm[4 * c + 0]);
                T1[c] = fmaf(j12, vm[4 * c + 2], j11 * vm[4 * c + 1]);
            }
            const float S[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
            float U0[3], U1[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                U0[c] = dot3_chain(T0[0], S[0][c], T0[1], S[1][c], T0[2], S[2][c]);
                U1[c] = dot3_chain(T1[0], S[0][c], T1[1], S[1][c], T1[2], S[2][c]);
            }
            const float c_xx = dot3_chain(U0[0], T0[0], U0[1], T0[1], U0[2], T0[2]) + LOWPASS;
            const float c_xy = dot3_chain(U0[0], T1[0], U0[1], T1[1], U0[2], T1[2]);
            const float c_yy = dot3_chain(U1[0], T1[0], U1[1], T1[1], U1[2], T1[2]) + LOWPASS;
            const float det = c_xx * c_yy - c_xy * c_xy;
            if (det != 0.0f) {
                const float det_inv = 1.0f / det;
                const float con_x = c_yy * det_inv, con_y = -c_xy * det_inv, con_z = c_xx * det_inv;
                const float mid = 0.5f * (c_xx + c_yy);
                const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
                const float lambda1 = mid + disc, lambda2 = mid - disc;
                const float rad_f = ceilf(3.0f * sqrtf(fmaxf(lambda1, lambda2)));
                const int rad = rad_f >= 2147483520.0f ? 2147483520 : (int)rad_f;
                const float pix_x = ((ndc_x + 1.0f) * (float)cam_w - 1.0f) * 0.5f;
                const float pix_y = ((ndc_y + 1.0f) * (float)cam_h - 1.0f) * 0.5f;
                const TileRect r = tile_rect(pix_x, pix_y, rad, cam_gx, cam_gy);
                const int w = r.maxx - r.minx, h = r.maxy - r.miny;
                if (w > 0 && h > 0) {
                    float3 rgb;
                    if (sc.colors_precomp) {
                        rgb = make_float3(sc.colors_precomp[3 * (size_t)i], sc.colors_precomp[3 * (size_t)i + 1],
                                          sc.colors_precomp[3 * (size_t)i + 2]);
                    } else {
                        if (!have_sh) {
                            if constexpr (SPLIT_SH)
                                load_sh_split<DEG>(sh, sc.shs + (size_t)i * 3, sc.shs_rest + (size_t)i * (sc.sh_stride - 1) * 3);
                            else
                                load_sh<DEG>(sh, sc.shs + (size_t)i * sc.sh_stride * 3, vec4);
                            have_sh = true;
                        }
                        float dx = px - cam_cx, dy = py - cam_cy, dz = pz - cam_cz;
                        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
                        dx = dx / len; dy = dy / len; dz = dz / len;
                        if (POSED && P) {        // the object's own frame: R^T d
                            const float ox = fmaf(P[6], dz, fmaf(P[3], dy, P[0] * dx));
                            const float oy = fmaf(P[7], dz, fmaf(P[4], dy, P[1] * dx));
                            const float oz = fmaf(P[8], dz, fmaf(P[5], dy, P[2] * dx));
                            dx = ox; dy = oy; dz = oz;
                        }
                        rgb = sh_regs_to_rgb<DEG>(sh, dx, dy, dz);
                    }
                    radius = rad;
                    rect = make_uint2((uint32_t)r.minx | ((uint32_t)r.miny << 16),
                                      (uint32_t)r.maxx | ((uint32_t)r.maxy << 16));
                    const float op = sc.opacities[i];
                    crect = candidate_rect(r, pix_x, pix_y, c_xx, c_yy, con_x, con_y, con_z, op, rad);
                    if (LAYERED) {           // tile rows of layer k start at (k - 1) * grid_y
                        const uint32_t off = (uint32_t)((layer - 1) * cam_gy) << 16;
                        rect.x += off; rect.y += off;
                        if (crect.x | crect.y) { crect.x += off; crect.y += off; }
                    }
                    float4* rec = out_splats + (size_t)i * SPLAT_F4;
                    rec[0] = make_float4(pix_x, pix_y, con_x, con_y);
                    rec[1] = make_float4(con_z, op, con_y / con_z, con_y / con_x);     // + the cull record's B/C, B/A (cull.hip.h)
                    rec[2] = make_float4(rgb.x, rgb.y, rgb.z, tz);
                }
            }
        }
        if (o.radii) o.radii[i] = radius;
        if (o.rects) o.rects[i] = rect;
        out_crects[i] = crect;
    }
}

// packs the callers' four device-side camera tensors + host scalars into CameraDev records, one workgroup per
// view (up to CAM_PACK_MAX views per launch; the pointers travel in the launch arguments)
constexpr int CAM_PACK_MAX = 32;
struct CamPack {
    const float* view[CAM_PACK_MAX];
    const float* proj[CAM_PACK_MAX];
    const float* campos[CAM_PACK_MAX];
    const float* bg[CAM_PACK_MAX];
    float tanfovx[CAM_PACK_MAX], tanfovy[CAM_PACK_MAX];
    int32_t depth_mode[CAM_PACK_MAX];
};

__device__ __forceinline__ void pack_camera(const CamPack& p, int v, int t, int width, int height, CameraDev* __restrict__ out) {
    if (t < 16) {
        out->view[t] = p.view[v][t];
        out->proj[t] = p.proj[v][t];
    }
    if (t < 3) {
        out->campos[t] = p.campos[v][t];
        out->bg[t] = p.bg[v][t];
    }
    if (t == 0) {
        const float tanfovx = p.tanfovx[v], tanfovy = p.tanfovy[v];
        out->tanfovx = tanfovx;
        out->tanfovy = tanfovy;
        out->focal_x = (float)width / (2.0f * tanfovx);
        out->focal_y = (float)height / (2.0f * tanfovy);
        out->width = width;
        out->height = height;
        out->grid_x = (width + TILE - 1) / TILE;
        out->grid_y = (height + TILE - 1) / TILE;
        out->depth_mode = p.depth_mode[v];
    }
}

__global__ void pack_camera_kernel(CamPack p, int width, int height, CameraDev* __restrict__ outs) {
    pack_camera(p, (int)blockIdx.x, (int)threadIdx.x, width, height, outs + blockIdx.x);
}

__global__ void mark_visible_kernel(int n, const float* __restrict__ means3d, const float* __restrict__ vm,
                                    uint8_t* __restrict__ present) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float px = means3d[3 * i], py = means3d[3 * i + 1], pz = means3d[3 * i + 2];
    const float tz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
    present[i] = tz > NEAR_Z ? 1 : 0;
}

}  // namespace pgr
