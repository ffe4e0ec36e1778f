// blockcull.hip.h -- conservative per-(view, 64-Gaussian block) visibility, so that the per-Gaussian stages skip
// whole waves whose Gaussians cannot be listed anywhere.
//
// A Gaussian leaves no trace in a view (radius 0, no tile rectangle, no list entry) when its view-space z is
// <= 0.2 or its tile rectangle is empty.  PEGASUS scenes are a room-sized environment
// plus a few objects the cameras look at (/root/reference/pegasus.py:166-245): most environment Gaussians are
// off-screen in every single view, yet each of them costs the preprocess ~200 VALU instructions per view (projection,
// six correctly rounded divisions, EWA covariance, eigenvalues) before its rectangle turns out empty, and both binning
// walks read its (empty) rectangle.  With the scene in Morton order (pegasus_amd/scene_order.py) 64 consecutive
// Gaussians -- one wave of the preprocess, one group of the binning walk -- fill a compact box, and a box that
// lies outside the view frustum by more than the largest radius any of its Gaussians can have is skipped as a whole.
//
// Result-preserving by construction: the test only ever claims "every Gaussian of this block is culled by the exact
// per-Gaussian code", with margins far above fp32 rounding, and answers "visible" whenever it cannot tell (box across
// the near plane, non-finite numbers, posed objects).  Any storage order is correct; a spatially incoherent one just
// has boxes too big to cull.  tests/test_block_cull.py: bit-identical radii, lists and images with the test on and off.
#pragma once
#include "pgr_common.h"

namespace pgr {

// Per block: box of the means, and the largest trace of a 3D covariance (an upper bound of its largest eigenvalue).
struct alignas(16) BlockBounds {
    float lo[3], sig2;       // sig2 = max trace(Sigma3D) * scale_modifier^2 ; +inf = never cull
    float hi[3], pad;
};
static_assert(sizeof(BlockBounds) == 32, "BlockBounds layout");

// wave-wide min / max, result in lane 63 (DPP row shifts + the two row broadcasts, like pgr_common.h's prefix sums;
// out-of-row reads return the identity through bound_ctrl = 0 -> the old value)
__device__ __forceinline__ float dpp_f32(float old, float v, int ctrl) {
    switch (ctrl) {
        case 0x111: return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x111, 0xF, 0xF, false));
        case 0x112: return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x112, 0xF, 0xF, false));
        case 0x114: return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x114, 0xF, 0xF, false));
        case 0x118: return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x118, 0xF, 0xF, false));
        case 0x142: return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x142, 0xA, 0xF, false));
        default:    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), 0x143, 0xC, 0xF, false));
    }
}
template <bool MAX>
__device__ __forceinline__ float wave_reduce_to_last(float v) {
    const int ctrls[6] = {0x111, 0x112, 0x114, 0x118, 0x142, 0x143};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float o = dpp_f32(v, v, ctrls[k]);      // lanes without a source keep their own value
        v = MAX ? fmaxf(v, o) : fminf(v, o);
    }
    return v;
}
__device__ __forceinline__ float wave_min_f(float v) { return wave_reduce_to_last<false>(v); }
__device__ __forceinline__ float wave_max_f(float v) { return wave_reduce_to_last<true>(v); }

// Bounds of the 64 Gaussians a wave holds (lane -> Gaussian i, `real` = i < n), identical in every lane.
// posed_oid != NULL: Gaussians of posed objects move per view -- their blocks are never culled.
__device__ __forceinline__ BlockBounds wave_block_bounds(const PgrScene& sc, int i, bool real,
                                                         const int32_t* __restrict__ posed_oid) {
    const float INF = __builtin_inff();
    float x0 = INF, y0 = INF, z0 = INF, x1 = -INF, y1 = -INF, z1 = -INF, s2 = 0.0f;
    bool bad = false;                       // a non-finite number anywhere: no claim about this block
    if (real) {
        const float x = sc.means3d[3 * i], y = sc.means3d[3 * i + 1], z = sc.means3d[3 * i + 2];
        x0 = x1 = x; y0 = y1 = y; z0 = z1 = z;
        float tr;
        if (sc.cov3d_precomp) {
            // caller-supplied matrix, not necessarily positive semi-definite: Frobenius norm >= spectral norm
            const float* c = sc.cov3d_precomp + 6 * (size_t)i;
            tr = sqrtf(c[0] * c[0] + c[3] * c[3] + c[5] * c[5] + 2.0f * (c[1] * c[1] + c[2] * c[2] + c[4] * c[4]));
        } else {
            // trace(R S S R^T) = sum_j s_j^2 |column j of R|^2 with R built exactly like the preprocess builds it
            // (NOT assumed orthonormal: an unnormalised quaternion scales it)
            const float4 q = reinterpret_cast<const float4*>(sc.rotations)[i];
            const float r = q.x, qx = q.y, qy = q.z, qz = q.w;
            const float R00 = 1.0f - 2.0f * (qy * qy + qz * qz), R01 = 2.0f * (qx * qy - r * qz), R02 = 2.0f * (qx * qz + r * qy);
            const float R10 = 2.0f * (qx * qy + r * qz), R11 = 1.0f - 2.0f * (qx * qx + qz * qz), R12 = 2.0f * (qy * qz - r * qx);
            const float R20 = 2.0f * (qx * qz - r * qy), R21 = 2.0f * (qy * qz + r * qx), R22 = 1.0f - 2.0f * (qx * qx + qy * qy);
            const float sx = sc.scale_modifier * sc.scales[3 * i], sy = sc.scale_modifier * sc.scales[3 * i + 1],
                        sz = sc.scale_modifier * sc.scales[3 * i + 2];
            tr = sx * sx * (R00 * R00 + R10 * R10 + R20 * R20) + sy * sy * (R01 * R01 + R11 * R11 + R21 * R21) +
                 sz * sz * (R02 * R02 + R12 * R12 + R22 * R22);
        }
        s2 = tr;
        bad = !(fabsf(x) < INF) || !(fabsf(y) < INF) || !(fabsf(z) < INF) || !(tr < INF) ||
              (posed_oid && posed_oid[i] > 0);
    }
    auto last = [](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), WAVE - 1)); };
    BlockBounds b;
    b.lo[0] = last(wave_min_f(x0)); b.lo[1] = last(wave_min_f(y0)); b.lo[2] = last(wave_min_f(z0));
    b.hi[0] = last(wave_max_f(x1)); b.hi[1] = last(wave_max_f(y1)); b.hi[2] = last(wave_max_f(z1));
    b.sig2 = __ballot(bad) != 0ull ? INF : last(wave_max_f(s2));
    b.pad = 0.0f;
    return b;
}

// One wave per block (the stand-alone form behind pgr_block_visibility; the batch path computes the same bounds inside
// its preprocess waves).
__global__ __launch_bounds__(256) void block_bounds_kernel(PgrScene sc, const int32_t* __restrict__ posed_oid,
                                                           BlockBounds* __restrict__ out, int n_groups) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int group = blockIdx.x * (256 / WAVE) + (int)(threadIdx.x / WAVE);
    if (group >= n_groups) return;
    const int i = group * WAVE + lane;
    const BlockBounds b = wave_block_bounds(sc, i, i < sc.n, posed_oid);
    if (lane == 0) out[group] = b;
}

// Is every Gaussian of the box certainly culled in this view?
__device__ __forceinline__ bool block_is_culled(const BlockBounds& b, const CameraDev& cam) {
    if (!(b.sig2 < 3.0e38f)) return false;
    const float* vm = cam.view;
    const float* pm = cam.proj;
    float zmin = 3.0e38f, zmax = -3.0e38f, wmin = 3.0e38f;
    float pxmin = 3.0e38f, pxmax = -3.0e38f, pymin = 3.0e38f, pymax = -3.0e38f;
    float mag = 0.0f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float x = (c & 1) ? b.hi[0] : b.lo[0], y = (c & 2) ? b.hi[1] : b.lo[1], z = (c & 4) ? b.hi[2] : b.lo[2];
        const float tz = vm[2] * x + vm[6] * y + vm[10] * z + vm[14];
        mag = fmaxf(mag, fabsf(vm[2] * x) + fabsf(vm[6] * y) + fabsf(vm[10] * z) + fabsf(vm[14]));
        const float hx = pm[0] * x + pm[4] * y + pm[8] * z + pm[12];
        const float hy = pm[1] * x + pm[5] * y + pm[9] * z + pm[13];
        const float hw = pm[3] * x + pm[7] * y + pm[11] * z + pm[15];
        zmin = fminf(zmin, tz); zmax = fmaxf(zmax, tz); wmin = fminf(wmin, hw);
        const float p_w = 1.0f / (hw + 0.0000001f);
        const float px = ((hx * p_w + 1.0f) * (float)cam.width - 1.0f) * 0.5f;
        const float py = ((hy * p_w + 1.0f) * (float)cam.height - 1.0f) * 0.5f;
        pxmin = fminf(pxmin, px); pxmax = fmaxf(pxmax, px);
        pymin = fminf(pymin, py); pymax = fmaxf(pymax, py);
    }
    const float eps_z = 1.0e-5f * mag + 1.0e-6f;
    // (1) the whole box at or behind the near plane: view z is linear, its range over the box is the corners' range
    if (zmax + eps_z <= NEAR_Z) return true;
    // (2) the whole box in front of it, with positive homogeneous w: pixel coordinates are linear-fractional in the
    // position, so their range over the box is the corners' range too
    if (!(zmin - eps_z > NEAR_Z) || !(wmin > 0.01f)) return false;
    // radius bound: ceil(3 sqrt(lambda1)) with lambda1 <= lambda_max(cov2D) + 0.32 (the 0.1 floor under the root), and
    //   lambda_max(cov2D) <= |J|_2^2 |W|_2^2 lambda_max(Sigma) + 0.3,   lambda_max(Sigma) <= trace(Sigma) <= sig2,
    //   |W|_2^2 <= largest absolute row sum of W^T W (Gershgorin; exactly 1 for a rigid view matrix -- the Frobenius norm
    //             used here before says 3),
    //   |J|_2^2 = lambda_max(J J^T) <= max(a, b) + |c| with a = fx^2 (1 + tx^2) / z^2, b = fy^2 (1 + ty^2) / z^2,
    //             c = fx fy tx ty / z^2, |tx| <= 1.3 tanfovx, |ty| <= 1.3 tanfovy (x/z, y/z are clamped there), z >= zmin
    //             (the sum of both rows' norms used before says a + b).
    float m00 = 0.f, m01 = 0.f, m02 = 0.f, m11 = 0.f, m12 = 0.f, m22 = 0.f;       // W^T W, W[r][c] = vm[4 c + r]
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float w0 = vm[r], w1 = vm[4 + r], w2 = vm[8 + r];
        m00 = fmaf(w0, w0, m00); m01 = fmaf(w0, w1, m01); m02 = fmaf(w0, w2, m02);
        m11 = fmaf(w1, w1, m11); m12 = fmaf(w1, w2, m12); m22 = fmaf(w2, w2, m22);
    }
    const float wn = fmaxf(fmaxf(fabsf(m00) + fabsf(m01) + fabsf(m02), fabsf(m01) + fabsf(m11) + fabsf(m12)),
                           fabsf(m02) + fabsf(m12) + fabsf(m22));
    const float lx = 1.3f * cam.tanfovx, ly = 1.3f * cam.tanfovy;
    const float zr = 1.0f / (zmin - eps_z);
    const float fx2 = cam.focal_x * cam.focal_x * (1.0f + lx * lx), fy2 = cam.focal_y * cam.focal_y * (1.0f + ly * ly);
    const float jn = (fmaxf(fx2, fy2) + fabsf(cam.focal_x * cam.focal_y) * lx * ly) * zr * zr;
    const float tr2 = 1.02f * b.sig2 * wn * jn + 0.92f;
    const float rmax = 3.0f * sqrtf(tr2) * 1.001f + 2.0f;
    if (!(rmax < 1.0e9f)) return false;
    const float mx = 1.0f + 1.0e-4f * fmaxf(fabsf(pxmin), fabsf(pxmax)), my = 1.0f + 1.0e-4f * fmaxf(fabsf(pymin), fabsf(pymax));
    // empty rectangle: x + r < 1 (left of the image), x - r >= 16 grid_x (right of it); same in y.  NaN -> false.
    if (pxmax + rmax + mx < 1.0f) return true;
    if (pxmin - rmax - mx >= (float)(TILE * cam.grid_x)) return true;
    if (pymax + rmax + my < 1.0f) return true;
    if (pymin - rmax - my >= (float)(TILE * cam.grid_y)) return true;
    return false;
}

// vis[group * words + w] bit k = block `group` may be visible in view 32 w + k.  One lane per (block, view): a wave
// holds two blocks x 32 views and its ballot IS the two visibility words.  grid = (ceil(groups / 8), words), 256 threads.
__global__ __launch_bounds__(256) void block_cull_kernel(const BlockBounds* __restrict__ bounds, int n_groups,
                                                         const CameraDev* __restrict__ cams, int n_views, int words,
                                                         uint32_t* __restrict__ vis) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int group = (blockIdx.x * (256 / WAVE) + (int)(threadIdx.x / WAVE)) * 2 + (lane >> 5);
    const int w = blockIdx.y, view = 32 * w + (lane & 31);
    bool visible = false;
    if (group < n_groups && view < n_views) visible = !block_is_culled(bounds[group], cams[view]);
    const unsigned long long m = __ballot(visible);
    if ((lane & 31) == 0 && group < n_groups) vis[(size_t)group * words + w] = (uint32_t)(m >> (lane & 32));
}

}  // namespace pgr
