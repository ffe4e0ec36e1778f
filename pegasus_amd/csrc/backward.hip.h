// backward.hip.h -- backward pass of the rasterizer (the in-tree user is training:
// /root/reference/src/gs/gs_training.py:7,46).  Same conventions as oracle/pgr_oracle_backward.c: the 0.99 clamp
// of alpha is not differentiated, the screen-space mean gradient is NDC-scaled (pixel gradient * W/2, H/2),
// colours clamped at 0 and clamped view-space coordinates pass no gradient.
//
//   composite_backward_block_kernel  one wave per 4x4-pixel block, four list entries per step (below): the ten
//       per-Gaussian partial gradients of an entry are summed over the block's pixels and land as one 10-lane float
//       atomic on the Gaussian's row.
//   preprocess_backward_kernel      one thread per Gaussian: conic -> cov2D -> (cov3D, view-space mean) ->
//       scale / rotation; projection; SH -> coefficients and view direction.
#pragma once
#include <type_traits>
#include "composite.hip.h"
#include "pgr_common.h"
#include "preprocess.hip.h"

namespace pgr {

// per-Gaussian accumulator row written by the compositor backward (same shape as the forward record)
//   [0] d/dx_pix  [1] d/dy_pix  [2] d/dA  [3] d/dB  [4] d/dC  [5] d/dopacity  [6..8] d/drgb  [9] d/dz
constexpr int GRAD_ROW = 12;

// ---- compositor backward: one wave per 4x4-pixel block, four list entries per step ------------------------------
// The tile's list is walked BACK to front from the deepest contributor of any pixel of the block, in 64-entry batches
// gathered one batch ahead (composite.hip.h's gather); only the entries the forward's skip test lets through for this
// block are parked (an entry that cannot reach alpha >= 1/255 in the block was blended by none of its pixels).
// With ONE view per launch (training) a kernel lasts as long as its longest wave; the first round-2 form -- one wave
// per 8x8 quarter, one entry per step, 0.72 ms on the 2 M-Gaussian scene after 1.97 ms for round 1's 16x8 half tiles
// with two pixels per lane -- spent 1 000 dependent steps on a quarter whose pixels blended 1 000 entries.  Here
// lane = 4 * pixel + slot, and the four lanes of a pixel take four CONSECUTIVE parked entries at once.  What is sequential in the walk -- T <- T / (1 - alpha) and the suffix sums
// of colour and depth -- is a prefix product and four prefix sums over the four lanes of a quad (two quad_perm steps
// each); everything else is per entry.  The ten partials are summed over the block's 16 pixels (two row shifts, then a
// 640-byte LDS transpose that also lines the 4 x 10 totals up for ONE atomic instruction).
__global__ __launch_bounds__(WAVE) void composite_backward_block_kernel(
    const CameraDev* __restrict__ camp, const uint2* __restrict__ ranges, const uint32_t* __restrict__ gauss_sorted,
    const float4* __restrict__ splats, const float* __restrict__ final_T, const uint32_t* __restrict__ n_contrib,
    const float* __restrict__ g_color, const float* __restrict__ g_depth, float* __restrict__ g_rows,
    const uint32_t* __restrict__ work_order) {
    const uint32_t item = work_order ? work_order[blockIdx.x >> 2] : (blockIdx.x >> 2);
    if (item == INVALID_ITEM) return;
    const CameraDev& cam = *camp;
    const int W = cam.width, H = cam.height;
    const int tile = (int)(item >> 2), quarter = (int)(item & 3), sub = (int)(blockIdx.x & 3);
    const int tile_x = tile % cam.grid_x, tile_y = tile / cam.grid_x;
    const int lane = threadIdx.x, slot = lane & 3, pl = lane >> 2;
    const int bx0 = tile_x * TILE + (quarter & 1) * 8 + (sub & 1) * 4, by0 = tile_y * TILE + (quarter >> 1) * 8 + (sub >> 1) * 4;
    if (bx0 >= W || by0 >= H) return;
    const int px = bx0 + (pl & 3), py = by0 + (pl >> 2);
    const bool inside = px < W && py < H;
    const size_t P = (size_t)W * H;
    const size_t pix = inside ? (size_t)py * W + px : 0;
    const uint2 range = ranges[tile];

    const uint32_t last = inside ? n_contrib[pix] : 0u;
    float T = inside ? final_T[pix] : 0.0f;        // per-pixel state, the same in the four lanes of a pixel
    float S[3], gC[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        gC[c] = inside ? g_color[c * P + pix] : 0.0f;
        S[c] = T * cam.bg[c];
    }
    const float gD = (inside && g_depth) ? g_depth[pix] : 0.0f;
    float SD = 0.0f;
    uint32_t n_used = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) n_used = max(n_used, (uint32_t)__shfl_xor((int)n_used, d, WAVE));
    n_used = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_used);
    if (n_used == 0) return;

    __shared__ float4 s_q0[WAVE_BATCH], s_q1[WAVE_BATCH], s_q2[WAVE_BATCH];
    __shared__ uint2 s_gi[WAVE_BATCH];           // (Gaussian, list position)
    __shared__ float s_red[10 * 16];             // [value][row of the wave][slot]
    const float pxf = (float)px, pyf = (float)py;
    const float rx0 = (float)bx0, ry0 = (float)by0;
    const float rx1 = fminf(rx0 + 3.0f, (float)(W - 1)), ry1 = fminf(ry0 + 3.0f, (float)(H - 1));
    const bool ge1 = slot >= 1, ge2 = slot >= 2;
    const int out_slot = lane / 10, out_k = lane - out_slot * 10;      // lanes 0..39: (entry slot, value) of the atomic

    f32x4_t q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0, q2 = q0;
    uint32_t g_cur = 0, g_next = 0;
    auto fetch_index = [&](int base) {
        const int i = base + lane;
        return (base >= 0 && i < (int)n_used) ? gload(gauss_sorted + range.x + i) : 0u;
    };
    auto fetch_record = [&](int base, uint32_t g) {
        g_cur = g;
        if (base >= 0 && base + lane < (int)n_used) {
            const float4* rec = splats + (size_t)g * 3;
            q0 = gload_quad(rec); q1 = gload_quad(rec + 1); q2 = gload_quad(rec + 2);
        }
    };
    // quad_perm lane exchange inside the four lanes of a pixel
    auto quad = [](float v, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    using QP_0012 = std::integral_constant<int, 0x90>;    // quad_perm [0,0,1,2]: the lane one slot down
    using QP_0101 = std::integral_constant<int, 0x44>;    // quad_perm [0,1,0,1]: the lane two slots down
    using QP_3333 = std::integral_constant<int, 0xFF>;    // quad_perm [3,3,3,3]: the pixel's last slot
    auto prefix_prod = [&](float x) {
        const float a = quad(x, QP_0012{}); x = ge1 ? x * a : x;
        const float b = quad(x, QP_0101{}); x = ge2 ? x * b : x;
        return x;
    };
    auto prefix_sum = [&](float x) {
        const float a = quad(x, QP_0012{}); x = ge1 ? x + a : x;
        const float b = quad(x, QP_0101{}); x = ge2 ? x + b : x;
        return x;
    };

    const int first = (int)((n_used - 1) / WAVE_BATCH) * WAVE_BATCH;
    fetch_record(first, fetch_index(first));
    g_next = fetch_index(first - WAVE_BATCH);

    for (int base = first; base >= 0; base -= WAVE_BATCH) {
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(g_cur));      // the quads are taken as they arrive, here
        const bool have = base + lane < (int)n_used;
        const bool live = have && rect_may_contribute(make_cull_splat(make_float2(q0.x, q0.y), make_float4(q0.z, q0.w, q1.x, q1.y),
                                                                      q1.z, q1.w), rx0, ry0, rx1, ry1);
        const unsigned long long mask = __ballot(live);
        const int cnt = __popcll(mask);
        if (live) {
            const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            s_q0[pos] = make_float4(q0.x, q0.y, q0.z, q0.w);
            s_q1[pos] = make_float4(q1.x, q1.y, q1.z, q1.w);
            s_q2[pos] = make_float4(q2.x, q2.y, q2.z, q2.w);
            s_gi[pos] = make_uint2(g_cur, (uint32_t)(base + lane));
        }
        __syncthreads();
        fetch_record(base - WAVE_BATCH, g_next);      // lands while this batch is walked
        g_next = fetch_index(base - 2 * WAVE_BATCH);
        for (int hi = cnt - 1; hi >= 0; hi -= 4) {     // slot 0 takes entry hi (the deepest), slot 3 entry hi - 3
            const int j = hi - slot;
            const int jc = j < 0 ? 0 : j;
            const float4 e0 = s_q0[jc], e1 = s_q1[jc], e2 = s_q2[jc];
            const uint2 gi = s_gi[jc];
            const float A = e0.z, B = e0.w, Cc = e1.x, op = e1.y;
            const float dx = e0.x - pxf, dy = e0.y - pyf;
            const float power = fmaf(dx, fmaf(-0.5f * A, dx, -B * dy), (-0.5f * Cc * dy) * dy);
            const float G = __builtin_amdgcn_exp2f(power * 1.4426950408889634f);
            const float alpha = fminf(ALPHA_MAX, op * G);
            const bool valid = j >= 0 && gi.y < last && !(power > 0.0f) && !(alpha < ALPHA_MIN);
            if (__builtin_amdgcn_ballot_w64(valid) == 0ull) continue;
            const float av = valid ? alpha : 0.0f;
            const float r = valid ? 1.0f / (1.0f - alpha) : 1.0f;
            const float Tin = T * prefix_prod(r);          // transmittance in front of this lane's entry
            const float w = av * Tin;
            const float col[3] = {e2.x, e2.y, e2.z};
            const float z = e2.w;
            float cin[4], cinc[4];
#pragma unroll
            for (int c = 0; c < 3; ++c) cin[c] = w * col[c];
            cin[3] = w * z;
#pragma unroll
            for (int c = 0; c < 4; ++c) cinc[c] = prefix_sum(cin[c]);
            float acc[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) acc[k] = 0.0f;
            if (valid) {
                float dL_dalpha = 0.0f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float Sb = S[c] + (cinc[c] - cin[c]);         // the sum behind this lane's entry
                    dL_dalpha += gC[c] * (Tin * col[c] - Sb * r);
                    acc[6 + c] = w * gC[c];
                }
                dL_dalpha += gD * (Tin * z - (SD + (cinc[3] - cin[3])) * r);
                acc[9] = w * gD;
                acc[5] = G * dL_dalpha;
                const float dLp = G * op * dL_dalpha;    // dL/dpower
                acc[2] = -0.5f * dx * dx * dLp;
                acc[3] = -dx * dy * dLp;
                acc[4] = -0.5f * dy * dy * dLp;
                acc[0] = -(A * dx + B * dy) * dLp;
                acc[1] = -(Cc * dy + B * dx) * dLp;
            }
            // the pixel's state after these four entries: what its last slot holds
            T = quad(Tin, QP_3333{});
#pragma unroll
            for (int c = 0; c < 3; ++c) S[c] += quad(cinc[c], QP_3333{});
            SD += quad(cinc[3], QP_3333{});
            // sums over the 16 pixels: lanes 4 apart inside a row of 16, then the four rows through LDS
            // A DPP operand read needs two wait states after the VALU write of that register.  Both row steps and the s_nop
            // in front of them are ONE asm statement: the hazard recognizer does not look into inline asm, and between separate
            // statements the compiler may place register copies for the "+v" operands (a v_mov right in front of a DPP read of
            // its destination).  Inside the block every register's two steps are ten instructions apart.
#define PGR_ROW_STEP(CTRL)                                                                                                  \
    "v_add_f32_dpp %0, %0, %0 " CTRL "\n\tv_add_f32_dpp %1, %1, %1 " CTRL "\n\tv_add_f32_dpp %2, %2, %2 " CTRL "\n\t"            \
    "v_add_f32_dpp %3, %3, %3 " CTRL "\n\tv_add_f32_dpp %4, %4, %4 " CTRL "\n\tv_add_f32_dpp %5, %5, %5 " CTRL "\n\t"            \
    "v_add_f32_dpp %6, %6, %6 " CTRL "\n\tv_add_f32_dpp %7, %7, %7 " CTRL "\n\tv_add_f32_dpp %8, %8, %8 " CTRL "\n\t"            \
    "v_add_f32_dpp %9, %9, %9 " CTRL "\n\t"
            asm volatile("s_nop 1\n\t"
                         PGR_ROW_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                         PGR_ROW_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0")
                         : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                           "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]));
#undef PGR_ROW_STEP
            if ((lane & 12) == 12) {
#pragma unroll
                for (int k = 0; k < 10; ++k) s_red[k * 16 + (lane >> 4) * 4 + slot] = acc[k];
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < 40 && hi - out_slot >= 0) {
                const float* red = s_red + out_k * 16 + out_slot;
                const float tot = (red[0] + red[4]) + (red[8] + red[12]);
                if (tot != 0.0f) atomicAdd(g_rows + (size_t)s_gi[hi - out_slot].x * GRAD_ROW + out_k, tot);
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
    }
}

struct GradOut {
    float* means2d;    // [n,3] or NULL
    float* means3d;    // [n,3] or NULL
    float* opacities;  // [n]   or NULL
    float* colors;     // [n,3] or NULL (gradient wrt the per-Gaussian rgb)
    float* shs;        // [n,sh_stride,3] or NULL
    float* cov3d;      // [n,6] or NULL
    float* scales;     // [n,3] or NULL
    float* rotations;  // [n,4] or NULL
};

template <int DEG>
__device__ __forceinline__ void sh_basis_grad(float x, float y, float z, float b[16], float bx[16], float by[16],
                                              float bz[16]) {
    constexpr float C1 = 0.4886025119029199f;
    constexpr float C2_0 = 1.0925484305920792f, C2_1 = -1.0925484305920792f, C2_2 = 0.31539156525252005f,
                    C2_3 = -1.0925484305920792f, C2_4 = 0.5462742152960396f;
    constexpr float C3_0 = -0.5900435899266435f, C3_1 = 2.890611442640554f, C3_2 = -0.4570457994644658f,
                    C3_3 = 0.3731763325901154f, C3_4 = -0.4570457994644658f, C3_5 = 1.445305721320277f,
                    C3_6 = -0.5900435899266435f;
#pragma unroll
    for (int k = 0; k < 16; ++k) b[k] = bx[k] = by[k] = bz[k] = 0.0f;
    b[0] = 0.28209479177387814f;
    if constexpr (DEG > 0) {
        b[1] = -C1 * y; by[1] = -C1;
        b[2] = C1 * z;  bz[2] = C1;
        b[3] = -C1 * x; bx[3] = -C1;
    }
    if constexpr (DEG > 1) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        b[4] = C2_0 * xy;                  bx[4] = C2_0 * y;  by[4] = C2_0 * x;
        b[5] = C2_1 * yz;                  by[5] = C2_1 * z;  bz[5] = C2_1 * y;
        b[6] = C2_2 * (2 * zz - xx - yy);  bx[6] = C2_2 * -2 * x; by[6] = C2_2 * -2 * y; bz[6] = C2_2 * 4 * z;
        b[7] = C2_3 * xz;                  bx[7] = C2_3 * z;  bz[7] = C2_3 * x;
        b[8] = C2_4 * (xx - yy);           bx[8] = C2_4 * 2 * x; by[8] = C2_4 * -2 * y;
        if constexpr (DEG > 2) {
            b[9] = C3_0 * y * (3 * xx - yy);   bx[9] = C3_0 * 6 * xy; by[9] = C3_0 * (3 * xx - 3 * yy);
            b[10] = C3_1 * xy * z;             bx[10] = C3_1 * yz; by[10] = C3_1 * xz; bz[10] = C3_1 * xy;
            b[11] = C3_2 * y * (4 * zz - xx - yy);
            bx[11] = C3_2 * -2 * xy; by[11] = C3_2 * (4 * zz - xx - 3 * yy); bz[11] = C3_2 * 8 * yz;
            b[12] = C3_3 * z * (2 * zz - 3 * xx - 3 * yy);
            bx[12] = C3_3 * -6 * xz; by[12] = C3_3 * -6 * yz; bz[12] = C3_3 * (6 * zz - 3 * xx - 3 * yy);
            b[13] = C3_4 * x * (4 * zz - xx - yy);
            bx[13] = C3_4 * (4 * zz - 3 * xx - yy); by[13] = C3_4 * -2 * xy; bz[13] = C3_4 * 8 * xz;
            b[14] = C3_5 * z * (xx - yy);      bx[14] = C3_5 * 2 * xz; by[14] = C3_5 * -2 * yz; bz[14] = C3_5 * (xx - yy);
            b[15] = C3_6 * x * (xx - 3 * yy);  bx[15] = C3_6 * (3 * xx - 3 * yy); by[15] = C3_6 * -6 * xy;
        }
    }
}

template <int DEG>
__global__ __launch_bounds__(256) void preprocess_backward_kernel(PgrScene sc, const CameraDev* __restrict__ camp,
                                                                  const int32_t* __restrict__ radii,
                                                                  const float* __restrict__ g_rows, GradOut o) {
    const CameraDev& cam = *camp;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= sc.n) return;
    const bool live = radii[i] > 0;
    // SH coefficients (and their gradients) as twelve aligned quads per Gaussian
    const bool sh_quads = sc.shs && sc.sh_stride == 16 && (reinterpret_cast<uintptr_t>(sc.shs) & 15u) == 0 &&
                          (!o.shs || (reinterpret_cast<uintptr_t>(o.shs) & 15u) == 0);
    const float* row = g_rows + (size_t)i * GRAD_ROW;
    float gp[3] = {0.f, 0.f, 0.f};
    float gS[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float gndc[2] = {0.f, 0.f};
    float gcol[3] = {0.f, 0.f, 0.f};
    float gop = 0.f;
    const float p[3] = {sc.means3d[3 * i], sc.means3d[3 * i + 1], sc.means3d[3 * i + 2]};
    const float* vm = cam.view;
    const float* pm = cam.proj;

    if (live) {
        gop = row[5];
        gcol[0] = row[6]; gcol[1] = row[7]; gcol[2] = row[8];
        // ---- screen position
        gndc[0] = row[0] * 0.5f * (float)cam.width;
        gndc[1] = row[1] * 0.5f * (float)cam.height;
        {
            const float hx = pm[0] * p[0] + pm[4] * p[1] + pm[8] * p[2] + pm[12];
            const float hy = pm[1] * p[0] + pm[5] * p[1] + pm[9] * p[2] + pm[13];
            const float hw = pm[3] * p[0] + pm[7] * p[1] + pm[11] * p[2] + pm[15];
            const float mw = 1.0f / (hw + 0.0000001f);
#pragma unroll
            for (int k = 0; k < 3; ++k)
                gp[k] += gndc[0] * (pm[4 * k + 0] * mw - hx * mw * mw * pm[4 * k + 3]) +
                         gndc[1] * (pm[4 * k + 1] * mw - hy * mw * mw * pm[4 * k + 3]);
        }
        // ---- view-space position
        float t[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) t[r] = vm[r] * p[0] + vm[4 + r] * p[1] + vm[8 + r] * p[2] + vm[12 + r];
        float gt[3] = {0.f, 0.f, row[9]};

        // ---- 3D covariance (recomputed)
        float cov[6];
        if (sc.cov3d_precomp) {
#pragma unroll
            for (int k = 0; k < 6; ++k) cov[k] = sc.cov3d_precomp[6 * (size_t)i + k];
        } else {
            const float4 q = reinterpret_cast<const float4*>(sc.rotations)[i];
            cov3d_from_scale_rot(sc.scales[3 * i], sc.scales[3 * i + 1], sc.scales[3 * i + 2], sc.scale_modifier, q, cov);
        }
        const float S3[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
        const float fx = cam.focal_x, fy = cam.focal_y;
        const float limx = 1.3f * cam.tanfovx, limy = 1.3f * cam.tanfovy;
        const float txtz = t[0] / t[2], tytz = t[1] / t[2];
        const float cx = fminf(limx, fmaxf(-limx, txtz)) * t[2], cy = fminf(limy, fmaxf(-limy, tytz)) * t[2];
        const float xmul = (txtz < -limx || txtz > limx) ? 0.f : 1.f, ymul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
        const float j00 = fx / t[2], j02 = -fx * cx / (t[2] * t[2]), j11 = fy / t[2], j12 = -fy * cy / (t[2] * t[2]);
        float T0[3], T1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            T0[k] = j00 * vm[4 * k + 0] + j02 * vm[4 * k + 2];
            T1[k] = j11 * vm[4 * k + 1] + j12 * vm[4 * k + 2];
        }
        float a = 0.f, b = 0.f, c = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                a += T0[r] * S3[r][s] * T0[s];
                b += T0[r] * S3[r][s] * T1[s];
                c += T1[r] * S3[r][s] * T1[s];
            }
        a += LOWPASS; c += LOWPASS;
        const float det = a * c - b * b, d2 = 1.0f / (det * det);
        const float gA = row[2], gB = row[3], gCc = row[4];
        const float ga = d2 * (-c * c * gA + b * c * gB - b * b * gCc);
        const float gb = d2 * (2 * b * c * gA - (det + 2 * b * b) * gB + 2 * a * b * gCc);
        const float gc = d2 * (-b * b * gA + a * b * gB - a * a * gCc);
        gS[0] = ga * T0[0] * T0[0] + gb * T0[0] * T1[0] + gc * T1[0] * T1[0];
        gS[3] = ga * T0[1] * T0[1] + gb * T0[1] * T1[1] + gc * T1[1] * T1[1];
        gS[5] = ga * T0[2] * T0[2] + gb * T0[2] * T1[2] + gc * T1[2] * T1[2];
        gS[1] = 2 * ga * T0[0] * T0[1] + gb * (T0[0] * T1[1] + T0[1] * T1[0]) + 2 * gc * T1[0] * T1[1];
        gS[2] = 2 * ga * T0[0] * T0[2] + gb * (T0[0] * T1[2] + T0[2] * T1[0]) + 2 * gc * T1[0] * T1[2];
        gS[4] = 2 * ga * T0[1] * T0[2] + gb * (T0[1] * T1[2] + T0[2] * T1[1]) + 2 * gc * T1[1] * T1[2];
        float gj00 = 0.f, gj02 = 0.f, gj11 = 0.f, gj12 = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int s = 0; s < 3; ++s) { s0 += S3[k][s] * T0[s]; s1 += S3[k][s] * T1[s]; }
            const float gT0 = 2 * ga * s0 + gb * s1, gT1 = 2 * gc * s1 + gb * s0;
            gj00 += gT0 * vm[4 * k + 0]; gj02 += gT0 * vm[4 * k + 2];
            gj11 += gT1 * vm[4 * k + 1]; gj12 += gT1 * vm[4 * k + 2];
        }
        const float tz2 = 1.0f / (t[2] * t[2]), tz3 = tz2 / t[2];
        gt[0] += xmul * -fx * tz2 * gj02;
        gt[1] += ymul * -fy * tz2 * gj12;
        gt[2] += -fx * tz2 * gj00 - fy * tz2 * gj11 + 2 * fx * cx * tz3 * gj02 + 2 * fy * cy * tz3 * gj12;
        if (xmul == 0.f) gt[2] += -fx * tz2 * gj02 * (cx / t[2]);
        if (ymul == 0.f) gt[2] += -fy * tz2 * gj12 * (cy / t[2]);
#pragma unroll
        for (int k = 0; k < 3; ++k) gp[k] += vm[4 * k + 0] * gt[0] + vm[4 * k + 1] * gt[1] + vm[4 * k + 2] * gt[2];

        // ---- colour: SH coefficients and the view direction
        if (sc.shs) {
            float d[3] = {p[0] - cam.campos[0], p[1] - cam.campos[1], p[2] - cam.campos[2]};
            const float len = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            const float u[3] = {d[0] / len, d[1] / len, d[2] / len};
            float bb[16], bx[16], by[16], bz[16];
            sh_basis_grad<DEG>(u[0], u[1], u[2], bb, bx, by, bz);
            constexpr int NC = (DEG + 1) * (DEG + 1);
            const float* sh = sc.shs + (size_t)i * sc.sh_stride * 3;
            float gu[3] = {0.f, 0.f, 0.f};
            if (sh_quads) {
                // the common layout (16 coefficients x rgb = twelve aligned quads per Gaussian): coefficients in and
                // gradients out as 16-byte accesses -- the scalar form is 48 four-byte loads (three times over) and 48
                // four-byte stores per thread, each touching 64 different cache lines per instruction
                float shv[48], outv[48];
#pragma unroll
                for (int q = 0; q < 12; ++q) {
                    const float4 v = reinterpret_cast<const float4*>(sh)[q];
                    shv[4 * q] = v.x; shv[4 * q + 1] = v.y; shv[4 * q + 2] = v.z; shv[4 * q + 3] = v.w;
                }
#pragma unroll
                for (int k = 0; k < 48; ++k) outv[k] = 0.f;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    float accv = 0.f;
#pragma unroll
                    for (int k = 0; k < NC; ++k) accv += bb[k] * shv[3 * k + ch];
                    const float g = (accv + 0.5f < 0.0f) ? 0.f : gcol[ch];
#pragma unroll
                    for (int k = 0; k < NC; ++k) {
                        outv[3 * k + ch] = bb[k] * g;
                        gu[0] += g * shv[3 * k + ch] * bx[k];
                        gu[1] += g * shv[3 * k + ch] * by[k];
                        gu[2] += g * shv[3 * k + ch] * bz[k];
                    }
                }
                if (o.shs) {
                    float4* dst = reinterpret_cast<float4*>(o.shs + (size_t)i * 48);
#pragma unroll
                    for (int q = 0; q < 12; ++q) dst[q] = make_float4(outv[4 * q], outv[4 * q + 1], outv[4 * q + 2], outv[4 * q + 3]);
                }
            } else {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float accv = 0.f;
#pragma unroll
                for (int k = 0; k < NC; ++k) accv += bb[k] * sh[3 * k + ch];
                const float g = (accv + 0.5f < 0.0f) ? 0.f : gcol[ch];
#pragma unroll
                for (int k = 0; k < NC; ++k) {
                    if (o.shs) o.shs[((size_t)i * sc.sh_stride + k) * 3 + ch] = bb[k] * g;
                    gu[0] += g * sh[3 * k + ch] * bx[k];
                    gu[1] += g * sh[3 * k + ch] * by[k];
                    gu[2] += g * sh[3 * k + ch] * bz[k];
                }
            }
            }
            const float dot = u[0] * gu[0] + u[1] * gu[1] + u[2] * gu[2];
#pragma unroll
            for (int k = 0; k < 3; ++k) gp[k] += (gu[k] - u[k] * dot) / len;
        }
    } else if (o.shs) {
        if (sh_quads) {
            float4* dst = reinterpret_cast<float4*>(o.shs + (size_t)i * 48);
#pragma unroll
            for (int q = 0; q < 12; ++q) dst[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            const int nf = sc.sh_stride * 3;
            for (int k = 0; k < nf; ++k) o.shs[(size_t)i * nf + k] = 0.f;
        }
    }
    if (live && o.shs && !sh_quads) {   // coefficients above the active degree receive no gradient
        constexpr int NC = (DEG + 1) * (DEG + 1);
        for (int k = NC; k < sc.sh_stride; ++k)
            for (int ch = 0; ch < 3; ++ch) o.shs[((size_t)i * sc.sh_stride + k) * 3 + ch] = 0.f;
    }

    // ---- cov3D -> scale, rotation
    if (o.scales || o.rotations) {
        float gs[3] = {0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f};
        if (live && sc.scales && sc.rotations) {
            const float4 q4 = reinterpret_cast<const float4*>(sc.rotations)[i];
            const float r = q4.x, x = q4.y, y = q4.z, z = q4.w;
            const float R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)},
                                   {2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)},
                                   {2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)}};
            const float s[3] = {sc.scale_modifier * sc.scales[3 * i], sc.scale_modifier * sc.scales[3 * i + 1],
                                sc.scale_modifier * sc.scales[3 * i + 2]};
            const float Gf[3][3] = {{gS[0], 0.5f * gS[1], 0.5f * gS[2]}, {0.5f * gS[1], gS[3], 0.5f * gS[4]},
                                    {0.5f * gS[2], 0.5f * gS[4], gS[5]}};
            float gR[3][3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float accs = 0.f;
#pragma unroll
                for (int a_ = 0; a_ < 3; ++a_) {
                    float gM = 0.f;
#pragma unroll
                    for (int m = 0; m < 3; ++m) gM += 2 * Gf[a_][m] * R[m][k] * s[k];
                    accs += gM * R[a_][k];
                    gR[a_][k] = gM * s[k];
                }
                gs[k] = accs * sc.scale_modifier;
            }
            gq[0] = 2 * (-z * gR[0][1] + y * gR[0][2] + z * gR[1][0] - x * gR[1][2] - y * gR[2][0] + x * gR[2][1]);
            gq[1] = 2 * (y * gR[0][1] + z * gR[0][2] + y * gR[1][0] - 2 * x * gR[1][1] - r * gR[1][2] + z * gR[2][0] +
                         r * gR[2][1] - 2 * x * gR[2][2]);
            gq[2] = 2 * (-2 * y * gR[0][0] + x * gR[0][1] + r * gR[0][2] + x * gR[1][0] + z * gR[1][2] - r * gR[2][0] +
                         z * gR[2][1] - 2 * y * gR[2][2]);
            gq[3] = 2 * (-2 * z * gR[0][0] - r * gR[0][1] + x * gR[0][2] + r * gR[1][0] - 2 * z * gR[1][1] + y * gR[1][2] +
                         x * gR[2][0] + y * gR[2][1]);
        }
        if (o.scales) { o.scales[3 * i] = gs[0]; o.scales[3 * i + 1] = gs[1]; o.scales[3 * i + 2] = gs[2]; }
        if (o.rotations) reinterpret_cast<float4*>(o.rotations)[i] = make_float4(gq[0], gq[1], gq[2], gq[3]);
    }
    if (o.means2d) { o.means2d[3 * i] = gndc[0]; o.means2d[3 * i + 1] = gndc[1]; o.means2d[3 * i + 2] = 0.f; }
    if (o.means3d) { o.means3d[3 * i] = gp[0]; o.means3d[3 * i + 1] = gp[1]; o.means3d[3 * i + 2] = gp[2]; }
    if (o.opacities) o.opacities[i] = gop;
    if (o.colors) { o.colors[3 * i] = gcol[0]; o.colors[3 * i + 1] = gcol[1]; o.colors[3 * i + 2] = gcol[2]; }
    if (o.cov3d) {
#pragma unroll
        for (int k = 0; k < 6; ++k) o.cov3d[6 * (size_t)i + k] = gS[k];
    }
}

}  // namespace pgr
