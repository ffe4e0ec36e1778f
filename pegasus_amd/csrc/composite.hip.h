// composite.hip.h -- front-to-back alpha compositing of one 16x16 tile per workgroup.
// SURVEY.md section 8a row a10.  Same operation order as oracle/pgr_oracle.c composite_tile();
// the only non-bit-exact step is exp (v_exp_f32 here, glibc expf in the oracle).
//
// Bound: VALU + transcendental issue (not HBM, not MFMA): per (pixel, list entry) ~20 VALU ops and
// one v_exp_f32.  The tile's list is staged through LDS in batches so each entry's 44 B is
// fetched from HBM/L2 once per tile and then broadcast-read by all four waves.
#pragma once
#include "cull.hip.h"
#include "pgr_common.h"

namespace pgr {

constexpr int NUM_XCD = 8;
constexpr uint32_t INVALID_ITEM = 0xffffffffu;   // unused slot of the interleaved work order

struct CompOut {
    float* color;        // [3,H,W]
    float* depth;        // [H,W]
    float* final_T;      // optional
    uint32_t* n_contrib; // optional
};

// One entry per view of a batch, resident in HBM; read through the scalar cache (wave-uniform).
struct alignas(16) ViewEntry {
    const CameraDev* cam;
    const uint2* ranges;
    const uint32_t* gauss_sorted;
    const float4* splats;        // [n, 3] records: q0 = (x,y,A,B), q1 = (C,op,r,g), q2 = (b,depth,..)
    CompOut out;
    const uint32_t* counters;    // [1] != 0: instance overflow, the view must not be composited
    uint64_t pad[3];
};
static_assert(sizeof(ViewEntry) == 96, "ViewEntry layout");

// Fused semantic pass (same for every view of a batch): which Gaussians are objects and what colour they carry.
struct SemanticDev {
    const int32_t* object_id;    // [n] 0 = environment, k = object k
    const float* colors;         // [K,3] the rgb value object k's Gaussians carry: max(C0*RGB2SH(c_k) + 0.5, 0)
    int32_t n_env;               // Gaussians with index < n_env are environment: skipped without being loaded
    int32_t k;
};

// ---------------------------------------------------------------------------------------------
// composite_wave_kernel: ONE WAVE per half tile (16 x 8 pixels), two pixels per lane.
//
// gfx950 mapping: the two pixels of a lane (same x, rows r and r+4) are evaluated with packed
// fp32 math (v_pk_mul/fma/add_f32: two IEEE fp32 results per issue slot), every threshold is a lane
// mask instead of a branch, and non-blended lanes run the blend with alpha = 0 (exact no-op:
// fma(c, 0, C) == C, fma(-0, T, T) == T), so the per-pixel operation order and results are those of the
// oracle.  The workgroup is a single wave: no cross-wave barrier, early-out per half tile.  The list is
// consumed in batches of 64: lane j gathers entry j's 40 B (xy, conic+opacity, rgb+depth) into registers
// one batch AHEAD, parks it in LDS, and every lane then broadcast-reads the batch.
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int HALF_ROWS = 8;     // rows per half tile
constexpr int WAVE_BATCH = 64;   // list entries staged per round

// tuning knobs (measured on MI355X)
#ifndef PGR_COMP_UNROLL
#define PGR_COMP_UNROLL 8        // entries per unrolled group (divides 8)
#endif
#ifndef PGR_COMP_WAVES
#define PGR_COMP_WAVES 0         // 0: let the compiler pick; k: cap VGPRs for k waves/SIMD
#endif
#if PGR_COMP_WAVES
#define PGR_COMP_OCC __attribute__((amdgpu_waves_per_eu(PGR_COMP_WAVES, PGR_COMP_WAVES)))
#else
#define PGR_COMP_OCC
#endif

// SEM = true: PEGASUS's object-only semantic render (/root/reference/src/gs/render.py:68-97: all objects in their
// semantic colours, environment REMOVED) from the scene's own data: `views` then points at the tiles'
// OBJECT lists (tile_sort writes them as a by-product: the scene's sorted list minus the environment entries,
// which is exactly the list an objects-only cloud would produce) and every entry takes its object's flat colour
// instead of its SH colour.  No second preprocess / binning / sort; pixel arithmetic is the sequence a separate
// pass would execute, so the image is bit-identical (tests/test_gpu_parity.py).
template <bool AUX, bool SEM>
__global__ __launch_bounds__(WAVE) PGR_COMP_OCC void composite_wave_kernel(const ViewEntry* __restrict__ views,
                                                              uint32_t items_per_view,
                                                              const uint32_t* __restrict__ work_order,
                                                              SemanticDev sem) {
    uint32_t item = work_order ? work_order[blockIdx.x] : blockIdx.x;
    if (item == INVALID_ITEM) return;
    const uint32_t view = item / items_per_view;
    item -= view * items_per_view;
    const ViewEntry& ve = views[view];
    if (ve.counters[1] || !ve.out.color) return;
    const CameraDev& cam = *ve.cam;
    const uint2* __restrict__ ranges = ve.ranges;
    const uint32_t* __restrict__ gauss_sorted = ve.gauss_sorted;
    const float4* __restrict__ splats = ve.splats;
    const CompOut o = ve.out;
    const int W = cam.width, H = cam.height;
    const int tile = (int)(item >> 1), half = (int)(item & 1);
    const int tile_x = tile % cam.grid_x, tile_y = tile / cam.grid_x;
    const int lane = threadIdx.x;
    const int px = tile_x * TILE + (lane & (TILE - 1));
    const int py0 = tile_y * TILE + half * HALF_ROWS + (lane >> 4);
    const int py1 = py0 + 4;
    const bool in0 = px < W && py0 < H, in1 = px < W && py1 < H;
    const float pxf = (float)px;
    const f32x2 pyf = {(float)py0, (float)py1};

    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);

    constexpr int PADDED = WAVE_BATCH + 8;        // room for the null entries that pad a batch to a multiple of 8
    // single-buffered: the workgroup is one wave, which writes a batch, reads it, then writes the next in program order
    __shared__ float4 s_a[1][PADDED];   // x, y, hx, ny
    __shared__ float4 s_b[1][PADDED];   // hz, opacity, r, g
    __shared__ float2 s_c[1][PADDED];   // b, depth
    __shared__ uint32_t s_i[1][PADDED]; // 1-based position in the tile's list (n_contrib bookkeeping)

    f32x2 T = {1.0f, 1.0f}, Cr = {0.f, 0.f}, Cg = {0.f, 0.f}, Cb = {0.f, 0.f}, D = {0.f, 0.f};
    uint32_t last0 = 0, last1 = 0;
    bool done0 = !in0, done1 = !in1;

    // register-staged prefetch of the first batch
    float2 p = make_float2(0.f, 0.f);
    float4 co = make_float4(0.f, 0.f, 0.f, 0.f), cd = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < n) {
        const uint32_t g = gauss_sorted[range.x + lane];
        const float4* rec = splats + (size_t)g * 3;
        const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
        p = make_float2(q0.x, q0.y);
        co = make_float4(q0.z, q0.w, q1.x, q1.y);
        cd = make_float4(q1.z, q1.w, q2.x, q2.y);
        if (SEM) {
            const float* col = sem.colors + 3 * (size_t)(sem.object_id[g] - 1);
            cd = make_float4(col[0], col[1], col[2], q2.y);
        }
    }

    // the wave's pixel-centre rectangle (clipped to the image), for the per-entry skip test
    const float rx0 = (float)(tile_x * TILE), ry0 = (float)(tile_y * TILE + half * HALF_ROWS);
    const float rx1 = fminf(rx0 + (float)(TILE - 1), (float)(W - 1));
    const float ry1 = fminf(ry0 + (float)(HALF_ROWS - 1), (float)(H - 1));

    constexpr int buf = 0;
    for (int base = 0; base < n; base += WAVE_BATCH) {
        // Skip + compact: lane j decides whether ITS entry can reach alpha >= 1/255 anywhere in this wave's
        // 16x8 pixels (same conservative predicate as the binning, on the half tile).  Entries that cannot are
        // no-ops for every lane, so only the live ones are parked in LDS, compacted in list order; the batch is
        // padded to a multiple of 8 with null splats (opacity 0 -> never valid) for the unrolled loop.
        const bool live = base + lane < n && rect_may_contribute(make_cull_splat(p, co), rx0, ry0, rx1, ry1);
        const unsigned long long mask = __ballot(live);
        const int cnt = __popcll(mask);
        const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (live) {
            s_a[buf][pos] = make_float4(p.x, p.y, -0.5f * co.x, -co.y);
            s_b[buf][pos] = make_float4(-0.5f * co.z, co.w, cd.x, cd.y);
            s_c[buf][pos] = make_float2(cd.z, cd.w);
            if (AUX) s_i[buf][pos] = (uint32_t)(base + lane + 1);
        }
        if (lane < 8) {
            s_a[buf][cnt + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
            s_b[buf][cnt + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
            s_c[buf][cnt + lane] = make_float2(0.f, 0.f);
            if (AUX) s_i[buf][cnt + lane] = 0u;
        }
        __syncthreads();
        // issue the gather of the NEXT batch now; it lands while this batch is composited
        p = make_float2(0.f, 0.f);
        co = make_float4(0.f, 0.f, 0.f, 0.f);
        cd = make_float4(0.f, 0.f, 0.f, 0.f);
        if (base + WAVE_BATCH + lane < n) {
            const uint32_t g = gauss_sorted[range.x + base + WAVE_BATCH + lane];
            const float4* rec = splats + (size_t)g * 3;
            const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
            p = make_float2(q0.x, q0.y);
            co = make_float4(q0.z, q0.w, q1.x, q1.y);
            cd = make_float4(q1.z, q1.w, q2.x, q2.y);
            if (SEM) {
                const float* col = sem.colors + 3 * (size_t)(sem.object_id[g] - 1);
                cd = make_float4(col[0], col[1], col[2], q2.y);
            }
        }
        for (int j0 = 0; j0 < cnt; j0 += PGR_COMP_UNROLL) {
#pragma unroll
            for (int u = 0; u < PGR_COMP_UNROLL; ++u) {
                const int j = j0 + u;
                const float4 a = s_a[buf][j];
                const float4 b = s_b[buf][j];
                const float2 c = s_c[buf][j];
                const float dx = a.x - pxf;
                const f32x2 dxv = {dx, dx};
                const f32x2 dy = (f32x2){a.y, a.y} - pyf;
                const f32x2 t1 = (f32x2){a.w, a.w} * dy;
                const f32x2 t2 = __builtin_elementwise_fma((f32x2){a.z, a.z}, dxv, t1);
                const f32x2 t4 = ((f32x2){b.x, b.x} * dy) * dy;
                const f32x2 power = __builtin_elementwise_fma(dxv, t2, t4);
                const f32x2 p2 = power * (f32x2){1.4426950408889634f, 1.4426950408889634f};
                const f32x2 e = {__builtin_amdgcn_exp2f(p2.x), __builtin_amdgcn_exp2f(p2.y)};
                const f32x2 araw = (f32x2){b.y, b.y} * e;
                const f32x2 alpha = {fminf(ALPHA_MAX, araw.x), fminf(ALPHA_MAX, araw.y)};
                const f32x2 test_T = __builtin_elementwise_fma(-alpha, T, T);
                const bool v0 = !done0 && !(power.x > 0.0f) && !(alpha.x < ALPHA_MIN);
                const bool v1 = !done1 && !(power.y > 0.0f) && !(alpha.y < ALPHA_MIN);
                const bool stop0 = v0 && test_T.x < T_EPS, stop1 = v1 && test_T.y < T_EPS;
                done0 = done0 || stop0;
                done1 = done1 || stop1;
                const bool b0 = v0 && !stop0, b1 = v1 && !stop1;
                const f32x2 aeff = {b0 ? alpha.x : 0.0f, b1 ? alpha.y : 0.0f};
                const f32x2 w = aeff * T;
                Cr = __builtin_elementwise_fma((f32x2){b.z, b.z}, w, Cr);
                Cg = __builtin_elementwise_fma((f32x2){b.w, b.w}, w, Cg);
                Cb = __builtin_elementwise_fma((f32x2){c.x, c.x}, w, Cb);
                D = __builtin_elementwise_fma((f32x2){c.y, c.y}, w, D);
                T = __builtin_elementwise_fma(-aeff, T, T);
                if (AUX) {
                    const uint32_t idx = s_i[buf][j];
                    last0 = b0 ? idx : last0;
                    last1 = b1 ? idx : last1;
                }
            }
            if (__all(done0 && done1)) goto finished;
        }
    }
finished:
    const size_t P = (size_t)W * H;
    if (in0) {
        const size_t pix = (size_t)py0 * W + px;
        o.color[0 * P + pix] = fmaf(T.x, cam.bg[0], Cr.x);
        o.color[1 * P + pix] = fmaf(T.x, cam.bg[1], Cg.x);
        o.color[2 * P + pix] = fmaf(T.x, cam.bg[2], Cb.x);
        if (o.depth) o.depth[pix] = D.x;
        if (AUX) {
            if (o.final_T) o.final_T[pix] = T.x;
            if (o.n_contrib) o.n_contrib[pix] = last0;
        }
    }
    if (in1) {
        const size_t pix = (size_t)py1 * W + px;
        o.color[0 * P + pix] = fmaf(T.y, cam.bg[0], Cr.y);
        o.color[1 * P + pix] = fmaf(T.y, cam.bg[1], Cg.y);
        o.color[2 * P + pix] = fmaf(T.y, cam.bg[2], Cb.y);
        if (o.depth) o.depth[pix] = D.y;
        if (AUX) {
            if (o.final_T) o.final_T[pix] = T.y;
            if (o.n_contrib) o.n_contrib[pix] = last1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// composite_quarter_kernel: ONE WAVE per quarter tile (8 x 8 pixels), one pixel per lane.  Same list walk, skip +
// compaction and per-pixel arithmetic as composite_wave_kernel; the finer unit tightens both the per-entry skip
// test (8x8 instead of 16x8 pixel rectangle) and the early-out (the wave stops when ITS 64 pixels are saturated).
// gfx950 issues packed fp32 at the plain fp32 lane rate, so one pixel per lane costs the same per pixel.
#ifdef PGR_COMP_STATS
// debug build only: [0] list entries walked, [1] live entries after the skip test, [2] wave-entries evaluated,
// [3] pixel-entries with the pixel still alive, [4] pixel-entries blended, [5] waves, [6] batches
__device__ unsigned long long g_comp_stats[8];
#endif

template <bool AUX, bool SEM>
__global__ __launch_bounds__(WAVE) PGR_COMP_OCC void composite_quarter_kernel(const ViewEntry* __restrict__ views,
                                                                              uint32_t items_per_view,
                                                                              const uint32_t* __restrict__ work_order,
                                                                              SemanticDev sem) {
    uint32_t item = work_order ? work_order[blockIdx.x] : blockIdx.x;
    if (item == INVALID_ITEM) return;
    const uint32_t view = item / items_per_view;
    item -= view * items_per_view;
    const ViewEntry& ve = views[view];
    if (ve.counters[1] || !ve.out.color) return;
    const CameraDev& cam = *ve.cam;
    const uint2* __restrict__ ranges = ve.ranges;
    const uint32_t* __restrict__ gauss_sorted = ve.gauss_sorted;
    const float4* __restrict__ splats = ve.splats;
    const CompOut o = ve.out;
    const int W = cam.width, H = cam.height;
    const int tile = (int)(item >> 2), quarter = (int)(item & 3);
    const int tile_x = tile % cam.grid_x, tile_y = tile / cam.grid_x;
    const int lane = threadIdx.x;
    const int qx0 = tile_x * TILE + (quarter & 1) * 8, qy0 = tile_y * TILE + (quarter >> 1) * 8;
    if (qx0 >= W || qy0 >= H) return;            // quarter entirely outside the image
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < W && py < H;
    const f32x2 pxf = {(float)px, (float)px}, pyf = {(float)py, (float)py};

    const uint2 range = ranges[tile];
    const int n = (int)(range.y - range.x);

    // LDS image of a compacted batch, laid out for PAIRS of consecutive entries (2k, 2k+1): the geometry of a pair is
    // evaluated with packed fp32 (one v_pk_* per two entries; v_pk_fma_f32 costs 1.2x a v_fma_f32 on gfx950,
    // scripts/microbench/valu_rates.hip) and arrives from LDS already in packed register pairs:
    //   s_g[3k+0] = (x0, x1, y0, y1)   s_g[3k+1] = (hx0, hx1, ny0, ny1)   s_g[3k+2] = (hz0, hz1, op0, op1)
    //   s_c[j]    = (r, g, b, depth) of entry j: two packed accumulators (Cr,Cg) and (Cb,D)
    constexpr int PAIRS = WAVE_BATCH / 2 + 1;     // +1 pair of null entries pads an odd batch
    __shared__ float4 s_g[3 * PAIRS];
    __shared__ float4 s_c[2 * PAIRS];
    __shared__ uint32_t s_i[2 * PAIRS];
    float* const s_gf = reinterpret_cast<float*>(s_g);

    float T = 1.0f;
    f32x2 Crg = {0.f, 0.f}, Cbd = {0.f, 0.f};
    uint32_t last = 0;
    // pixel state masks live in SGPR pairs: v_cmp writes them, s_and/s_andn2 combine them (scalar unit, beside the
    // VALU stream), and the all-done test is a scalar compare -- no VALU instruction is spent on control
    unsigned long long alive = __builtin_amdgcn_ballot_w64(inside);

    float2 p = make_float2(0.f, 0.f);
    float4 co = make_float4(0.f, 0.f, 0.f, 0.f), cd = make_float4(0.f, 0.f, 0.f, 0.f);
    if (lane < n) {
        const uint32_t g = gauss_sorted[range.x + lane];
        const float4* rec = splats + (size_t)g * 3;
        const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
        p = make_float2(q0.x, q0.y);
        co = make_float4(q0.z, q0.w, q1.x, q1.y);
        cd = make_float4(q1.z, q1.w, q2.x, q2.y);
        if (SEM) {
            const float* col = sem.colors + 3 * (size_t)(sem.object_id[g] - 1);
            cd = make_float4(col[0], col[1], col[2], q2.y);
        }
    }
    const float rx0 = (float)qx0, ry0 = (float)qy0;
    const float rx1 = fminf(rx0 + 7.0f, (float)(W - 1)), ry1 = fminf(ry0 + 7.0f, (float)(H - 1));

#ifdef PGR_COMP_STATS
    unsigned long long st_walk = 0, st_live = 0, st_eval = 0, st_alive = 0, st_blend = 0, st_batches = 0;
#endif
    for (int base = 0; base < n; base += WAVE_BATCH) {
        // Skip + compact (see composite_wave_kernel): only entries that can reach alpha >= 1/255 somewhere in this
        // wave's 8x8 pixels are parked, in list order; an odd batch is padded with one null entry (opacity 0).
        const bool live = base + lane < n && rect_may_contribute(make_cull_splat(p, co), rx0, ry0, rx1, ry1);
        const unsigned long long mask = __ballot(live);
        const int cnt = __popcll(mask);
        const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
#ifdef PGR_COMP_STATS
        st_walk += min(WAVE_BATCH, n - base); st_live += cnt; st_batches++;
#endif
        if (live) {
            float* gq = s_gf + 12 * (pos >> 1) + (pos & 1);
            gq[0] = p.x;  gq[2] = p.y;
            gq[4] = -0.5f * co.x;  gq[6] = -co.y;
            gq[8] = -0.5f * co.z;  gq[10] = co.w;
            s_c[pos] = cd;
            if (AUX) s_i[pos] = (uint32_t)(base + lane + 1);
        }
        if (lane == 0 && (cnt & 1)) {
            float* gq = s_gf + 12 * (cnt >> 1) + 1;
            gq[0] = 0.f; gq[2] = 0.f; gq[4] = 0.f; gq[6] = 0.f; gq[8] = 0.f; gq[10] = 0.f;
            s_c[cnt] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (AUX) s_i[cnt] = 0u;
        }
        __syncthreads();
        // issue the gather of the NEXT batch now; it lands while this batch is composited
        p = make_float2(0.f, 0.f);
        co = make_float4(0.f, 0.f, 0.f, 0.f);
        cd = make_float4(0.f, 0.f, 0.f, 0.f);
        if (base + WAVE_BATCH + lane < n) {
            const uint32_t g = gauss_sorted[range.x + base + WAVE_BATCH + lane];
            const float4* rec = splats + (size_t)g * 3;
            const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
            p = make_float2(q0.x, q0.y);
            co = make_float4(q0.z, q0.w, q1.x, q1.y);
            cd = make_float4(q1.z, q1.w, q2.x, q2.y);
            if (SEM) {
                const float* col = sem.colors + 3 * (size_t)(sem.object_id[g] - 1);
                cd = make_float4(col[0], col[1], col[2], q2.y);
            }
        }
        const int pairs = __builtin_amdgcn_readfirstlane((cnt + 1) >> 1);
        for (int k = 0; k < pairs; ++k) {
            const float4 g0 = s_g[3 * k], g1 = s_g[3 * k + 1], g2 = s_g[3 * k + 2];
            // geometry of entries 2k and 2k+1 side by side; per entry this is the oracle's operation order
            const f32x2 dx = (f32x2){g0.x, g0.y} - pxf;
            const f32x2 dy = (f32x2){g0.z, g0.w} - pyf;
            const f32x2 t1 = (f32x2){g1.z, g1.w} * dy;
            const f32x2 t2 = __builtin_elementwise_fma((f32x2){g1.x, g1.y}, dx, t1);
            const f32x2 t4 = ((f32x2){g2.x, g2.y} * dy) * dy;
            const f32x2 power = __builtin_elementwise_fma(dx, t2, t4);
            const f32x2 p2 = power * (f32x2){1.4426950408889634f, 1.4426950408889634f};
            const f32x2 ex = {__builtin_amdgcn_exp2f(p2.x), __builtin_amdgcn_exp2f(p2.y)};
            const f32x2 araw = (f32x2){g2.z, g2.w} * ex;
            const unsigned long long m_pw[2] = {__builtin_amdgcn_ballot_w64(!(power.x > 0.0f)),
                                                __builtin_amdgcn_ballot_w64(!(power.y > 0.0f))};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float4 c = s_c[2 * k + u];
                const float alpha = fminf(ALPHA_MAX, u ? araw.y : araw.x);
                const float test_T = fmaf(-alpha, T, T);
                const unsigned long long valid = alive & m_pw[u] & __builtin_amdgcn_ballot_w64(!(alpha < ALPHA_MIN));
                const unsigned long long stop = valid & __builtin_amdgcn_ballot_w64(test_T < T_EPS);
                alive &= ~stop;
                const unsigned long long blend = valid & ~stop;
                const bool bl = __builtin_amdgcn_inverse_ballot_w64(blend);
#ifdef PGR_COMP_STATS
                st_eval++; st_alive += __popcll(alive | stop); st_blend += __popcll(blend);
#endif
                // blended: T' = fma(-alpha, T, T) = test_T; not blended: weight 0 (exact no-op) and T unchanged
                const float w = bl ? alpha * T : 0.0f;
                const f32x2 wv = {w, w};
                Crg = __builtin_elementwise_fma((f32x2){c.x, c.y}, wv, Crg);
                Cbd = __builtin_elementwise_fma((f32x2){c.z, c.w}, wv, Cbd);
                T = bl ? test_T : T;
                if (AUX) last = bl ? s_i[2 * k + u] : last;
            }
            if (alive == 0ull) goto finished;
        }
        __syncthreads();
    }
finished:
#ifdef PGR_COMP_STATS
    if (lane == 0 && !SEM) {
        atomicAdd(&g_comp_stats[0], st_walk); atomicAdd(&g_comp_stats[1], st_live); atomicAdd(&g_comp_stats[2], st_eval);
        atomicAdd(&g_comp_stats[3], st_alive); atomicAdd(&g_comp_stats[4], st_blend); atomicAdd(&g_comp_stats[5], 1ull);
        atomicAdd(&g_comp_stats[6], st_batches);
    }
#endif
    if (inside) {
        const size_t P = (size_t)W * H;
        const size_t pix = (size_t)py * W + px;
        o.color[0 * P + pix] = fmaf(T, cam.bg[0], Crg.x);
        o.color[1 * P + pix] = fmaf(T, cam.bg[1], Crg.y);
        o.color[2 * P + pix] = fmaf(T, cam.bg[2], Cbd.x);
        if (o.depth) o.depth[pix] = Cbd.y;
        if (AUX) {
            if (o.final_T) o.final_T[pix] = T;
            if (o.n_contrib) o.n_contrib[pix] = last;
        }
    }
}

#ifndef PGR_COMP_ITEMS
#define PGR_COMP_ITEMS 4         // work items per tile: 2 = half tiles (16x8, 2 px/lane), 4 = quarter tiles (8x8)
#endif
constexpr uint32_t ITEMS_PER_TILE = PGR_COMP_ITEMS;

template <bool AUX, bool SEM>
inline void launch_composite(uint32_t slots, hipStream_t stream, const ViewEntry* views, uint32_t items_per_view,
                             const uint32_t* work_order, SemanticDev sem) {
    if (ITEMS_PER_TILE == 4)
        composite_quarter_kernel<AUX, SEM><<<slots, WAVE, 0, stream>>>(views, items_per_view, work_order, sem);
    else
        composite_wave_kernel<AUX, SEM><<<slots, WAVE, 0, stream>>>(views, items_per_view, work_order, sem);
}

// Work ordering for the wave compositor: half-tile work items sorted by DESCENDING list length
// (256 log-spaced length classes), so the long lists start first and the short ones back-fill the
// SIMDs that finish early (longest-processing-time-first).  Order never affects results.
// ---- work order --------------------------------------------------------------------------------
// The compositor's work items (view, tile, half) are laid out as NUM_XCD interleaved streams: position p
// belongs to stream p % 8, and the hardware dispatcher is observed to place workgroup b on XCD b % 8
// (MI355X_MICROARCH.md, a speed hint only -- any placement is correct).  Stream x owns every 8th band of
// ORDER_BAND_ROWS tile rows of every view, and inside a stream the items keep their row-major order, so a tile's
// two halves and its horizontal neighbours -- whose lists share most of their Gaussians -- run back to back on
// the SAME XCD and find each other's gathers in its L2 (PMC before: 486 MB/view fetched from the fabric for
// 161 MB of algorithmic bytes).  Only a coarse longest-first split is kept (ORDER_CLASSES_USED length classes:
// the long lists of every stream start first, short ones back-fill).  Unused slots hold INVALID_ITEM.
// Contiguous bands per XCD share more vertically but leave XCDs idle when the image's work is uneven
// (measured 1.9x slower on C3); striping keeps every XCD's share statistically equal.
constexpr int ORDER_BAND_ROWS = 1;
constexpr int ORDER_CLASSES_USED = 3;            // > 4096, > 1024, rest
constexpr int ORDER_BINS = NUM_XCD * ORDER_CLASSES_USED;

// order_state layout (uint32): [ORDER_BINS] counters -> cursors, then [1] number of long lists
constexpr int ORDER_STATE_WORDS = ORDER_BINS + 1;

__device__ __forceinline__ int xcd_of_tile(int tile, int grid_x) { return (tile / grid_x / ORDER_BAND_ROWS) % NUM_XCD; }

__device__ __forceinline__ int coarse_class(uint32_t len) { return len > 4096u ? 0 : (len > 1024u ? 1 : 2); }

__host__ __device__ inline int max_band_rows(int grid_y) {
    const int bands = (grid_y + ORDER_BAND_ROWS - 1) / ORDER_BAND_ROWS;
    return (bands + NUM_XCD - 1) / NUM_XCD * ORDER_BAND_ROWS;
}

// grid = (ceil(tiles/256), n_views), 256 threads
__global__ __launch_bounds__(256) void order_count_kernel(const ViewEntry* __restrict__ views, int tiles, int grid_x,
                                                          uint32_t* __restrict__ state) {
    __shared__ uint32_t hist[ORDER_BINS];
    if (threadIdx.x < ORDER_BINS) hist[threadIdx.x] = 0;
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < tiles) {
        const uint2 r = views[blockIdx.y].ranges[t];
        atomicAdd(&hist[xcd_of_tile(t, grid_x) * ORDER_CLASSES_USED + coarse_class(r.y - r.x)], ITEMS_PER_TILE);
    }
    __syncthreads();
    if (threadIdx.x < ORDER_BINS && hist[threadIdx.x]) atomicAdd(&state[threadIdx.x], hist[threadIdx.x]);
}

// 1 thread per stream: exclusive prefix over the classes = write cursors
__global__ void order_scan_kernel(uint32_t* __restrict__ state) {
    const int x = threadIdx.x;
    if (x >= NUM_XCD) return;
    uint32_t acc = 0;
    for (int c = 0; c < ORDER_CLASSES_USED; ++c) {
        const uint32_t v = state[x * ORDER_CLASSES_USED + c];
        state[x * ORDER_CLASSES_USED + c] = acc;
        acc += v;
    }
}

// grid = (ceil(tiles/256), n_views), 256 threads.  work_order is pre-filled with INVALID_ITEM.  Ranks inside a
// workgroup follow the tile order (ballot prefix per bin), so row-major neighbours stay adjacent in their
// stream; one global atomic per non-empty (workgroup, bin) claims the slots.  Tiles whose list exceeds
// `long_threshold` are also appended to long_list for the long sort tiers.
__global__ __launch_bounds__(256) void order_scatter_kernel(const ViewEntry* __restrict__ views, int tiles, int grid_x,
                                                            uint32_t* __restrict__ state,
                                                            uint32_t* __restrict__ work_order, uint32_t long_threshold,
                                                            uint32_t* __restrict__ long_list) {
    __shared__ uint32_t wave_cnt[4][ORDER_BINS];
    __shared__ uint32_t base[ORDER_BINS];
    __shared__ uint32_t n_long_s, long_base_s;
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    if (threadIdx.x == 0) n_long_s = 0;
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    int bin = -1, x = 0;
    uint32_t long_rank = INVALID_ITEM;
    if (t < tiles) {
        const uint2 r = views[blockIdx.y].ranges[t];
        const uint32_t len = r.y - r.x;
        x = xcd_of_tile(t, grid_x);
        bin = x * ORDER_CLASSES_USED + coarse_class(len);
        if (len > long_threshold) long_rank = atomicAdd(&n_long_s, 1u);
    }
    // ordered rank inside the wave, per bin
    uint32_t rank = 0;
    for (int b = 0; b < ORDER_BINS; ++b) {
        const unsigned long long m = __ballot(bin == b);
        if (bin == b) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave][b] = (uint32_t)__popcll(m);
    }
    __syncthreads();
    if (threadIdx.x < ORDER_BINS) {
        const uint32_t tot = wave_cnt[0][threadIdx.x] + wave_cnt[1][threadIdx.x] + wave_cnt[2][threadIdx.x] +
                             wave_cnt[3][threadIdx.x];
        base[threadIdx.x] = tot ? atomicAdd(&state[threadIdx.x], ITEMS_PER_TILE * tot) : 0u;
    }
    if (threadIdx.x == 0 && n_long_s) long_base_s = atomicAdd(&state[ORDER_BINS], n_long_s);
    __syncthreads();
    if (t < tiles) {
        uint32_t before = 0;
        for (int w = 0; w < wave; ++w) before += wave_cnt[w][bin];
        const uint32_t r0 = base[bin] + ITEMS_PER_TILE * (before + rank);     // position inside stream x
        const uint32_t item = ITEMS_PER_TILE * ((uint32_t)blockIdx.y * (uint32_t)tiles + (uint32_t)t);
        for (uint32_t k = 0; k < ITEMS_PER_TILE; ++k) work_order[(size_t)(r0 + k) * NUM_XCD + x] = item + k;
        if (long_rank != INVALID_ITEM) long_list[long_base_s + long_rank] = (uint32_t)blockIdx.y * (uint32_t)tiles + (uint32_t)t;
    }
}

// ---- frame post-processing ---------------------------------------

// masks[k,p] = || img[:,p] - colors[k] ||_2 <= thr   (reference: src/gs/render.py:60-63,89-93)
// grid.y = image index of a contiguous batch: img [B,3,P], masks [B,k,P]
__global__ void color_masks_kernel(const float* __restrict__ img, size_t P, const float* __restrict__ colors, int k,
                                   float thr, uint8_t* __restrict__ masks) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    img += (size_t)blockIdx.y * 3 * P;
    masks += (size_t)blockIdx.y * (size_t)k * P;
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
