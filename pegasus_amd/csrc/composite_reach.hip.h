// composite_reach.hip.h -- EXPERIMENT (round 4), compiled only with -DPGR_REACH_BITS; not part of the product.
//
// "Compact on the index, not on the record": a per-instance byte of four STATIC reach bits (can this instance reach
// alpha >= 1/255 anywhere in quarter q of its tile?) lets a quarter wave ballot over the index words, gather records only for
// reaching entries and form its 64-entry batches from reaching entries only.  This build measures both sides of that trade
// on the raster-only path:
//   reach_bits_kernel            the cost of producing the bits (here: a pass after the sort; 32-B gather + 4 rectangle tests
//                                per listed instance, written to the `alt` buffer, which is dead after the sort)
//   composite_quarter_reach      what the compositor gains: same blend arithmetic and order as composite_quarter's plain
//                                loop (frames are bit-identical), batches formed through an LDS queue of reaching entries
// Result: measured negative.
#pragma once
#include "composite.hip.h"
#include "tilebin.hip.h"

namespace pgr {

// one workgroup per (view, tile) list; reach[p] bit q = instance p may contribute to quarter q of its tile
__global__ __launch_bounds__(256) void reach_bits_kernel(const ViewEntry* __restrict__ views, int tiles,
                                                          const BinView* __restrict__ bins) {
    const int view = blockIdx.y, tile = blockIdx.x;
    const ViewEntry& ve = views[view];
    if (ve.counters[1]) return;
    const CameraDev& cam = *ve.cam;
    const uint2 range = gload(ve.ranges + tile);
    const int n = (int)(range.y - range.x);
    if (n == 0) return;
    const int tile_x = tile % cam.grid_x, tile_y = tile / cam.grid_x;
    uint8_t* reach = reinterpret_cast<uint8_t*>(bins[view].alt);      // (dead after the sort)
    const float W1 = (float)(cam.width - 1), H1 = (float)(cam.height - 1);
    for (int i = threadIdx.x; i < n; i += 256) {
        const uint32_t g = gload(ve.gauss_sorted + range.x + i);
        const float4 q0 = gload(ve.splats + (size_t)g * 3), q1 = gload(ve.splats + (size_t)g * 3 + 1);
        const CullSplat cs = make_cull_splat(make_float2(q0.x, q0.y), make_float4(q0.z, q0.w, q1.x, q1.y), q1.z, q1.w);
        uint32_t bits = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float x0 = (float)(tile_x * TILE + (q & 1) * 8), y0 = (float)(tile_y * TILE + (q >> 1) * 8);
            if (x0 > W1 || y0 > H1) continue;
            if (rect_may_contribute(cs, x0, y0, fminf(x0 + 7.0f, W1), fminf(y0 + 7.0f, H1))) bits |= 1u << q;
        }
        gstore(reach + range.x + i, (uint8_t)bits);
    }
}

template <bool AUX>
__device__ __forceinline__ void composite_quarter_reach(const ViewEntry& ve, uint32_t item, const uint8_t* __restrict__ reach,
                                                        float4* __restrict__ s_g, float4* __restrict__ s_c,
                                                        uint32_t* __restrict__ s_i, uint32_t* __restrict__ s_q) {
    const CameraDev& cam = *ve.cam;
    const uint32_t* __restrict__ gauss_sorted = ve.gauss_sorted;
    const float4* __restrict__ splats = ve.splats;
    const CompOut o = ve.out;
    const int W = cam.width, H = cam.height;
    const int tile = (int)(item >> 2), quarter = (int)(item & 3);
    const int tile_x = tile % cam.grid_x, tile_y = tile / cam.grid_x;
    const int lane = threadIdx.x;
    const int qx0 = tile_x * TILE + (quarter & 1) * 8, qy0 = tile_y * TILE + (quarter >> 1) * 8;
    if (qx0 >= W || qy0 >= H) return;
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < W && py < H;
    uint32_t pix32 = (uint32_t)py * (uint32_t)W + (uint32_t)px;
    asm volatile("" : "+v"(pix32));
    const f32x2 pxf = {(float)px, (float)px}, pyf = {(float)py, (float)py};
    const uint2 range = gload(ve.ranges + tile);
    const int n = (int)(range.y - range.x);
    float* const s_gf = reinterpret_cast<float*>(s_g);

    float T = 1.0f;
    f32x2 Crg = {0.f, 0.f}, Cbd = {0.f, 0.f};
    uint32_t last = 0;
    unsigned long long alive = __builtin_amdgcn_ballot_w64(inside);
    const float rx0 = (float)qx0, ry0 = (float)qy0;
    const float rx1 = fminf(rx0 + 7.0f, (float)(W - 1)), ry1 = fminf(ry0 + 7.0f, (float)(H - 1));

    // ---- the queue of reaching entries: s_q[2 j] = Gaussian index, s_q[2 j + 1] = 1-based list position
    int scan = 0, qn = 0;                       // next list position to scan; queued entries (both wave-uniform)
    auto load_words = [&](int base, uint32_t& w, uint32_t& rb) {
        const int i = base + lane;
        w = i < n ? gload(gauss_sorted + range.x + i) : 0u;
        rb = i < n ? (uint32_t)gload(reach + range.x + i) : 0u;
    };
    uint32_t w_next, r_next;
    load_words(0, w_next, r_next);
    f32x4_t q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0, q2 = q0;
    uint32_t posn = 0;
    bool have = false;
    // fills the queue to >= 64 entries (or the end of the list), takes the first 64 into registers and requests their records
    auto prepare = [&]() {
        while (qn < WAVE_BATCH && scan < n) {
            const bool r = (scan + lane < n) && ((r_next >> quarter) & 1u);
            const unsigned long long m = __ballot(r);
            const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (r) { s_q[2 * (qn + pos)] = w_next; s_q[2 * (qn + pos) + 1] = (uint32_t)(scan + lane + 1); }
            qn += __popcll(m);
            scan += WAVE_BATCH;
            load_words(scan, w_next, r_next);
        }
        asm volatile("" ::: "memory");
        const int take = min(qn, WAVE_BATCH);
        have = lane < take;
        uint32_t g = 0;
        posn = 0;
        if (have) { g = s_q[2 * lane]; posn = s_q[2 * lane + 1]; }
        // the rest of the queue moves to the front (at most 63 entries: a fill round adds at most 64 to fewer than 64)
        uint32_t mg = 0, mp = 0;
        const bool mv = lane < qn - take;
        if (mv) { mg = s_q[2 * (take + lane)]; mp = s_q[2 * (take + lane) + 1]; }
        asm volatile("" ::: "memory");
        if (mv) { s_q[2 * lane] = mg; s_q[2 * lane + 1] = mp; }
        qn -= take;
        if (have) {
            const float4* rec = splats + (size_t)g * 3;
            q0 = gload_quad(rec); q1 = gload_quad(rec + 1); q2 = gload_quad(rec + 2);
        }
        return take;
    };
    int take = prepare();
    while (take > 0 && alive != 0ull) {
        const unsigned long long any = alive;
        const int ay0 = __builtin_ctzll(any) >> 3, ay1 = (63 - __builtin_clzll(any)) >> 3;
        unsigned int cols = (unsigned int)(any | (any >> 32));
        cols |= cols >> 16; cols |= cols >> 8; cols &= 0xffu;
        const int ax0 = __builtin_ctz(cols), ax1 = 31 - __builtin_clz(cols);
        const float bx0 = rx0 + (float)ax0, by0 = ry0 + (float)ay0;
        const float bx1 = fminf(rx0 + (float)ax1, rx1), by1 = fminf(ry0 + (float)ay1, ry1);
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2));
        const float2 p = make_float2(q0.x, q0.y);
        const float4 co = make_float4(q0.z, q0.w, q1.x, q1.y);
        const bool live = have && rect_may_contribute(make_cull_splat(p, co, q1.z, q1.w), bx0, by0, bx1, by1);
        const unsigned long long mask = __ballot(live);
        const int cnt = __popcll(mask);
        const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (live) {
            float* gq = s_gf + 12 * (pos >> 1) + (pos & 1);
            gq[0] = p.x;  gq[2] = p.y;
            gq[4] = -0.5f * co.x;  gq[6] = -co.y;
            gq[8] = -0.5f * co.z;  gq[10] = co.w;
            s_c[pos] = make_float4(q2.x, q2.y, q2.z, q2.w);
            if (AUX) s_i[pos] = posn;
        }
        if (lane == 0 && (cnt & 1)) {
            float* gq = s_gf + 12 * (cnt >> 1) + 1;
            gq[0] = 0.f; gq[2] = 0.f; gq[4] = 0.f; gq[6] = 0.f; gq[8] = 0.f; gq[10] = 0.f;
            s_c[cnt] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (AUX) s_i[cnt] = 0u;
        }
        __syncthreads();
        take = prepare();                          // the next batch's records land while this one is composited
        const int pairs = __builtin_amdgcn_readfirstlane((cnt + 1) >> 1);
        bool all_done = false;
        uint32_t og = 0, oc = 0;
        asm volatile("" : "+v"(og), "+v"(oc));
        for (int k0 = 0; k0 < pairs && !all_done; k0 += PAIR_UNROLL, og += PAIR_UNROLL * 48u, oc += PAIR_UNROLL * 32u) {
#pragma unroll
            for (int ku = 0; ku < PAIR_UNROLL; ++ku) {
                const int k = k0 + ku;
                if (ku > 0 && k >= pairs) break;
                const char* pg = reinterpret_cast<const char*>(s_g) + og + ku * 48;
                const float4 g0 = *reinterpret_cast<const float4*>(pg), g1 = *reinterpret_cast<const float4*>(pg + 16),
                             g2 = *reinterpret_cast<const float4*>(pg + 32);
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
                    float alpha;
                    asm("v_min_f32 %0, %1, %2" : "=v"(alpha) : "s"(ALPHA_MAX), "v"(u ? araw.y : araw.x));
                    const unsigned long long hit = m_pw[u] & __builtin_amdgcn_ballot_w64(!(alpha < ALPHA_MIN));
                    if (const unsigned long long valid = alive & hit; valid != 0ull) {
                        const float4 c = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_c) + oc + (2 * ku + u) * 16);
                        const float test_T = fmaf(-alpha, T, T);
                        const unsigned long long stop = valid & __builtin_amdgcn_ballot_w64(test_T < T_EPS);
                        alive &= ~stop;
                        const bool bl = __builtin_amdgcn_inverse_ballot_w64(valid & ~stop);
                        const float w = bl ? alpha * T : 0.0f;
                        const f32x2 wv = {w, w};
                        Crg = __builtin_elementwise_fma((f32x2){c.x, c.y}, wv, Crg);
                        Cbd = __builtin_elementwise_fma((f32x2){c.z, c.w}, wv, Cbd);
                        T = bl ? test_T : T;
                        if (AUX) last = bl ? s_i[2 * k + u] : last;
                    }
                }
                if (alive == 0ull) { all_done = true; break; }
            }
        }
        if (all_done) break;
        __syncthreads();
    }
    if (inside) {
        const size_t P = (size_t)W * H, pix = pix32;
        gstore(o.color + 0 * P + pix, fmaf(T, cam.bg[0], Crg.x));
        gstore(o.color + 1 * P + pix, fmaf(T, cam.bg[1], Crg.y));
        gstore(o.color + 2 * P + pix, fmaf(T, cam.bg[2], Cbd.x));
        if (o.depth) gstore(o.depth + pix, Cbd.y);
        if (AUX) {
            if (o.final_T) gstore(o.final_T + pix, T);
            if (o.n_contrib) gstore(o.n_contrib + pix, last);
        }
    }
}

template <bool AUX>
__global__ __launch_bounds__(WAVE) void composite_reach_kernel(const ViewEntry* __restrict__ views, uint32_t items_per_view,
                                                               const uint32_t* __restrict__ work_order,
                                                               const BinView* __restrict__ bins) {
    uint32_t item = work_order[blockIdx.x];
    if (item == INVALID_ITEM) return;
    const uint32_t view = item / items_per_view;
    item -= view * items_per_view;
    const ViewEntry& ve = views[view];
    if (ve.counters[1] || !ve.out.color) return;
    constexpr int PAIRS = WAVE_BATCH / 2 + 1;
    __shared__ float4 s_g[3 * PAIRS];
    __shared__ float4 s_c[2 * PAIRS];
    __shared__ uint32_t s_i[AUX ? 2 * PAIRS : 1];
    __shared__ uint32_t s_q[2 * 2 * WAVE_BATCH];
    composite_quarter_reach<AUX>(ve, item, reinterpret_cast<const uint8_t*>(bins[view].alt), s_g, s_c, s_i, s_q);
}

}  // namespace pgr
