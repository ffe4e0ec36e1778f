// composite.hip.h -- front-to-back alpha compositing, one wave per 8x8-pixel quarter tile.
// SURVEY.md section 8a row a10.  Same per-pixel operation order as oracle/pgr_oracle.c composite_tile();
// the only non-bit-exact step is exp (v_exp_f32 here, glibc expf in the oracle).
//
// Bound: VALU + transcendental issue (not HBM, not MFMA): measured on MI355X (scripts/microbench/valu_rates.hip)
// a wave64 v_mul/v_mov issues in ~2.6 cycles, v_fma_f32 3.9, v_pk_fma_f32 4.7 (two FMAs), v_exp_f32 8.1 and a
// v_cmp 4-6; the inner loop below is ~75 such cycles per (wave, list entry) and SQ counters show the VALU >90 % busy.
#pragma once
#include <type_traits>

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
    const float4* splats;        // [n, 3] records: q0 = (x,y,A,B), q1 = (C,op,B/C,B/A), q2 = (r,g,b,depth)
    CompOut out;
    const uint32_t* counters;    // [1] != 0: instance overflow, the view must not be composited
    float* sem_color;            // fused semantic pass: [3,H,W] objects-only image in semantic colours (or NULL)
    float* sem_depth;            // [H,W] objects-only depth (or NULL)
    const uint32_t* obj_last;    // [tiles] 1 + position of the last object entry of each sorted list (tile sort)
    uint8_t* sem_masks;          // [K,H,W] colour-distance masks of the semantic image / (layered call) of the layers, or NULL
    uint8_t* record;             // the view's frame record (rgb u8 | depth u16 mm | mask bit planes), or NULL
};
static_assert(sizeof(ViewEntry) == 112, "ViewEntry layout");

// Fused semantic pass (same for every view of a batch): which Gaussians are objects and what colour they carry.
struct SemanticDev {
    const int32_t* object_id;    // [n] 0 = environment, k = object k
    const uint8_t* object_u8;    // [n - n_env] the object ids of the object Gaussians as bytes (batch header, written per
                                 // batch by pack_object_ids_kernel when k <= 255), or NULL: the fused walk looks an id up
                                 // per object entry, and a 4-byte gather from the 8 MB int32 array pulled a whole line from
                                 // HBM for it (PMC, round 3: +41 MB fetched per view); the byte table of a 2 M-Gaussian
                                 // scene is 0.64 MB and stays in the L2s
    const float* colors;         // [K,3] the rgb value object k's Gaussians carry: max(C0*RGB2SH(c_k) + 0.5, 0)
    int32_t n_env;               // Gaussians with index < n_env are environment
    int32_t k;
    const float* mask_colors;    // [K,3] colours the masks are thresholded against (NULL: no masks)
    float mask_thr;
    int32_t layer_tiles;         // LAYERED kernel: tiles of one layer (work item's tile index = layer * layer_tiles + tile)
};

// One pixel against colours [c0, c1): masks[c, pix] = || (r,g,b) - colors[c] ||_2 <= thr with the arithmetic of
// color_masks_kernel below, bit for bit (same expression tree; the wave-wide shortcut only skips work whose result is
// known).  Every lane of the wave calls it; `inside` lanes store.  masks == NULL: no planes.  rec_masks != NULL: the same
// verdicts also go into the frame record's bit planes (byte j of a pixel = masks 8j .. 8j+7; c0 must be 0 then).
__device__ __forceinline__ void pixel_masks(float r, float g, float b, const float* __restrict__ colors, int c0, int c1,
                                            float thr, uint8_t* __restrict__ masks, size_t P, size_t pix, bool inside,
                                            uint8_t* __restrict__ rec_masks = nullptr) {
    const float far = thr * 1.000001f;
    const int bytes = (c1 + 7) >> 3;
    uint32_t bits = 0;
    for (int c = c0; c < c1; ++c) {
        const float d0 = r - colors[3 * c], d1 = g - colors[3 * c + 1], d2 = b - colors[3 * c + 2];
        const bool surely_out = fabsf(d0) > far || fabsf(d1) > far || fabsf(d2) > far;
        uint8_t m = 0;
        if (__ballot(!surely_out) != 0ull) {
            const float dist = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
            m = dist <= thr ? 1 : 0;
        }
        if (masks && inside) gstore(masks + (size_t)c * P + pix, m);
        if (rec_masks) {
            bits |= (uint32_t)m << (c & 7);
            if ((c & 7) == 7 || c == c1 - 1) {
                if (inside) gstore(rec_masks + pix * (size_t)bytes + (size_t)(c >> 3), (uint8_t)bits);
                bits = 0;
            }
        }
    }
}

// the two casts of a frame record (pack_records_kernel's, bit for bit): uint8(v * 255) wrapping, uint16(d * 1000) wrapping
__device__ __forceinline__ uint8_t quant_u8(float v) {
    v = fminf(fmaxf(v * 255.0f, -2147483520.0f), 2147483520.0f);
    return (uint8_t)((int)v & 0xFF);
}
__device__ __forceinline__ uint16_t quant_mm(float d) {
    d = fminf(fmaxf(d * 1000.0f, -2147483520.0f), 2147483520.0f);
    return (uint16_t)((int)d & 0xFFFF);
}
__device__ __forceinline__ size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int HALF_ROWS = 8;     // rows per half tile (backward.hip.h's work unit)
constexpr int WAVE_BATCH = 64;   // list entries staged per round
constexpr uint32_t ITEMS_PER_TILE = 4;   // work items per 16x16 tile: its four 8x8 quarters
constexpr int SEM_LDS_OBJECTS = 64;      // semantic colours kept in LDS by the fused walk (more objects: read from memory)
#ifndef PGR_PAIR_UNROLL
#define PGR_PAIR_UNROLL 4
#endif
constexpr int PAIR_UNROLL = PGR_PAIR_UNROLL;   // entry pairs per trip of the compositing loop

#ifndef PGR_COMP_WAVES
#define PGR_COMP_WAVES 0         // 0: let the compiler pick; k: cap VGPRs for k waves/SIMD (tuning builds)
#endif
#if PGR_COMP_WAVES
#define PGR_COMP_OCC __attribute__((amdgpu_waves_per_eu(PGR_COMP_WAVES, PGR_COMP_WAVES)))
#else
#define PGR_COMP_OCC
#endif


// ---------------------------------------------------------------------------------------------
// composite_quarter_kernel: ONE WAVE per quarter tile (8 x 8 pixels), one pixel per lane; ONE launch covers every
// (view, tile, quarter) of a batch.  The workgroup is a single wave: no cross-wave barrier, early-out per quarter.
//
//  * The tile's sorted list is consumed in batches of 64: lane j gathers entry j's 48-B record into registers one
//    batch AHEAD (the loads land while the current batch is composited).
//  * Skip + compact: lane j decides whether ITS entry can reach alpha >= 1/255 anywhere in this wave's 8x8 pixels
//    (the binning's conservative predicate, on the quarter).  Entries that cannot are no-ops for every lane, so only
//    the live ones are parked in LDS, compacted in list order (ballot + mbcnt).
//  * The parked batch is laid out for PAIRS of consecutive entries: the geometry of a pair (dx, dy, conic form, exp,
//    opacity) is evaluated with packed fp32 -- one v_pk_* per two entries -- and arrives from LDS already as packed
//    register pairs; the two colour accumulators are packed too ((r,g) and (b,depth)).  Per entry and pixel the
//    operations and their order are the oracle's.
//  * Pixel state masks (alive / valid / blended) live in SGPR pairs: v_cmp writes them, the scalar unit combines them
//    beside the VALU stream, the all-done test is a scalar compare.  Non-blended lanes run the blend with weight 0
//    (exact no-op: fma(c, 0, C) == C) and keep T through a v_cndmask.
//
// FUSED = true adds PEGASUS's object-only semantic render (/root/reference/src/gs/render.py:68-97: all objects in
// their semantic colours, environment REMOVED) to the same walk.  An objects-only cloud's per-tile list is the
// scene's list minus the environment entries, and an entry's alpha does not depend on what else is in the list, so
// the semantic image is a second (T, colour, depth) accumulator per pixel that is advanced ONLY by object entries --
// with the alpha already computed for the scene image.  After the scene pixels of the wave are saturated the walk
// continues over object entries only (environment entries are dropped before their record is even gathered) up to
// the tile's last object entry (ViewEntry::obj_last, a by-product of the tile sort).  Pixel arithmetic is the
// sequence a separate objects-only pass executes: bit-identical (tests/test_gpu_parity.py).
// LAYERED (pgr_forward_layers_async): the work item's tile index runs over n_layers stacked copies of the tile grid;
// pixels and lists are those of the layer rendered alone, and the only output is the layer's colour-distance mask.
template <bool AUX, bool FUSED, bool LAYERED = false>
__device__ __forceinline__ void composite_quarter(const ViewEntry& ve, uint32_t item, const SemanticDev& sem, int n_sem,
                                                  bool sem_background, float4* __restrict__ s_g, float4* __restrict__ s_c,
                                                  float4* __restrict__ s_s, uint32_t* __restrict__ s_i,
                                                  const float* __restrict__ s_col) {
    const CameraDev& cam = *ve.cam;
    const uint32_t* __restrict__ gauss_sorted = ve.gauss_sorted;
    const float4* __restrict__ splats = ve.splats;
    const CompOut o = ve.out;
    const int W = cam.width, H = cam.height;
    const int list = (int)(item >> 2), quarter = (int)(item & 3);     // list = tile, or layer * layer_tiles + tile
    const int layer = LAYERED ? list / sem.layer_tiles : 0;
    const int tile = LAYERED ? list - layer * sem.layer_tiles : list;
    const int tile_x = tile % cam.grid_x, tile_y = tile / cam.grid_x;
    const int lane = threadIdx.x;
    const int qx0 = tile_x * TILE + (quarter & 1) * 8, qy0 = tile_y * TILE + (quarter >> 1) * 8;
    if (qx0 >= W || qy0 >= H) return;            // quarter entirely outside the image
    const int px = qx0 + (lane & 7), py = qy0 + (lane >> 3);
    const bool inside = px < W && py < H;
    // the pixel's index, formed ONCE (pinned: the epilogue's stores would otherwise each rebuild it from the lane's row and
    // column, and keep those live through the walk -- 78 instead of 72 VGPRs in the fused kernel, 6 instead of 7 waves)
    uint32_t pix32 = (uint32_t)py * (uint32_t)W + (uint32_t)px;
    asm volatile("" : "+v"(pix32));
    const f32x2 pxf = {(float)px, (float)px}, pyf = {(float)py, (float)py};

    const uint2 range = gload(ve.ranges + list);
    const int n = (int)(range.y - range.x);
    const bool want_sem = FUSED;                  // (the kernel sends only quarters with object entries down this path)

    // LDS image of a compacted batch, pair-major (arrays of the kernel):
    //   s_g[3k+0] = (x0, x1, y0, y1)   s_g[3k+1] = (hx0, hx1, ny0, ny1)   s_g[3k+2] = (hz0, hz1, op0, op1)
    //   s_c[j] = (r, g, b, depth) of entry j      s_s[j] = (sem r, sem g, sem b, depth) for object entries, 0 otherwise
    //   s_i[j] = 1-based list position (n_contrib bookkeeping)
    float* const s_gf = reinterpret_cast<float*>(s_g);

    float T = 1.0f, Ts = 1.0f;
    f32x2 Crg = {0.f, 0.f}, Cbd = {0.f, 0.f}, Srg = {0.f, 0.f}, Sbd = {0.f, 0.f};
    uint32_t last = 0;
    unsigned long long alive = __builtin_amdgcn_ballot_w64(inside);
    unsigned long long sem_alive = n_sem > 0 ? alive : 0ull;
    // `pure` (wave-uniform): no environment entry has touched this quarter's pixels yet.  Until one does, both images
    // have blended exactly the same (object) entries from the same start: Ts == T and sem_alive == alive BITWISE, and an
    // object entry's weight alpha * Ts is the scene blend's alpha * T.  Its semantic blend is then two more packed FMAs
    // with that weight instead of a second transmittance test, stop mask, weight and update.  The interior quarters of
    // an object seen from the camera stay in this state until they saturate: 92 % of C3's semantic entries (where the
    // kernel time does not notice), ALL of them in an object-only scene such as C2, whose fused compositor it takes from
    // 1.23 to 0.9 ms per 32 views.
    bool pure = FUSED && n_sem > 0;

    // Register-staged gather, software-pipelined over the batches: this lane's record for the CURRENT batch was requested
    // during the previous batch's pair loop, and the list index of the NEXT batch's entry right behind it -- an index ->
    // record chain issued in one go costs the wave two exposed memory round trips per batch (the record's address needs
    // the index).  The record's quads are used as they were loaded: q2 = (r, g, b, depth) IS the colour image of the
    // entry, so nothing has to be shuffled into place behind the loads (a shuffle there drags the s_waitcnt with it,
    // in front of the pair loop: the first layout, q1 = (C, op, r, g), q2 = (b, depth, ..), did exactly that).
    f32x4_t q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0, q2 = q0;
    int32_t oid = 0;      // FUSED: the entry's object id (0 = not an object entry of interest)
    bool have = false;    // the record was gathered (entries nobody needs any more are not)
    auto fetch_index = [&](int base) {
        const int i = base + lane;
        return i < n ? gload(gauss_sorted + range.x + i) : 0u;
    };
    auto fetch_record = [&](int base, uint32_t g) {
        have = false;
        oid = 0;
        const int i = base + lane;
        if (i < n) {
            const bool is_obj = FUSED && i < n_sem && (int)g >= sem.n_env;
            if (alive != 0ull || is_obj) {
                const float4* rec = splats + (size_t)g * 3;
                q0 = gload_quad(rec); q1 = gload_quad(rec + 1); q2 = gload_quad(rec + 2);
                have = true;
                if (is_obj) oid = sem.object_u8 ? (int32_t)gload(sem.object_u8 + (g - (uint32_t)sem.n_env)) : gload(sem.object_id + g);
            }
        }
    };
    fetch_record(0, fetch_index(0));
    uint32_t g_next = fetch_index(WAVE_BATCH);
    const float rx0 = (float)qx0, ry0 = (float)qy0;
    const float rx1 = fminf(rx0 + 7.0f, (float)(W - 1)), ry1 = fminf(ry0 + 7.0f, (float)(H - 1));

    for (int base = 0; base < n; base += WAVE_BATCH) {
        if (alive == 0ull && (sem_alive == 0ull || base >= n_sem)) break;
        // The skip test only has to cover pixels that can still change: the bounding box of the alive lanes (lane =
        // 8 y + x), which shrinks as the quarter saturates (scalar bit tricks on the SGPR masks).
        const unsigned long long any = alive | sem_alive;
        const int ay0 = __builtin_ctzll(any) >> 3, ay1 = (63 - __builtin_clzll(any)) >> 3;
        unsigned int cols = (unsigned int)(any | (any >> 32));
        cols |= cols >> 16; cols |= cols >> 8; cols &= 0xffu;
        const int ax0 = __builtin_ctz(cols), ax1 = 31 - __builtin_clz(cols);
        const float bx0 = rx0 + (float)ax0, by0 = ry0 + (float)ay0;
        const float bx1 = fminf(rx0 + (float)ax1, rx1), by1 = fminf(ry0 + (float)ay1, ry1);
        // with the scene pixels saturated only object entries are still of interest
        // the quads are taken as they arrive, HERE: without this the compiler forms the packed operands of the staging
        // arithmetic right behind the loads, in the previous iteration, and waits for the loads there
        asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(oid));
        const float2 p = make_float2(q0.x, q0.y);
        const float4 co = make_float4(q0.z, q0.w, q1.x, q1.y);
        float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
        if (FUSED && oid > 0) {
            // the colour table sits in LDS (s_col, filled once per wave): fetched from memory here, the three loads were a
            // round trip per batch that the next batch's record requests had to queue behind (s_waitcnt vmcnt(0) in the ISA)
            if (s_col) {
                const float* col = s_col + 3 * (oid - 1);
                cs = make_float4(col[0], col[1], col[2], q2.w);                          // .w = depth (> 0.2: doubles as "object entry")
            } else {
                const float* col = sem.colors + 3 * (size_t)(oid - 1);
                cs = make_float4(gload(col), gload(col + 1), gload(col + 2), q2.w);
            }
        }
        const bool live = have && (alive != 0ull || cs.w != 0.0f) &&
                          rect_may_contribute(make_cull_splat(p, co, q1.z, q1.w), bx0, by0, bx1, by1);
        const unsigned long long mask = __ballot(live);
        const int cnt = __popcll(mask);
        const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (live) {
            float* gq = s_gf + 12 * (pos >> 1) + (pos & 1);
            gq[0] = p.x;  gq[2] = p.y;
            gq[4] = -0.5f * co.x;  gq[6] = -co.y;
            gq[8] = -0.5f * co.z;  gq[10] = co.w;
            s_c[pos] = make_float4(q2.x, q2.y, q2.z, q2.w);
            if (FUSED) s_s[pos] = cs;
            if (AUX) s_i[pos] = (uint32_t)(base + lane + 1);
        }
        if (lane == 0 && (cnt & 1)) {             // null entry (opacity 0: never valid) completes the last pair
            float* gq = s_gf + 12 * (cnt >> 1) + 1;
            gq[0] = 0.f; gq[2] = 0.f; gq[4] = 0.f; gq[6] = 0.f; gq[8] = 0.f; gq[10] = 0.f;
            s_c[cnt] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (FUSED) s_s[cnt] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (AUX) s_i[cnt] = 0u;
        }
        __syncthreads();
        // which of the parked entries are objects': one bit per compacted position, kept in an SGPR pair
        unsigned long long objbits = 0ull;
        if (FUSED && sem_alive != 0ull) objbits = __builtin_amdgcn_ballot_w64(lane < cnt && s_s[lane].w != 0.0f);
        fetch_record(base + WAVE_BATCH, g_next);  // lands while this batch is composited (its index: requested a batch ago)
        g_next = fetch_index(base + 2 * WAVE_BATCH);
        const int pairs = __builtin_amdgcn_readfirstlane((cnt + 1) >> 1);
        // PAIR_UNROLL pairs per trip.  The LDS byte offsets of the trip live in VGPRs the compiler cannot see through
        // (it would otherwise keep them on the scalar unit and pay a v_mov per ds_read: 3-4 of the ~36 VALU instructions
        // of a pair); inside a trip every address is one of them + an immediate.
        // The fused kernel holds the pair loop three times and picks one per batch (scalar):
        //   RIDE     every parked entry is an object's and no environment entry has blended yet (`pure`): the semantic image
        //            advances with the scene image -- same weight, same stop mask -- so the loop is the plain loop plus two
        //            packed FMAs per valid entry, with NO per-entry bookkeeping; Ts and sem_alive are set once, behind it;
        //   PLAIN    no object entry parked, or the semantic pixels all saturated: nothing to do for the semantic image
        //            (a batch of environment entries ends the object-only state up front: conservative, and `pure` is only
        //            ever a licence for a shortcut whose result is bitwise the general path's);
        //   GENERAL  mixed batches: object bit, `pure`, second set of masks per entry.
        // The general loop is bound by SCALAR issue (226 scalar + 41 branch instructions per 8-entry trip against 214 vector
        // ones; SQ counters: 0.79 of the VALU issue slots where the plain loop reaches 0.91), so the batches that do not
        // need its bookkeeping do not run it.
        const unsigned long long parked_bits = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1ull);
        constexpr int MODE_PLAIN = 0, MODE_RIDE = 1, MODE_GENERAL = 2;
        int mode = MODE_PLAIN;
        if (FUSED && objbits != 0ull) mode = (pure && objbits == parked_bits) ? MODE_RIDE : MODE_GENERAL;
        if (FUSED && mode == MODE_PLAIN && cnt > 0) pure = false;
        bool all_done = false;
        auto pair_loop = [&](auto mode_tag) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool SEM = MODE == MODE_GENERAL;
        uint32_t og = 0, oc = 0;
        asm volatile("" : "+v"(og), "+v"(oc));
        for (int k0 = 0; k0 < pairs; k0 += PAIR_UNROLL, og += PAIR_UNROLL * 48u, oc += PAIR_UNROLL * 32u) {
#pragma unroll
            for (int ku = 0; ku < PAIR_UNROLL; ++ku) {
            const int k = k0 + ku;
            if (ku > 0 && k >= pairs) break;
            const char* pg = reinterpret_cast<const char*>(s_g) + og + ku * 48;
            const float4 g0 = *reinterpret_cast<const float4*>(pg), g1 = *reinterpret_cast<const float4*>(pg + 16),
                         g2 = *reinterpret_cast<const float4*>(pg + 32);
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
                // min(0.99, .) as the bare instruction: fminf() on the unpacked high half costs a canonicalising v_max first
                // (the product is already canonical); v_min_f32 returns the other operand for a NaN, like fminf
                float alpha;
                asm("v_min_f32 %0, %1, %2" : "=v"(alpha) : "s"(ALPHA_MAX), "v"(u ? araw.y : araw.x));
                // power <= 0 and alpha >= 1/255: the entry counts for this pixel (in either image)
                const unsigned long long hit = m_pw[u] & __builtin_amdgcn_ballot_w64(!(alpha < ALPHA_MIN));
                // (scalar branch) about a fifth of the parked entries reach no pixel that is still alive
                const bool obj_entry = SEM && ((objbits >> (2 * k + u)) & 1ull);
                if (const unsigned long long valid = alive & hit; valid != 0ull) {
                    if (SEM && !obj_entry) pure = false;     // an environment entry: the two images part ways here
                    const float4 c = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_c) + oc + (2 * ku + u) * 16);
                    const float test_T = fmaf(-alpha, T, T);
                    const unsigned long long stop = valid & __builtin_amdgcn_ballot_w64(test_T < T_EPS);
                    alive &= ~stop;
                    const unsigned long long blend = valid & ~stop;
                    const bool bl = __builtin_amdgcn_inverse_ballot_w64(blend);
                    // blended: T' = fma(-alpha, T, T) = test_T; not blended: weight 0 (exact no-op) and T unchanged
                    const float w = bl ? alpha * T : 0.0f;
                    const f32x2 wv = {w, w};
                    Crg = __builtin_elementwise_fma((f32x2){c.x, c.y}, wv, Crg);
                    Cbd = __builtin_elementwise_fma((f32x2){c.z, c.w}, wv, Cbd);
                    T = bl ? test_T : T;
                    if (AUX) last = bl ? s_i[2 * k + u] : last;
                    if (MODE == MODE_RIDE) {                 // every entry of this batch rides: no test, state set behind the loop
                        const float4 sc = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_s) + oc + (2 * ku + u) * 16);
                        Srg = __builtin_elementwise_fma((f32x2){sc.x, sc.y}, wv, Srg);
                        Sbd = __builtin_elementwise_fma((f32x2){sc.z, sc.w}, wv, Sbd);
                    }
                    if (SEM && pure && obj_entry) {          // same weight, same stop mask: two FMAs are the whole blend
                        const float4 sc = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_s) + oc + (2 * ku + u) * 16);
                        Srg = __builtin_elementwise_fma((f32x2){sc.x, sc.y}, wv, Srg);
                        Sbd = __builtin_elementwise_fma((f32x2){sc.z, sc.w}, wv, Sbd);
                        Ts = T;
                        sem_alive = alive;
                    }
                }
                if (SEM && !pure) {
                    // wave-uniform (scalar) test: is this entry an object's?
                    if (const unsigned long long valid = sem_alive & hit; obj_entry && valid != 0ull) {
                        const float4 sc = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(s_s) + oc + (2 * ku + u) * 16);
                        const float depth = sc.w;
                        const float test_T = fmaf(-alpha, Ts, Ts);
                        const unsigned long long stop = valid & __builtin_amdgcn_ballot_w64(test_T < T_EPS);
                        sem_alive &= ~stop;
                        const bool bl = __builtin_amdgcn_inverse_ballot_w64(valid & ~stop);
                        const float w = bl ? alpha * Ts : 0.0f;
                        const f32x2 wv = {w, w};
                        Srg = __builtin_elementwise_fma((f32x2){sc.x, sc.y}, wv, Srg);
                        Sbd = __builtin_elementwise_fma((f32x2){sc.z, depth}, wv, Sbd);
                        Ts = bl ? test_T : Ts;
                    }
                }
            }
            if (MODE == MODE_RIDE ? alive == 0ull : (alive | sem_alive) == 0ull) { all_done = true; return; }
            }
        }
        };
        if (mode == MODE_GENERAL) pair_loop(std::integral_constant<int, MODE_GENERAL>{});
        else if (mode == MODE_RIDE) {
            pair_loop(std::integral_constant<int, MODE_RIDE>{});
            Ts = T;                     // the object-only state: the semantic image has blended exactly what the scene image has
            sem_alive = alive;
        } else pair_loop(std::integral_constant<int, MODE_PLAIN>{});
        if (all_done) break;
        __syncthreads();
    }
    const size_t P = (size_t)W * H;
    const size_t pix = pix32;
    // depth rule (PgrDepthMode): the un-normalised sum, or the sum over 1 - T_final (= the blended weights' total)
    const bool norm_depth = cam.depth_mode == PGR_DEPTH_NORMALIZED;
    auto depth_out = [&](float D, float Tf) {
        if (!norm_depth) return D;
        const float wsum = 1.0f - Tf;
        return wsum > 0.0f ? D / wsum : 0.0f;
    };
    if (LAYERED) {
        pixel_masks(fmaf(T, cam.bg[0], Crg.x), fmaf(T, cam.bg[1], Crg.y), fmaf(T, cam.bg[2], Cbd.x), sem.mask_colors, layer,
                    layer + 1, sem.mask_thr, ve.sem_masks, P, pix, inside);
        return;
    }
    // the frame record: what leaves the GPU for this view, from the registers that hold the pixel (no pass re-reads the images)
    uint8_t* const rec = ve.record;
    const size_t rec_depth = align16(3 * P), rec_masks = rec_depth + align16(2 * P);
    if (inside) {
        const float cr = fmaf(T, cam.bg[0], Crg.x), cg = fmaf(T, cam.bg[1], Crg.y), cb = fmaf(T, cam.bg[2], Cbd.x);
        const float dd = depth_out(Cbd.y, T);
        if (o.color) {                   // (NULL: a records-only view -- the record below is its whole product)
            gstore(o.color + 0 * P + pix, cr);
            gstore(o.color + 1 * P + pix, cg);
            gstore(o.color + 2 * P + pix, cb);
        }
        if (o.depth) gstore(o.depth + pix, dd);
        if (AUX) {
            if (o.final_T) gstore(o.final_T + pix, T);
            if (o.n_contrib) gstore(o.n_contrib + pix, last);
        }
        if (rec) {
            gstore(rec + 3 * pix + 0, quant_u8(cr));
            gstore(rec + 3 * pix + 1, quant_u8(cg));
            gstore(rec + 3 * pix + 2, quant_u8(cb));
            gstore(reinterpret_cast<uint16_t*>(rec + rec_depth + 2 * pix), quant_mm(dd));
        }
    }
    if (want_sem || sem_background) {
        // the objects-only image's pixel; a tile without object entries holds the background (= fmaf(1, bg, 0) of the
        // general form, bit for bit)
        const float sr = want_sem ? fmaf(Ts, cam.bg[0], Srg.x) : cam.bg[0];
        const float sg = want_sem ? fmaf(Ts, cam.bg[1], Srg.y) : cam.bg[1];
        const float sb = want_sem ? fmaf(Ts, cam.bg[2], Sbd.x) : cam.bg[2];
        if (inside) {
            if (ve.sem_color) {          // (NULL: records only -- the semantic image lives on as the record's mask planes)
                gstore(ve.sem_color + 0 * P + pix, sr);
                gstore(ve.sem_color + 1 * P + pix, sg);
                gstore(ve.sem_color + 2 * P + pix, sb);
            }
            if (ve.sem_depth) gstore(ve.sem_depth + pix, want_sem ? depth_out(Sbd.y, Ts) : 0.0f);
        }
        // the K masks of that pixel, from the registers that hold it (round 3: a separate pass re-read 12 P bytes per view)
        if ((ve.sem_masks || rec) && sem.mask_colors)
            pixel_masks(sr, sg, sb, sem.mask_colors, 0, sem.k, sem.mask_thr, ve.sem_masks, P, pix, inside,
                        rec ? rec + rec_masks : nullptr);
    }
}

// The kernel: one wave per (view, tile, quarter) work item.  In the FUSED form a quarter whose tile holds no object entry
// at all (most of the image) runs the plain loop and only copies the background into the semantic image: the fused
// loop's per-entry bookkeeping (object bit, second set of masks, their branches) was costing EVERY entry of EVERY quarter
// -- of the 27 us per view the semantic image cost on C3, 11 were this.  (Tried on top, slower: routing the environment
// entries of fused quarters through a copy of the plain blend -- three inlined copies of the blend cost more than the
// scalar tests they saved.)
template <bool AUX, bool FUSED, bool LAYERED = false>
__global__ __launch_bounds__(WAVE) PGR_COMP_OCC void composite_quarter_kernel(const ViewEntry* __restrict__ views,
                                                                              uint32_t items_per_view,
                                                                              const uint32_t* __restrict__ work_order,
                                                                              SemanticDev sem, uint32_t n_slots) {
    constexpr int PAIRS = WAVE_BATCH / 2 + 1;     // +1: a null entry pads an odd batch
    __shared__ float4 s_g[3 * PAIRS];
    __shared__ float4 s_c[2 * PAIRS];
    __shared__ float4 s_s[FUSED ? 2 * PAIRS : 1];
    __shared__ uint32_t s_i[AUX ? 2 * PAIRS : 1];
    __shared__ float s_col[FUSED ? 3 * SEM_LDS_OBJECTS : 1];
    if constexpr (LAYERED) {
        // A layered call has n_layers times the slots and nine of ten are INVALID (empty lists get no item): one
        // single-wave workgroup per SLOT made the launch dispatch-bound (2.56 M workgroups per 32-view batch of 8 layers:
        // 1.75 ms, 0.7 ns each, whatever they did).  The grid is a fixed number of waves that stride over the slots; a
        // stream's valid items sit at its front, and gridDim.x is a multiple of NUM_XCD, so a wave stays in its stream and
        // stops at the first INVALID slot.
        for (uint32_t pos = blockIdx.x; pos < n_slots; pos += gridDim.x) {
            uint32_t item = work_order[pos];
            if (item == INVALID_ITEM) return;
            const uint32_t view = item / items_per_view;
            item -= view * items_per_view;
            const ViewEntry& ve = views[view];
            if (ve.counters[1] || !ve.sem_masks) continue;
            composite_quarter<false, false, true>(ve, item, sem, 0, false, s_g, s_c, s_s, s_i, nullptr);
            __syncthreads();
        }
        return;
    } else {
    uint32_t item = work_order ? work_order[blockIdx.x] : blockIdx.x;
    if (item == INVALID_ITEM) return;
    const uint32_t view = item / items_per_view;
    item -= view * items_per_view;
    const ViewEntry& ve = views[view];
    if (ve.counters[1] || !(ve.out.color || ve.record)) return;
    if constexpr (FUSED) {
        // the objects-only image is wanted as an image, or (records-only view) as the mask planes of the frame record
        const bool want_sem = ve.sem_color != nullptr || (ve.record != nullptr && sem.mask_colors != nullptr);
        // object entries live in [0, n_sem).  readfirstlane: the value arrives through a vector load; everything derived
        // from it (the semantic masks, the loop exits) must stay on the scalar unit
        const int n_sem = want_sem ? __builtin_amdgcn_readfirstlane((int)ve.obj_last[item >> 2]) : 0;
        if (n_sem > 0) {
            const bool table = sem.k <= SEM_LDS_OBJECTS;
            if (table) {
                for (int i = threadIdx.x; i < 3 * sem.k; i += WAVE) s_col[i] = gload(sem.colors + i);
                __syncthreads();
            }
            composite_quarter<AUX, true>(ve, item, sem, n_sem, false, s_g, s_c, s_s, s_i, table ? s_col : nullptr);
        } else {
            composite_quarter<AUX, false>(ve, item, sem, 0, want_sem, s_g, s_c, s_s, s_i, nullptr);
        }
    } else {
        composite_quarter<AUX, false>(ve, item, sem, 0, false, s_g, s_c, s_s, s_i, nullptr);
    }
    }
}

constexpr uint32_t LAYERED_GRID = NUM_XCD * 8192;     // waves of a layered compositor launch (8 x what the chip holds)

template <bool AUX, bool FUSED, bool LAYERED = false>
inline void launch_composite(uint32_t slots, hipStream_t stream, const ViewEntry* views, uint32_t items_per_view,
                             const uint32_t* work_order, SemanticDev sem) {
    // (measured on C3, 8 layers, 32 views: 2 048 .. 65 536 waves per XCD stream all land within 3 % -- 2.91 .. 3.09 ms per batch)
    const uint32_t grid = LAYERED ? (slots < LAYERED_GRID ? (slots + NUM_XCD - 1) / NUM_XCD * NUM_XCD : LAYERED_GRID) : slots;
    composite_quarter_kernel<AUX, FUSED, LAYERED><<<grid, WAVE, 0, stream>>>(views, items_per_view, work_order, sem, slots);
}

// Layered call: the mask of a pixel no list entry touches is the background's verdict -- || bg - c_k ||_2 <= thr, the value
// pixel_masks() computes for an untouched pixel (fmaf(1, bg, 0) = bg), bit for bit.  Planes are pre-filled with it, and
// only the non-empty (layer, tile) lists get compositor waves (9 of 10 lists of a silhouette pass are empty).
// A layer WITHOUT ANY Gaussian (an object id no Gaussian carries) keeps a zero plane, whatever the background: the
// reference never renders such an object and leaves its column of the mask array 0
// (/root/reference/src/gs/render.py:44-63) -- the same rule as an empty scene (N == 0: every plane 0).  layer_id is
// non-decreasing, so presence is one binary search per workgroup.
// grid = (LAYER_FILL_BLOCKS, n_layers, n_views), 256 threads, 16 pixels per store, grid-stride over the plane: the binary
// search is ~18 dependent scalar loads (3-4 us) per workgroup -- with one workgroup per 1024 pixels (round 4: 160 000 of them per
// 32-view batch of 8 layers) the launch lasted 0.29 ms for 164 MB of stores; 10 000 workgroups of forty 16-byte stores per thread
// leave the time to the stores.
constexpr unsigned LAYER_FILL_BLOCKS = 40;
__global__ __launch_bounds__(256) void layer_mask_fill_kernel(const ViewEntry* __restrict__ views, const float* __restrict__ colors,
                                                              float thr, size_t P, const int32_t* __restrict__ layer_id, int n) {
    const ViewEntry& ve = views[blockIdx.z];
    if (!ve.sem_masks) return;
    const CameraDev& cam = *ve.cam;
    const int c = blockIdx.y;
    int lo = 0, hi = n;                       // first Gaussian with layer_id >= c + 1 (wave-uniform: scalar loads)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (layer_id[mid] < c + 1) lo = mid + 1; else hi = mid;
    }
    const bool present = lo < n && layer_id[lo] == c + 1;
    const float d0 = cam.bg[0] - colors[3 * c], d1 = cam.bg[1] - colors[3 * c + 1], d2 = cam.bg[2] - colors[3 * c + 2];
    const uint8_t m = present && sqrtf(d0 * d0 + d1 * d1 + d2 * d2) <= thr ? 1 : 0;
    uint8_t* plane = ve.sem_masks + (size_t)c * P;
    // bytes in front of the plane's first 16-byte boundary, then whole quads, then the bytes behind the last one
    const size_t head = min(P, (size_t)((16u - (unsigned)(reinterpret_cast<uintptr_t>(plane) & 15u)) & 15u));
    const size_t quads = (P - head) / 16;
    const uint32_t w = 0x01010101u * m;
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t q = {w, w, w, w};
    u32x4_t* body = reinterpret_cast<u32x4_t*>(plane + head);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < quads; k += stride) *(PGR_GLOBAL u32x4_t*)(body + k) = q;
    if (blockIdx.x == 0 && threadIdx.x < 32) {
        if (threadIdx.x < 16) {
            if (threadIdx.x < head) gstore(plane + threadIdx.x, m);
        } else {
            const size_t p = head + 16 * quads + (threadIdx.x - 16);
            if (p < P) gstore(plane + p, m);
        }
    }
}

// object ids of the object Gaussians as one byte each (see SemanticDev::object_u8)
__global__ void pack_object_ids_kernel(const int32_t* __restrict__ object_id, int n_env, int n, uint8_t* __restrict__ out) {
    const int i = n_env + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i - n_env] = (uint8_t)object_id[i];
}

// ---- work order --------------------------------------------------------------------------------
// The compositor's work items (view, tile, quarter) are laid out as NUM_XCD interleaved streams: position p
// belongs to stream p % 8, and the hardware dispatcher is observed to place workgroup b on XCD b % 8
// (MI355X_MICROARCH.md, a speed hint only -- any placement is correct).  Stream x owns every 8th band of
// ORDER_BAND_ROWS tile rows of every view, and inside a stream the items keep their row-major order, so a tile's
// four quarters and its horizontal neighbours -- whose lists share most of their Gaussians -- run back to back on
// the SAME XCD and find each other's gathers in its L2 (PMC before: 486 MB/view fetched from the fabric for
// 161 MB of algorithmic bytes).  Only a coarse longest-first split is kept (ORDER_CLASSES_USED length classes:
// the long lists of every stream start first, short ones back-fill).  Unused slots hold INVALID_ITEM.
// Contiguous bands per XCD share more vertically but leave XCDs idle when the image's work is uneven
// (measured 1.9x slower on C3); striping keeps every XCD's share statistically equal.
constexpr int ORDER_BAND_ROWS = 1;
constexpr int ORDER_CLASSES_USED = 3;            // > 4096, > 1024, rest
constexpr int ORDER_BINS = NUM_XCD * ORDER_CLASSES_USED;

// order_state layout (uint32): [ORDER_BINS] counters -> cursors, then [SORT_TIERS] lengths of the sort queues, then the
// ticket of tile_scan_kernel's workgroups (the last one turns the counters into cursors)
// Sort queues: every non-empty list of a view that did not overflow is one uint4 (item, first instance, keys, 0) in the
// queue of its tier -- 1..2048 keys, 2049..4096, 4097..8192, 8193..SORT_WINDOW_MAX (round 6: the windowed sort), longer
// (round 6: the split pre-pass, whose depth segments join the SEGMENT queue of the 4097..8192 tier's kernel), and the open-ended
// kernel's queue, which order_scatter fills only in one- and two-view calls and which otherwise receives what the windowed
// sort and the split pre-pass reject: one queue and one sort launch each.  A sort
// workgroup learns its list from that one word (the view's table entry arrives beside it through the scalar cache)
// instead of chasing item -> view table -> ranges -> keys; and a launch has no workgroups for empty tiles.
constexpr int SORT_TIERS = 6;
constexpr int SORT_WINDOW_MAX = 2 * (8192 - 256);   // tilebin.hip.h: two windows of the windowed sort's 8192-key image
constexpr int ORDER_DONE_WORD = ORDER_BINS + SORT_TIERS;
constexpr int ORDER_SEG_WORD = ORDER_DONE_WORD + 1; // entries of the segment queue
constexpr int ORDER_STATE_WORDS = ORDER_BINS + SORT_TIERS + 2;

#ifndef PGR_WINDOW_TIER
#define PGR_WINDOW_TIER 1        // 0: no windowed sort (A/B builds): lists of 8193..15872 keys take the next tier's path
#endif
#ifndef PGR_SPLIT_TIER
#define PGR_SPLIT_TIER 1         // 0: no split pre-pass (A/B builds): lists beyond the windowed sort go to the open-ended kernel
#endif
__device__ __forceinline__ int sort_tier(uint32_t len) {
    const int beyond = PGR_SPLIT_TIER ? 4 : 5;
    return len > (uint32_t)SORT_WINDOW_MAX ? beyond : (len > 8192u ? (PGR_WINDOW_TIER ? 3 : beyond) : (len > 4096u ? 2 : (len > 2048u ? 1 : 0)));
}

__device__ __forceinline__ int xcd_of_tile(int tile, int grid_x) { return (tile / grid_x / ORDER_BAND_ROWS) % NUM_XCD; }

__device__ __forceinline__ int coarse_class(uint32_t len) { return len > 4096u ? 0 : (len > 1024u ? 1 : 2); }

__host__ __device__ inline int max_band_rows(int grid_y) {
    const int bands = (grid_y + ORDER_BAND_ROWS - 1) / ORDER_BAND_ROWS;
    return (bands + NUM_XCD - 1) / NUM_XCD * ORDER_BAND_ROWS;
}

// (the counts per (stream, class) and their prefix sums: tile_scan_kernel, tilebin.hip.h)

// grid = (ceil(tiles/256), n_views), 256 threads.  work_order is pre-filled with INVALID_ITEM.  Ranks inside a
// workgroup follow the tile order (ballot prefix per bin), so row-major neighbours stay adjacent in their
// stream; one global atomic per non-empty (workgroup, bin) claims the slots.  Non-empty lists are also appended to the
// sort queue of their tier: sort_queue[tier * queue_stride + ...].
__global__ __launch_bounds__(256) void order_scatter_kernel(const ViewEntry* __restrict__ views, int tiles, int grid_x,
                                                            uint32_t* __restrict__ state,
                                                            uint32_t* __restrict__ work_order,
                                                            uint4* __restrict__ sort_queue, uint32_t queue_stride,
                                                            int merge_long_tiers, int skip_empty) {
    __shared__ uint32_t wave_cnt[4][ORDER_BINS];
    __shared__ uint32_t base[ORDER_BINS];
    __shared__ uint32_t n_queue_s[SORT_TIERS], queue_base_s[SORT_TIERS];
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    if (threadIdx.x < SORT_TIERS) n_queue_s[threadIdx.x] = 0;
    __syncthreads();
    const int t = blockIdx.x * 256 + threadIdx.x;
    int bin = -1, x = 0, tier = 0;
    uint32_t queue_rank = INVALID_ITEM;
    uint2 r = make_uint2(0u, 0u);
    if (t < tiles) {
        const ViewEntry& ve = views[blockIdx.y];
        const uint32_t overflowed = ve.counters[1];
        r = ve.ranges[t];
        const uint32_t len = r.y - r.x;
        x = xcd_of_tile(t, grid_x);
        if (!(skip_empty && len == 0u)) bin = x * ORDER_CLASSES_USED + coarse_class(len);
        if (len > 0u && !overflowed) {
            tier = sort_tier(len);
            // a launch of one or two views has a handful of lists per long tier: three launches that each wait for their
            // longest list.  They all go to the open-ended tier's kernel, which sorts any length.
            if (merge_long_tiers && tier > 0) tier = SORT_TIERS - 1;
            queue_rank = atomicAdd(&n_queue_s[tier], 1u);
        }
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
    if (threadIdx.x >= 64 && threadIdx.x < 64 + SORT_TIERS && n_queue_s[threadIdx.x - 64])
        queue_base_s[threadIdx.x - 64] = atomicAdd(&state[ORDER_BINS + threadIdx.x - 64], n_queue_s[threadIdx.x - 64]);
    __syncthreads();
    if (t < tiles && bin >= 0) {
        uint32_t before = 0;
        for (int w = 0; w < wave; ++w) before += wave_cnt[w][bin];
        const uint32_t r0 = base[bin] + ITEMS_PER_TILE * (before + rank);     // position inside stream x
        const uint32_t list = (uint32_t)blockIdx.y * (uint32_t)tiles + (uint32_t)t;
        for (uint32_t k = 0; k < ITEMS_PER_TILE; ++k) work_order[(size_t)(r0 + k) * NUM_XCD + x] = ITEMS_PER_TILE * list + k;
        if (queue_rank != INVALID_ITEM)
            sort_queue[(size_t)tier * queue_stride + queue_base_s[tier] + queue_rank] = make_uint4(list, r.x, r.y - r.x, 0u);
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
    // A wave whose 64 pixels are all farther than thr from colour c along ONE channel (the background, other objects'
    // pixels) skips the squares and the square root: dist >= |d_i| (1 - 2^-23) in fp32 -- the sum of non-negative terms and
    // the correctly rounded root are monotone, sqrt(fl(x^2)) is within an ulp of |x| -- so |d_i| > thr (1 + 2^-20) decides.
    const float far = thr * 1.000001f;
    for (int c = 0; c < k; ++c) {
        const float d0 = r - colors[3 * c], d1 = g - colors[3 * c + 1], d2 = b - colors[3 * c + 2];
        const bool surely_out = fabsf(d0) > far || fabsf(d1) > far || fabsf(d2) > far;
        uint8_t m = 0;
        if (__ballot(!surely_out) != 0ull) {
            const float dist = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
            m = dist <= thr ? 1 : 0;
        }
        masks[(size_t)c * P + p] = m;
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

// Batch form of the two casts plus the K visibility masks of a frame as bit planes: one launch turns a finished batch
// into what leaves the GPU (disk writers, or the gather to the root rank: SURVEY.md section 8e's 3.84 MB per 800x800
// frame = u8 x 3 + u16 + one mask byte).  grid.y = image; mask byte j of a pixel holds masks 8j .. 8j+7 (bit k % 8).
__global__ void pack_frames_kernel(const float* __restrict__ color, const float* __restrict__ depth,
                                   const uint8_t* __restrict__ masks, size_t P, int k, uint8_t* __restrict__ rgb_hwc,
                                   uint16_t* __restrict__ depth_mm, uint8_t* __restrict__ mask_bits) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const size_t b = blockIdx.y;
    if (color && rgb_hwc) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = color[(b * 3 + c) * P + p] * 255.0f;
            v = fminf(fmaxf(v, -2147483520.0f), 2147483520.0f);
            rgb_hwc[(b * P + p) * 3 + c] = (uint8_t)((int)v & 0xFF);
        }
    }
    if (depth && depth_mm) {
        float v = depth[b * P + p] * 1000.0f;
        v = fminf(fmaxf(v, -2147483520.0f), 2147483520.0f);
        depth_mm[b * P + p] = (uint16_t)((int)v & 0xFFFF);
    }
    if (masks && mask_bits) {
        const int bytes = (k + 7) / 8;
        for (int j = 0; j < bytes; ++j) {
            uint32_t bits = 0;
            for (int m = 8 * j; m < min(k, 8 * j + 8); ++m)
                bits |= (masks[(b * (size_t)k + m) * P + p] ? 1u : 0u) << (m & 7);
            mask_bits[(b * P + p) * bytes + j] = (uint8_t)bits;
        }
    }
}

// One record per frame (include/pegasus_raster.h PgrRecordLayout): the two casts of pack_frames_kernel and the mask bit
// planes, written into the frame's contiguous record -- the unit the gather to the root rank and the disk writers move.
__global__ void pack_records_kernel(const float* __restrict__ color, const float* __restrict__ depth,
                                    const uint8_t* __restrict__ masks, size_t P, int k, uint8_t* __restrict__ records,
                                    size_t record_stride, size_t off_depth, size_t off_masks) {
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const size_t b = blockIdx.y;
    uint8_t* rec = records + b * record_stride;
    if (color) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = color[(b * 3 + c) * P + p] * 255.0f;
            v = fminf(fmaxf(v, -2147483520.0f), 2147483520.0f);
            rec[p * 3 + c] = (uint8_t)((int)v & 0xFF);
        }
    }
    if (depth) {
        float v = depth[b * P + p] * 1000.0f;
        v = fminf(fmaxf(v, -2147483520.0f), 2147483520.0f);
        reinterpret_cast<uint16_t*>(rec + off_depth)[p] = (uint16_t)((int)v & 0xFFFF);
    }
    if (masks) {
        const int bytes = (k + 7) / 8;
        for (int j = 0; j < bytes; ++j) {
            uint32_t bits = 0;
            for (int m = 8 * j; m < min(k, 8 * j + 8); ++m)
                bits |= (masks[(b * (size_t)k + m) * P + p] ? 1u : 0u) << (m & 7);
            rec[off_masks + p * bytes + j] = (uint8_t)bits;
        }
    }
}

}  // namespace pgr
