// tilebin.hip.h -- sync-free tile binning for gfx950: (Gaussian, tile) instances -> per-tile lists
// sorted by (depth bits, Gaussian index).  Replaces SURVEY.md section 8a rows a6-a9 (scan, emission,
// global 44-bit radix sort, range search) of the reference design with a layout that never leaves the
// chip's fast paths.  Only instances that can reach alpha >= 1/255 somewhere in their tile are listed
// (tile_may_contribute below): the images are unchanged, the lists are ~1.7x shorter.
//
//   bin_count    one workgroup per CHUNK of Gaussians: tile histogram in LDS (LDS atomics), then ONE
//                coalesced global atomicAdd per (chunk, touched tile) that both accumulates the tile
//                total and RESERVES the chunk's slice of the tile's list (returned offset -> rel[][]).
//                Scattered per-instance global atomics measured 9 G/s on MI355X (700 us per view);
//                this form measured 16 us (scripts/microbench/atomics_bin.hip).
//   tile_scan    exclusive scan of the tile totals -> ranges[tile] = [start,end), instance total,
//                overflow flag -- all on the device, no host round trip.
//   bin_scatter  same chunks: LDS cursors = range start + rel; every instance takes its slot with an
//                LDS atomic and stores (depth bits, index) = 8 B.
//   tile_sort    one workgroup per tile list: sorted by the 64-bit key (depth bits << 32 | index) in LDS -- a bucket sort on
//                the depth bits with position-owned ranking, a merge sort where depths pile up -- which is exactly the order
//                of a stable sort of Gaussian-major emission by (tile, depth), the order the compositor's blending depends
//                on.  One launch per length tier: <= 2048 keys (256 threads, 7 workgroups per CU), <= 4096 (512 x 8),
//                <= 8192 (512 x 16; also the depth segments of the longest lists), 8193..15872 (windowed sort: depths in
//                registers, the key image filled window by window; its launch carries the split pre-pass that cuts longer
//                lists into segments), and the open-ended kernel (157 KiB: one- and two-view calls, and whatever the
//                others reject).
//
// Everything is integer/bit work: bit-exact against the oracle's sorted lists.
#pragma once
#include <type_traits>

#include "composite.hip.h"       // work-order constants (tile_scan_kernel also counts the compositor's work-order bins)
#include "cull.hip.h"
#include "pgr_common.h"

namespace pgr {

constexpr int BIN_CHUNK = 4096;        // Gaussians per binning workgroup
#ifndef PGR_BIN_THREADS
#define PGR_BIN_THREADS 512     // 8 waves share a chunk's 64 groups by ticket; 37 KiB of LDS at 2500 tiles = four per CU
#endif                           // (measured: 1024 threads +5 % in the count walk, 768 and 256 +8..10 %)
constexpr int BIN_THREADS = PGR_BIN_THREADS;
constexpr int BIN_LDS_TILES = 16384;   // tiles histogrammed per LDS pass (64 KiB)
constexpr int SORT_THREADS = 256;

// rect packed as 4 x uint16: minx, miny, maxx, maxy (exclusive); all zero = culled
__device__ __forceinline__ uint2 pack_rect(int minx, int miny, int maxx, int maxy) {
    return make_uint2((uint32_t)minx | ((uint32_t)miny << 16), (uint32_t)maxx | ((uint32_t)maxy << 16));
}

struct BinView {                 // per-view pointers used by the binning kernels (device table)
    const uint2* rects;          // [n] packed CANDIDATE tile rectangles (preprocess.hip.h candidate_rect)
    const float4* splats;        // [n, 3] records (preprocess.hip.h): q0 = (x,y,A,B), q1 = (C,op,B/C,B/A), q2 = (r,g,b,depth)
    uint32_t* tile_count;        // [tiles] zero-filled before bin_count
    uint32_t* rel;               // [chunks, tiles]
    uint2* ranges;               // [tiles]
    uint32_t* counters;          // [0] total instances, [1] overflow flag
    uint2* bucket;               // [max_instances] (depth bits, index), unsorted per tile
    uint32_t* gauss_sorted;      // [max_instances]
    uint64_t* alt;               // [max_instances] second key buffer for lists beyond the LDS tiers
    uint32_t* obj_last;          // [tiles] zero-filled; 1 + position of the last object entry of the sorted list
    int32_t n_env;               // Gaussians < n_env are environment; < 0: no semantic pass wanted
    // Exact depth ties are broken by tie_index[i] instead of the position i (PgrScene::tie_index: a scene stored in
    // another order than the caller's keeps the caller's tie order).  Lists are sorted by (depth, position) as always;
    // the bucket sort then looks the tie index up for the few keys that share their depth with another key of the
    // list and corrects their ranks; the merge-sort fallback (piled-up depths) sorts (depth, tie index) keys and maps
    // them back to positions through tie_inv.  NULL = the position is the tie index.
    const int32_t* tie_index;
    const uint32_t* tie_inv;
    // What the count walk found, kept for the scatter walk (see bin_kernel): a fixed region per 64-Gaussian group (its 64
    // depths + one ballot per window of candidates), in the space the sort only needs AFTER the scatter walk (alt +
    // gauss_sorted).
    uint2* records;
};


__device__ __forceinline__ uint32_t dpp_u32(uint32_t old, uint32_t v, int ctrl, int row_mask, bool bound) {
    switch (ctrl) {   // the builtin wants immediates
        case 0x111: return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x111, 0xF, 0xF, true);
        case 0x112: return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x112, 0xF, 0xF, true);
        case 0x114: return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x114, 0xF, 0xF, true);
        case 0x118: return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x118, 0xF, 0xF, true);
        case 0x142: return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x142, 0xA, 0xF, false);
        default:    return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x143, 0xC, 0xF, false);
    }
}

// inclusive prefix sum over the 64 lanes of a wave (row shifts, then the two row broadcasts of gfx9)
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v) {
    v += dpp_u32(0, v, 0x111, 0xF, true);   // row_shr:1
    v += dpp_u32(0, v, 0x112, 0xF, true);   // row_shr:2
    v += dpp_u32(0, v, 0x114, 0xF, true);   // row_shr:4
    v += dpp_u32(0, v, 0x118, 0xF, true);   // row_shr:8
    v += dpp_u32(0, v, 0x142, 0xA, false);  // row_bcast:15 -> rows 1, 3
    v += dpp_u32(0, v, 0x143, 0xC, false);  // row_bcast:31 -> rows 2, 3
    return v;
}

// inclusive prefix MAX over the 64 lanes (0 is the identity: out-of-row DPP reads return 0)
__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t v) {
    v = max(v, dpp_u32(0, v, 0x111, 0xF, true));
    v = max(v, dpp_u32(0, v, 0x112, 0xF, true));
    v = max(v, dpp_u32(0, v, 0x114, 0xF, true));
    v = max(v, dpp_u32(0, v, 0x118, 0xF, true));
    v = max(v, dpp_u32(0, v, 0x142, 0xA, false));
    v = max(v, dpp_u32(0, v, 0x143, 0xC, false));
    return v;
}

// Both binning passes walk the same candidate space: every tile of every Gaussian's rectangle.  Rectangle
// areas vary from 1 to hundreds of tiles, so a loop per Gaussian leaves most lanes idle (measured 6x on
// MI355X).  Instead each wave takes 64 Gaussians, prefix-sums their areas (DPP scan), and its lanes sweep the
// FLATTENED candidate index space 64 at a time; every lane evaluates exactly one candidate per iteration.
//   candidate c -> owning Gaussian: every owner whose segment starts inside the 64-candidate window drops a head
//     flag (position << 6 | lane) at its start; a DPP prefix-max hands every candidate the last head at or before
//     it (candidates before the first head continue the previous window's owner).  ~14 VALU per iteration where a
//     binary search over the prefix with ds_bpermute took 6 LDS round trips.
//   owner's data (cull record, rectangle, depth bits): parked once per 64 Gaussians in a per-wave LDS table of
//     three float4 and fetched with three ds_read_b128 -- not 12 ds_bpermute per iteration.
//   tile inside the rectangle: k / w by a reciprocal multiply with an exact +-1 correction, no integer division.
// The binning kernels were VALU-bound (SQ_ACTIVE_INST_VALU ~ duration): these three cut the instruction count per
// iteration from ~180 to ~85.
//   SCATTER = false: lds[] = tile histogram; flushed with one coalesced reserving atomic per touched tile.
//   SCATTER = true : lds[] = write cursors (range start + this chunk's reserved offset).
// The count walk leaves what it found behind, ONE BIT PER CANDIDATE: per window of a group's walk the ballot of the
// predicate (8 bytes per 64 candidates, ~1 MB per view), in front of them the group's 64 depths (256 contiguous bytes
// instead of 64 words at a 48-byte stride).  Every 64-Gaussian group owns a fixed 768-byte region (64 depths + 62 ballots:
// up to 3 968 candidates; a group with more is simply evaluated twice as before), so there is nothing to reserve and
// nothing to look up: region = group index x 768 B, in the space the sort only needs after the scatter walk (alt +
// gauss_sorted).  The scatter walk then repeats the candidate enumeration -- same groups, same windows -- but takes the
// verdict from the bit instead of evaluating the predicate again, skips windows without a set bit, and reads neither the
// splat records nor their strided depths.
constexpr int BIN_STAGE_WORDS = WAVE * 12 + WAVE;         // per wave: 64 x 3 float4 + 64 head words
// One- and two-view calls launch 488 workgroups (C3) on 256 CUs and last as long as their heaviest chunk (7x the mean number
// of candidates in a C3 view): 16 waves per chunk instead of 8 -- count walk 44.7 -> 33.1 us, scatter walk 58.4 -> 51.1 us
// for a single view.  (Several workgroups per chunk, each with its own `rel` row: the scatter walk gains what the count
// walk loses to the extra zero-fills and flushes of 2 500 tile counters -- 2 parts 43 / 45 us, 4 parts 55 / 38: dropped.)
constexpr int BIN_THREADS_SMALL = 1024;
// tiles: tiles histogrammed per LDS pass (all of them up to BIN_LDS_TILES; a LAYERED call: the tiles of ONE layer)
__host__ __device__ inline size_t bin_lds_bytes(int tiles, int threads) {
    return ((size_t)(tiles < BIN_LDS_TILES ? tiles : BIN_LDS_TILES) + 3) / 4 * 16 + (size_t)(threads / WAVE) * BIN_STAGE_WORDS * 4 + 16;
}

// LAYERED binning (pgr_forward_layers_async): the view is one tall image of n_layers x layer_rows tile rows, `tiles` =
// n_layers x layer_tiles.  Gaussians are ordered by layer, so a chunk of 4096 holds one layer (two at a boundary): the
// chunk histograms ONE layer per LDS pass -- lo = (layer - 1) x layer_tiles, span = layer_tiles -- over the layers its
// first and last Gaussian name; a 64-Gaussian group takes part in the passes of the layers it holds, and a group that
// straddles a boundary keeps no verdict bits (its region would be written once per pass).
struct BinLayers {
    const int32_t* layer_id;     // [n] non-decreasing; NULL = not layered
    int32_t layer_tiles;         // grid_x * grid_y of one layer
    int32_t layer_rows;          // grid_y of one layer
    int32_t n_layers;
};
constexpr uint32_t VERDICT_WINDOWS = 62;                               // ballots per group region
constexpr uint32_t VERDICT_REGION_WORDS = 192;                         // 4-byte words: 64 depths + 2 x 62 (+ 4 unused)

// vis: blockcull.hip.h's visibility words (NULL = all visible); a 64-Gaussian group whose bit is clear has no rectangle
// to read (the preprocess did not write one).
//
// The 64 groups of a chunk are handed out to the waves one at a time (an LDS ticket): walks last from nothing to dozens
// of iterations, and a static split left most waves of a workgroup waiting for its slowest one.  A group's rectangles and
// records are fetched with all loads in flight at once (a visible group's Gaussians are nearly all visible, so the
// records do not wait for the rectangle).
//
// verdict_groups: how many groups have a region (0: none -- more tiles than one LDS pass holds, or PGR_BIN_RECORDS=0).
template <bool SCATTER, int THREADS>
__global__ __launch_bounds__(THREADS) void bin_kernel(const BinView* __restrict__ views, int n, int grid_x,
                                                          int tiles, int W, int H, const uint32_t* __restrict__ vis,
                                                          int vis_words, int verdict_groups,
                                                          BinLayers layers = BinLayers{nullptr, 0, 0, 0}) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    constexpr int BIN_WAVES = THREADS / WAVE;
    const BinView& bv = views[blockIdx.y];
    if (SCATTER && gload(bv.counters + 1)) return;   // overflow: reported by the host, nothing may be written past the buffers
    const int chunk = blockIdx.x;
    const int begin = chunk * BIN_CHUNK, end = min(n, begin + BIN_CHUNK);
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const bool layered = layers.layer_id != nullptr;
    const int hist_words = (min(layered ? layers.layer_tiles : tiles, BIN_LDS_TILES) + 3) / 4 * 4;
    float4* const stage = reinterpret_cast<float4*>(lds + hist_words + wave * BIN_STAGE_WORDS);   // [64][3]
    // lanes talk to each other through this array inside one wave: DS operations of a wave execute in order, and a
    // compiler memory barrier keeps the load behind the stores.  (NOT volatile: a volatile generic pointer turns the
    // three accesses into flat_store/flat_load ... sc0 sc1 with a vmcnt(0) wait each.)
    uint32_t* const heads = reinterpret_cast<uint32_t*>(stage + WAVE * 3);                           // [64]
    // ctrl[0] = next ticket, ctrl[1..2] = visibility bits of the chunk's groups (fetched once, ahead of the walk)
    uint32_t* const ctrl = lds + hist_words + BIN_WAVES * BIN_STAGE_WORDS;
    constexpr int GROUPS = BIN_CHUNK / WAVE;
    static_assert(GROUPS == WAVE, "one visibility bit per lane of a wave");
    // LDS passes: windows of BIN_LDS_TILES tiles, or (layered) the layers this chunk's Gaussians belong to
    int pass_first = 0, pass_last = (tiles - 1) / BIN_LDS_TILES;
    if (layered) {
        if (begin >= end) return;
        pass_first = max(1, min(layers.n_layers + 1, (int)gload(layers.layer_id + begin)));
        pass_last = min(layers.n_layers, (int)gload(layers.layer_id + end - 1));
    }
    for (int lp = pass_first; lp <= pass_last; ++lp) {
        const int lo = layered ? (lp - 1) * layers.layer_tiles : lp * BIN_LDS_TILES;
        const int span = layered ? layers.layer_tiles : min(BIN_LDS_TILES, tiles - lo);
        const int row_off = layered ? (lp - 1) * layers.layer_rows : 0;     // tile row of the layer's first row
        for (int t = threadIdx.x; t < span; t += THREADS)
            lds[t] = SCATTER ? gload(bv.ranges + lo + t).x + gload(bv.rel + (size_t)chunk * tiles + lo + t) : 0u;
        if (wave == 0) {
            const int b = begin + lane * WAVE;
            bool visible = b < end;
            if (visible && vis) visible = (vis[(size_t)(b / WAVE) * vis_words + (blockIdx.y >> 5)] >> (blockIdx.y & 31)) & 1u;
            const uint64_t m = __ballot(visible);
            if (lane == 0) { ctrl[0] = 0u; ctrl[1] = (uint32_t)m; ctrl[2] = (uint32_t)(m >> 32); }
        }
        __syncthreads();
        const uint64_t vis_bits = ((uint64_t)ctrl[2] << 32) | ctrl[1];
        // next visible group of the chunk, or GROUPS
        auto take_ticket = [&]() {
            for (;;) {
                uint32_t ticket = 0;
                if (lane == 0) ticket = atomicAdd(&ctrl[0], 1u);
                ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)ticket);
                if (ticket >= (uint32_t)GROUPS || ((vis_bits >> ticket) & 1ull)) return ticket;
            }
        };
        for (uint32_t ticket = take_ticket(); ticket < (uint32_t)GROUPS; ticket = take_ticket()) {
            const int base = begin + (int)ticket * WAVE;
            bool has_region = base / WAVE < verdict_groups;
            if (layered) {                   // (scalar: the group's first and last layer)
                const int l0 = gload(layers.layer_id + base), l1 = gload(layers.layer_id + min(base + WAVE, end) - 1);
                if (lp < l0 || lp > l1) continue;
                has_region = has_region && l0 == l1;
            }
            uint32_t* const region = reinterpret_cast<uint32_t*>(bv.records) + (size_t)(base / WAVE) * VERDICT_REGION_WORDS;
            uint64_t* const ballots = reinterpret_cast<uint64_t*>(region + WAVE);
            const int i = base + lane;
            const int ic = i < end ? i : end - 1;
            uint2 r = gload(bv.rects + ic);
            float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
            float depth = 0.0f;
            uint64_t my_ballot = 0;                  // scatter walk: lane l = the verdicts of window l
            if (SCATTER && has_region) {             // (requested before the candidate count says whether they were written)
                depth = __uint_as_float(gload(region + lane));
                if (lane < (int)VERDICT_WINDOWS) my_ballot = gload(ballots + lane);
            } else {
                q0 = gload(bv.splats + (size_t)ic * 3);
                q1 = gload(bv.splats + (size_t)ic * 3 + 1);
                if (SCATTER || has_region) depth = gload(reinterpret_cast<const float*>(bv.splats + (size_t)ic * 3 + 2) + 3);
            }
            if (i >= end) r = make_uint2(0u, 0u);
            const int w = (int)(r.y & 0xffff) - (int)(r.x & 0xffff), h = (int)(r.y >> 16) - (int)(r.x >> 16);
            const uint32_t area = (w > 0 && h > 0) ? (uint32_t)(w * h) : 0u;
            const uint32_t incl = wave_inclusive_scan(area);
            const uint32_t excl = incl - area;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
            const bool bits = has_region && total <= VERDICT_WINDOWS * (uint32_t)WAVE;   // same verdict in both walks
            if (SCATTER && has_region && !bits) {
                q0 = gload(bv.splats + (size_t)ic * 3);
                q1 = gload(bv.splats + (size_t)ic * 3 + 1);
                depth = gload(reinterpret_cast<const float*>(bv.splats + (size_t)ic * 3 + 2) + 3);
            }
            // one group's walk, compiled per combination of FROM_BITS (scatter walk: verdicts are read) and WRITE_BITS
            // (count walk: verdicts are kept): the loop bodies are free of mode branches (one body with run-time flags:
            // scatter walk 22.5 -> 27 us)
            auto walk = [&](auto from_bits_c, auto write_bits_c) {
                constexpr bool FROM_BITS = decltype(from_bits_c)::value, WRITE_BITS = decltype(write_bits_c)::value;
                float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0;
                if (area) {
                    uint32_t flags = 0;
                    if (!FROM_BITS) {
                        const CullSplat cs = make_cull_splat(make_float2(q0.x, q0.y), make_float4(q0.z, q0.w, q1.x, q1.y), q1.z, q1.w);
                        s0 = make_float4(cs.mx, cs.my, cs.A, cs.B);
                        s1 = make_float4(cs.C, cs.rBC, cs.rBA, cs.tau);
                        flags = cs.flags;
                    }
                    s2 = make_float4(__uint_as_float(flags | ((uint32_t)w << 2)), __uint_as_float(r.x), 1.0f / (float)w, depth);
                }
                if (!FROM_BITS) { stage[lane * 3 + 0] = s0; stage[lane * 3 + 1] = s1; }
                stage[lane * 3 + 2] = s2;
                __builtin_amdgcn_wave_barrier();
                if (WRITE_BITS && total) gstore(region + lane, __float_as_uint(depth));
                uint32_t carry_key = 0;        // owner of the candidate just before the window: (start + 1) << 6 | lane
                uint32_t it = 0;
                for (uint32_t c0 = 0; c0 < total; c0 += WAVE, ++it) {
                    uint64_t verdicts = ~0ull;
                    if (FROM_BITS) {
                        const uint32_t vlo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_ballot, (int)it);
                        const uint32_t vhi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_ballot >> 32), (int)it);
                        verdicts = ((uint64_t)vhi << 32) | vlo;
                    }
                    heads[lane] = 0u;
                    asm volatile("" ::: "memory");
                    if (area && excl - c0 < (uint32_t)WAVE) heads[excl - c0] = ((excl + 1u) << 6) | (uint32_t)lane;
                    asm volatile("" ::: "memory");
                    uint32_t key = wave_inclusive_max(heads[lane]);     // same wave: LDS ops are ordered
                    asm volatile("" ::: "memory");
                    key = key ? key : carry_key;
                    carry_key = (uint32_t)__builtin_amdgcn_readlane((int)key, WAVE - 1);
                    if (FROM_BITS && verdicts == 0ull) continue;          // nothing of this window is listed
                    const uint32_t c = c0 + (uint32_t)lane;
                    const int g = (int)(key & 63u);
                    const uint32_t k = c - ((key >> 6) - 1u);
                    const float4 o2 = stage[g * 3 + 2];
                    const uint32_t fw = __float_as_uint(o2.x), rlo = __float_as_uint(o2.y);
                    const int ow = (int)(fw >> 2);
                    // ty = k / w: reciprocal estimate, then an exact +-1 correction
                    int ty = (int)(((float)k + 0.5f) * o2.z);
                    int tx = (int)k - ty * ow;
                    if (tx < 0) { --ty; tx += ow; } else if (tx >= ow) { ++ty; tx -= ow; }
                    const int x = (int)(rlo & 0xffff) + tx, y = (int)(rlo >> 16) + ty;
                    const int t = y * grid_x + x - lo;
                    bool pass = c < total && (unsigned)t < (unsigned)span;
                    if (FROM_BITS) {
                        pass = pass && ((verdicts >> lane) & 1ull);
                    } else if (pass) {
                        const float4 o0 = stage[g * 3 + 0], o1 = stage[g * 3 + 1];
                        CullSplat cs;
                        cs.mx = o0.x; cs.my = o0.y; cs.A = o0.z; cs.B = o0.w;
                        cs.C = o1.x; cs.rBC = o1.y; cs.rBA = o1.z; cs.tau = o1.w; cs.flags = fw & 3u;
                        pass = tile_may_contribute(cs, x, y - row_off, W, H);
                    }
                    if (WRITE_BITS) {
                        const uint64_t m = __ballot(pass);
                        if (lane == 0) gstore(ballots + it, m);
                    }
                    if (pass) {
                        const uint32_t slot = atomicAdd(&lds[t], 1u);
                        if (SCATTER) gstore(bv.bucket + slot, make_uint2(__float_as_uint(o2.w), (uint32_t)(base + g)));
                    }
                }
            };
            if (SCATTER) {
                if (bits) walk(std::true_type{}, std::false_type{}); else walk(std::false_type{}, std::false_type{});
            } else {
                if (bits) walk(std::false_type{}, std::true_type{}); else walk(std::false_type{}, std::false_type{});
            }
        }
        __syncthreads();
        if (!SCATTER) {
            for (int t = threadIdx.x; t < span; t += THREADS) {
                const uint32_t cnt = lds[t];
                gstore(bv.rel + (size_t)chunk * tiles + lo + t, cnt ? gatomic_add(bv.tile_count + lo + t, cnt) : 0u);
            }
            __syncthreads();
        }
    }
}

// tie_inv[tie_index[i]] = i  (tie_index is a permutation; entries out of range are ignored)
__global__ void invert_tie_index_kernel(int n, const int32_t* __restrict__ tie_index, uint32_t* __restrict__ tie_inv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k = tie_index[i];
    if ((unsigned)k < (unsigned)n) tie_inv[k] = (uint32_t)i;
}

// one workgroup per view.  The running total is kept in 64 bits: a view's instance count can pass 2^32 (millions of
// screen-filling splats x thousands of tiles), and a wrapped 32-bit sum could land below max_instances and let the
// scatter pass write through wrapped ranges.  Ranges are clamped to max_instances (no cursor can pass the buffers
// even if a later stage ignored the flag); counters[0] saturates at 0xffffffff.
//
// The same pass knows every list's length, so it also does what two small launches did until round 3 (order_count,
// order_scan): it counts the compositor's work items per (XCD stream, length class) into order_state, and the LAST
// workgroup to finish (agent-scope ticket behind a fence) turns the counts into the streams' write cursors.
__global__ __launch_bounds__(1024) void tile_scan_kernel(const BinView* __restrict__ views, int tiles,
                                                         uint32_t max_instances, int grid_x,
                                                         uint32_t* __restrict__ order_state, int skip_empty,
                                                         uint32_t* __restrict__ host_status) {
    __shared__ unsigned long long wave_tot[1024 / WAVE];
    __shared__ unsigned long long carry_s;
    __shared__ uint32_t hist[ORDER_BINS];
    __shared__ uint32_t last_s;
    const BinView& bv = views[blockIdx.x];
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    if (threadIdx.x == 0) carry_s = 0;
    if (threadIdx.x < ORDER_BINS) hist[threadIdx.x] = 0;
    __syncthreads();
    for (int base = 0; base < tiles; base += 1024) {
        const int idx = base + threadIdx.x;
        const unsigned long long v = idx < tiles ? (unsigned long long)bv.tile_count[idx] : 0ull;
        unsigned long long s = v;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const unsigned long long t = __shfl_up(s, d, WAVE);
            if (lane >= d) s += t;
        }
        if (lane == WAVE - 1) wave_tot[wid] = s;
        __syncthreads();
        unsigned long long wave_prefix = 0;
        for (int w = 0; w < wid; ++w) wave_prefix += wave_tot[w];
        const unsigned long long carry = carry_s;
        const unsigned long long excl = carry + wave_prefix + s - v;
        const unsigned long long cap = max_instances;
        if (idx < tiles) {
            const uint2 r = make_uint2((uint32_t)min(excl, cap), (uint32_t)min(excl + v, cap));
            bv.ranges[idx] = r;
            // (a layered call gives empty lists no work item: their pixels are pre-filled, layer_mask_fill_kernel)
            if (order_state && !(skip_empty && r.y == r.x))
                atomicAdd(&hist[xcd_of_tile(idx, grid_x) * ORDER_CLASSES_USED + coarse_class(r.y - r.x)], ITEMS_PER_TILE);
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wave_prefix + s;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const uint32_t total = (uint32_t)min(carry_s, 0xffffffffull);
        const uint32_t over = carry_s > (unsigned long long)max_instances ? 1u : 0u;
        bv.counters[0] = total;
        bv.counters[1] = over;
        // early status (pgr_forward_posed_early_status): the two words are FINAL here -- nothing after the scan changes them --
        // so they go straight to the caller's pinned host memory, and the event recorded behind this kernel lets the host
        // decide about an overflow while scatter, sort and compositor still run
        if (host_status) {
            host_status[2 * blockIdx.x] = total;
            host_status[2 * blockIdx.x + 1] = over;
            __threadfence_system();
        }
    }
    if (!order_state) return;
    if (gridDim.x == 1) {
        // one view, one workgroup: the counts are this workgroup's own LDS words -- no agent-scope atomics, ticket or fences
        // (two __threadfence() are ~7 us of the ~14 us this kernel takes in a single-view call, where every launch is latency)
        __syncthreads();
        if (threadIdx.x < NUM_XCD) {
            uint32_t acc = 0;
            for (int c = 0; c < ORDER_CLASSES_USED; ++c) {
                gstore(order_state + threadIdx.x * ORDER_CLASSES_USED + c, acc);
                acc += hist[threadIdx.x * ORDER_CLASSES_USED + c];
            }
        }
        return;
    }
    if (threadIdx.x < ORDER_BINS && hist[threadIdx.x]) gatomic_add(order_state + threadIdx.x, hist[threadIdx.x]);
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last_s = gatomic_add(order_state + ORDER_DONE_WORD, 1u) == gridDim.x - 1u ? 1u : 0u;
    __syncthreads();
    if (!last_s) return;
    __threadfence();
    if (threadIdx.x < NUM_XCD) {             // one thread per stream: exclusive prefix over the classes = write cursors
        uint32_t acc = 0;
        for (int c = 0; c < ORDER_CLASSES_USED; ++c) {
            const uint32_t v = gatomic_load(order_state + threadIdx.x * ORDER_CLASSES_USED + c);
            gstore(order_state + threadIdx.x * ORDER_CLASSES_USED + c, acc);
            acc += v;
        }
    }
}

// ---- per-tile sort ---------------------------------------------------------------------------
// Keys are 64-bit (depth bits << 32 | Gaussian index): unique, so any comparison sort yields THE order.
//
// LDS merge sort: thread t owns E consecutive keys; (1) sorts them in registers with a bitonic network,
// (2) log2(THREADS) merge rounds: the two sorted runs a thread's E outputs fall in are co-ranked by a
// binary search (merge path), E outputs are merged into registers, and after a barrier written back
// over the same LDS buffer.  Work ~ n (log2 n) compares, all in LDS/registers; the list is read
// from HBM once (8 B/entry) and the sorted indices written once (4 B/entry).

constexpr uint64_t KEY_INF = ~0ull;

// low word of a sort key -> position; inv == NULL: they are the same
__device__ __forceinline__ uint32_t key_to_pos(const uint32_t* __restrict__ inv, uint32_t tk) { return inv ? gload(inv + tk) : tk; }

template <int E>
__device__ __forceinline__ void register_sort(uint64_t (&r)[E]) {
#pragma unroll
    for (int k = 2; k <= E; k <<= 1) {
#pragma unroll
        for (int i = 0; i < E; ++i) {           // flip step
            const int j = i ^ (k - 1);
            if (j > i) {
                const uint64_t a = r[i], b = r[j];
                const bool sw = a > b;
                r[i] = sw ? b : a;
                r[j] = sw ? a : b;
            }
        }
#pragma unroll
        for (int s = k >> 2; s >= 1; s >>= 1) {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int j = i ^ s;
                if (j > i) {
                    const uint64_t a = r[i], b = r[j];
                    const bool sw = a > b;
                    r[i] = sw ? b : a;
                    r[j] = sw ? a : b;
                }
            }
        }
    }
}

// LDS index padding: one spare key after every E keys.  Thread t touches keys t*E.. : without the pad
// the lane stride is E*8 B (128 B at E = 16), i.e. every lane of a ds_read/write_b64 on the same bank
// pair; with it the stride is (E+1)*8 B and the 32 lanes of a half-wave land on 32 distinct bank pairs.
template <int E>
__device__ __forceinline__ int pad_idx(int i) {
    return i + i / E;   // E is a power of two: a shift
}

// Fused semantic pass support: *obj_last = 1 + position of the LAST entry with index >= n_env (an object's Gaussian)
// in the tile's sorted list, 0 if there is none (the word is zero-filled per batch).  The compositor walks a tile's
// list beyond the saturation of its scene pixels only up to there.  `get(i)` returns the i-th sorted index.
template <int THREADS, typename Get>
__device__ __forceinline__ void mark_last_object(Get get, int n, int n_env, uint32_t* __restrict__ obj_last,
                                                 uint32_t pos_offset = 0) {
    uint32_t best = 0;
    for (int i = threadIdx.x; i < n; i += THREADS)
        if ((int)get(i) >= n_env) best = pos_offset + (uint32_t)i + 1u;
#pragma unroll
    for (int m = 1; m < WAVE; m <<= 1) best = max(best, (uint32_t)__shfl_xor((int)best, m));
    if ((threadIdx.x & (WAVE - 1)) == 0 && best) gatomic_max(obj_last, best);
}

// Sorts n <= THREADS*E keys of `bucket` (global, (depth,idx) pairs) into out[] (indices only).
// skeys must hold THREADS*(E+1) keys.
template <int THREADS, int E>
__device__ __forceinline__ void merge_sort_tile(uint64_t* __restrict__ skeys, const uint2* __restrict__ bucket,
                                                uint32_t* __restrict__ out, int n, uint64_t* keys_out = nullptr,
                                                int n_env = -1, uint32_t* __restrict__ obj_last = nullptr,
                                                uint32_t pos_offset = 0, const int32_t* __restrict__ tie = nullptr,
                                                const uint32_t* __restrict__ inv = nullptr) {
    const int t = threadIdx.x;
    uint64_t r[E];
    // the list is unordered, so WHICH keys a thread starts with is free: take them coalesced
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * THREADS + t;
        uint64_t k = KEY_INF;
        if (i < n) {
            const uint2 v = gload(bucket + i);
            k = ((uint64_t)v.x << 32) | (tie ? (uint32_t)gload(tie + v.y) : v.y);
        }
        r[e] = k;
    }
    register_sort<E>(r);
#pragma unroll
    for (int e = 0; e < E; ++e) skeys[t * (E + 1) + e] = r[e];
    __syncthreads();

    const int total = THREADS * E;
    for (int run = E; run < total; run <<= 1) {
        const int o_glob = t * E;                 // first output position of this thread
        const int a0 = o_glob & ~(2 * run - 1), b0 = a0 + run;   // run is a power of two
        const int o = o_glob - a0;                // output offset inside the merged pair
        // co-rank: largest i in [lo,hi] with A[i-1] <= B[o-i]  (A wins ties; keys are unique anyway)
        int lo = o > run ? o - run : 0, hi = o < run ? o : run;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;   // candidate count taken from A
            const uint64_t a = skeys[pad_idx<E>(a0 + mid - 1)];
            const uint64_t b = skeys[pad_idx<E>(b0 + o - mid)];   // o-mid < run guaranteed by lo bound
            if (a <= b) lo = mid; else hi = mid - 1;
        }
        int i = lo, j = o - lo;
        uint64_t av = i < run ? skeys[pad_idx<E>(a0 + i)] : KEY_INF;
        uint64_t bv = j < run ? skeys[pad_idx<E>(b0 + j)] : KEY_INF;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool takeA = (j >= run) || (i < run && av <= bv);
            r[e] = takeA ? av : bv;
            if (takeA) { ++i; av = i < run ? skeys[pad_idx<E>(a0 + i)] : KEY_INF; }
            else       { ++j; bv = j < run ? skeys[pad_idx<E>(b0 + j)] : KEY_INF; }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) skeys[t * (E + 1) + e] = r[e];
        __syncthreads();
    }
    if (keys_out) {
        for (int i = t; i < n; i += THREADS) gstore(keys_out + i, skeys[pad_idx<E>(i)]);
    } else {
        for (int i = t; i < n; i += THREADS) gstore(out + i, key_to_pos(inv, (uint32_t)skeys[pad_idx<E>(i)]));
        if (n_env >= 0)
            mark_last_object<THREADS>([&](int i) { return key_to_pos(inv, (uint32_t)skeys[pad_idx<E>(i)]); }, n, n_env, obj_last,
                                      pos_offset);
    }
}

// ---- bucket sort (interpolation sort) of one tile list in LDS ---------------------------------------------------
// A tile's depths span a narrow interval that its Gaussians fill fairly evenly, so a monotone map of the depth bits
// onto NB = capacity buckets puts about one key in each: one LDS atomic per key builds the histogram AND hands the
// key its arrival slot inside the bucket, one scan turns counts into bucket starts, one LDS store parks the key, and
// the exact order inside a bucket (same 64-bit key compare as the merge sort, so the result is the same permutation)
// is a count of the smaller keys among the bucket's few members.  Lists whose depths pile up (sum of squared bucket
// counts > 8 n) are rejected and take the merge sort; the result never depends on which path ran.
//
// Round 5 (position-owned ranking; rounds 1-4 ranked a key in the thread that had loaded it, stored the sorted
// indices into an LDS image and read that back: eight workgroup barriers, C3 sort stage 0.0272 ms per view, C5 0.0817):
//   * Once the keys are parked (bucket start + arrival slot) the image is sorted by bucket, and the exact place of a key
//     is decided by the thread that owns its PARKED POSITION: position i is read back coalesced, its bucket recomputed
//     from its depth bits (4 VALU), the bucket's extent read from the counter word (the lanes of a wave read consecutive
//     or equal words), the members of a multi-key bucket are the position's own neighbours.  The random LDS reads of the
//     ranking and the random store + read-back of an index image are gone (23 % of the sort's LDS-array cycles at a
//     2.1 - 2.9-fold conflict factor: scripts/sim/sort_bank_conflicts.py); the sorted indices go straight to global
//     memory from position order (a wave's 64 stores fall into the 64..130-entry window its positions map to: whole
//     lines).  What stays random -- the histogram atomic, the start look-up and the 8-byte park -- is random by
//     construction (a hash of the depth): no bucket swizzle or key-to-lane assignment changes its 2.7-fold conflict
//     factor (same simulation).
//   * Wave-level partial results (min, max, totals, squares) go to per-wave words instead of atomics on shared words,
//     which need no zero-fill barrier in front: five workgroup barriers per list.
//   Measured (A/B on one box, profiles/r05_sort_ab.txt): C3 sort stage 0.0272 -> 0.0240 ms per view, C5 0.0817 -> 0.0738.

constexpr uint32_t BUCKET_SQ_LIMIT = 8;

// Sorts n <= THREADS*E keys of `bucket` (global, (depth,idx) pairs) into out[] (indices) -- same contract as
// merge_sort_tile without keys_out.  NB = number of buckets (a multiple of THREADS; THREADS*E = one per key of
// capacity); KEYS = the most keys a call may hold (THREADS*E by default).  lds: KEYS*8 + NB*4 + SORT_MISC_BYTES.
// Returns false (LDS free for reuse, nothing written) when the list is rejected.
constexpr int SORT_MISC_BYTES = 256;

template <int THREADS, int E, int NB = THREADS * E, int KEYS = THREADS * E>
__device__ __forceinline__ bool bucket_sort_tile(unsigned char* __restrict__ lds, const uint2* __restrict__ bucket,
                                                     uint32_t* __restrict__ out, int n, int n_env,
                                                     uint32_t* __restrict__ obj_last, uint32_t pos_offset = 0,
                                                     const int32_t* __restrict__ tie = nullptr) {
    constexpr int CAP = THREADS * E, WAVES = THREADS / WAVE, CH = NB / (WAVES * WAVE);   // 64-bucket chunks per wave
    static_assert(NB % (WAVES * WAVE) == 0, "bucket count");
    static_assert(4 * WAVES * 4 <= SORT_MISC_BYTES, "per-wave words must fit the bytes the callers reserve behind the counters");
    static_assert(KEYS <= CAP, "key image");
    uint64_t* s_keys = reinterpret_cast<uint64_t*>(lds);                       // [KEYS] (n <= KEYS <= CAP)
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(lds + (size_t)KEYS * 8);    // [NB] counts -> start | members << 16
    uint32_t* s_wmin = s_hist + NB;                                            // [WAVES] per-wave minimum of the depth bits
    uint32_t* s_wmax = s_wmin + WAVES;                                         // [WAVES] maximum
    uint32_t* s_wtot = s_wmax + WAVES;                                         // [WAVES] keys in the wave's buckets
    uint32_t* s_wsq = s_wtot + WAVES;                                          // [WAVES] sum of squared bucket counts
    const int t = threadIdx.x, lane = t & (WAVE - 1), wave = t / WAVE;

    // all E loads in flight at once, index clamped into the list (n >= 1)
    uint32_t d[E], id[E];
    uint32_t dmin = 0xffffffffu, dmax = 0u;
    {
        uint2 v[E];
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = gload(bucket + min(e * THREADS + t, n - 1));
        for (int i = t; i < NB; i += THREADS) s_hist[i] = 0u;              // (beside the loads)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const bool in = e * THREADS + t < n;
            d[e] = in ? v[e].x : 0xffffffffu; id[e] = in ? v[e].y : 0xffffffffu;
            dmin = min(dmin, d[e]); dmax = max(dmax, in ? v[e].x : 0u);
        }
    }
    dmin = ~wave_inclusive_max(~dmin);          // min as max of the complement (0 is the prefix-max identity)
    dmax = wave_inclusive_max(dmax);
    if (lane == WAVE - 1) { s_wmin[wave] = dmin; s_wmax[wave] = dmax; }
    __syncthreads();                            // 1: counters zero, per-wave extremes visible
    uint32_t mn = 0xffffffffu, mx = 0u;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { mn = min(mn, s_wmin[w]); mx = max(mx, s_wmax[w]); }
    const float scale = (float)NB / ((float)(mx - mn) + 1.0f);
    // monotone in d: uint->float conversion, multiplication by a positive constant, truncation and clamp all are
    auto bucket_of = [&](uint32_t depth) { return min((uint32_t)((float)(depth - mn) * scale), (uint32_t)(NB - 1)); };
    uint32_t br[E];      // bucket << 16 | arrival slot inside the bucket
#pragma unroll
    for (int e = 0; e < E; ++e) {
        br[e] = 0u;
        if (e * THREADS + t < n) {
            const uint32_t b = bucket_of(d[e]);
            br[e] = (b << 16) | atomicAdd(&s_hist[b], 1u);
        }
    }
    __syncthreads();                            // 2: histogram complete
    // pass A: every wave owns NB / WAVES consecutive buckets (CH chunks of 64): totals and sum of squares
    const int wbase = wave * (WAVE * CH);
    constexpr bool KEEP_H = E <= 8;             // the counts stay in registers between the passes (16 keys per thread: no room)
    uint32_t h[KEEP_H ? CH : 1];
    uint32_t tot = 0, sq = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const uint32_t hc = s_hist[wbase + c * WAVE + lane];
        if (KEEP_H) h[c] = hc;
        tot += hc; sq += hc * hc;
    }
    tot = wave_inclusive_scan(tot);
    sq = wave_inclusive_scan(sq);
    if (lane == WAVE - 1) { s_wtot[wave] = tot; s_wsq[wave] = sq; }
    __syncthreads();                            // 3: per-wave totals visible
    uint32_t carry = 0, sqsum = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const uint32_t tw = s_wtot[w];
        carry += w < wave ? tw : 0u;
        sqsum += s_wsq[w];
    }
    // average occupancy is n / NB by construction: reject when the squares exceed what an even spread would cost
    if (sqsum > BUCKET_SQ_LIMIT * (uint32_t)n * (uint32_t)((CAP + NB - 1) / NB)) { __syncthreads(); return false; }
    // pass B: exclusive scan of the counts -> bucket start | members << 16
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const uint32_t hc = KEEP_H ? h[c] : s_hist[wbase + c * WAVE + lane];
        const uint32_t incl = wave_inclusive_scan(hc);
        s_hist[wbase + c * WAVE + lane] = (carry + incl - hc) | (hc << 16);
        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
    }
    __syncthreads();                            // 4: bucket starts
#pragma unroll
    for (int e = 0; e < E; ++e)
        if (e * THREADS + t < n)
            s_keys[(s_hist[br[e] >> 16] & 0xffffu) + (br[e] & 0xffffu)] = ((uint64_t)d[e] << 32) | id[e];
    __syncthreads();                            // 5: every key parked in its bucket
    // exact place inside the bucket, by parked position: all E keys of the thread and their buckets' extents in flight
    // at once, then the members of the multi-key buckets (neighbouring positions)
    // (EB positions per round: sixteen 64-bit keys and their extents at once do not fit the open-ended tier's 128 VGPRs)
    constexpr int EB = E < 8 ? E : 8;
    uint32_t best = 0;
#pragma unroll
    for (int e0 = 0; e0 < E; e0 += EB) {
        uint64_t key[EB];
        uint32_t ext[EB];
#pragma unroll
        for (int e = 0; e < EB; ++e) key[e] = s_keys[min((e0 + e) * THREADS + t, n - 1)];
#pragma unroll
        for (int e = 0; e < EB; ++e) ext[e] = s_hist[bucket_of((uint32_t)(key[e] >> 32))];
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int i = (e0 + e) * THREADS + t;
            if (i < n) {
                const uint32_t s0 = ext[e] & 0xffffu, cnt = ext[e] >> 16;
                const uint32_t kd = (uint32_t)(key[e] >> 32), kid = (uint32_t)key[e];
                uint32_t rank = (uint32_t)i - s0;                    // alone in its bucket: in place
                if (cnt > 1u) {
                    rank = 0;
                    uint32_t same = 0;                               // members with this key's depth bits (itself included)
                    constexpr uint32_t RANK_BATCH = 4;               // the first members in ONE LDS round trip
                    uint64_t kb[RANK_BATCH];
#pragma unroll
                    for (uint32_t m = 0; m < RANK_BATCH; ++m) kb[m] = s_keys[s0 + min(m, cnt - 1u)];
#pragma unroll
                    for (uint32_t m = 0; m < RANK_BATCH; ++m) {
                        const bool in = m < cnt;
                        rank += (in && kb[m] < key[e]) ? 1u : 0u;
                        same += (in && (uint32_t)(kb[m] >> 32) == kd) ? 1u : 0u;
                    }
                    for (uint32_t j = s0 + RANK_BATCH; j < s0 + cnt; ++j) {
                        const uint64_t kj = s_keys[j];
                        rank += kj < key[e] ? 1u : 0u;
                        same += (uint32_t)(kj >> 32) == kd ? 1u : 0u;
                    }
                    if (tie && same > 1u) {
                        // exact depth tie between different Gaussians: the CALLER's index decides (a few keys per list)
                        const int32_t mine = gload(tie + kid);
                        rank = 0;
                        for (uint32_t j = s0; j < s0 + cnt; ++j) {
                            const uint64_t kj = s_keys[j];
                            const uint32_t dj = (uint32_t)(kj >> 32);
                            rank += (dj < kd || (dj == kd && (uint32_t)kj != kid && gload(tie + (uint32_t)kj) < mine)) ? 1u : 0u;
                        }
                    }
                }
                const uint32_t f = s0 + rank;
                gstore(out + f, kid);
                if (n_env >= 0 && (int)kid >= n_env) best = max(best, pos_offset + f + 1u);
            }
        }
    }
    if (n_env >= 0) {
        best = wave_inclusive_max(best);
        if (lane == WAVE - 1 && best) gatomic_max(obj_last, best);
    }
    return true;
}

// ---- windowed bucket sort: lists of 8193 .. SORT_WINDOW_MAX keys in an 80 KiB workgroup (round 6) -------------------
// Until round 5 every list beyond 8192 keys went to the open-ended tier: 1024 threads over a 157 KiB image, ONE workgroup
// per CU, whose phases (loads, histogram atomics, scan, park, ranking) nothing overlaps -- VALU 0.35, LDS 0.16, 76 keys/ns
// on C5 against 135 for the two-per-CU 512 x 16 tier, with 30 % of C5's keys.  The LDS image is what holds a CU alone,
// and it only has to hold the keys being RANKED: here a thread keeps the depth bits of its E = 32 keys in registers, the
// histogram (one arrival slot per key, eight bits, four per register) and the scan run once over the whole list, and the
// image is filled and ranked one WINDOW of bucket-aligned positions at a time: window w = the buckets whose start lies in
// [w WK, (w + 1) WK), WK = KEYS - 256, parked at (position - w WK).  A bucket holds at most 255 keys (else the list is
// rejected), so a window's positions end below w WK + KEYS.
//   Counters are HALF WORDS, two buckets per LDS word (a 32-bit atomic adds 1 << 16 (b & 1) to word b >> 1 and returns both
//   halves; no half can overflow into its neighbour below 65 536 keys): 7168 buckets in the 14 KiB that 3584 word counters
//   take -- 1.1 .. 2.2 keys per bucket instead of 2.3 .. 4.4, which is what the in-bucket ranking loops cost (first build,
//   word counters over 3584 buckets: sort stage +6 % on C5 against the open-ended tier, profiles/r06_sort_window_ab.txt).
//   After the scan a half word is its bucket's first position; a bucket's extent = its half and the next one (two
//   consecutive words, one ds_read2_b32).
// Same position-owned ranking and direct stores as bucket_sort_tile: the result is THE order; two workgroups per CU as in
// the 512 x 16 tier, the list's depths read once (the indices again when a key is parked).  Rejected lists (piled-up
// depths) are appended to the open-ended tier's queue, which is launched afterwards.
#ifndef PGR_WINDOW_EB
#define PGR_WINDOW_EB 4
#endif
constexpr int SORT_WINDOW_THREADS = 512;
constexpr int SORT_WINDOW_E = 32;
constexpr int SORT_WINDOW_KEYS = 8192;                      // image: 64 KiB
#ifndef PGR_WINDOW_NB
#define PGR_WINDOW_NB 7168
#endif
constexpr int SORT_WINDOW_BUCKETS = PGR_WINDOW_NB;          // half-word counters: + 14 KiB (+ SORT_MISC_BYTES + window words): 2 per CU
constexpr int SORT_WINDOW_WK = SORT_WINDOW_KEYS - 256;
static_assert(SORT_WINDOW_MAX == 2 * SORT_WINDOW_WK, "composite.hip.h's tier bound: 15 872 keys, two windows");
constexpr int SORT_WINDOW_ROUNDS = (SORT_WINDOW_THREADS * SORT_WINDOW_E + SORT_WINDOW_WK - 1) / SORT_WINDOW_WK;   // by capacity: 3
constexpr size_t SORT_WINDOW_LDS = (size_t)SORT_WINDOW_KEYS * 8 + ((size_t)SORT_WINDOW_BUCKETS / 2 + 2) * 4 + SORT_MISC_BYTES;

template <int THREADS, int E, int NB, int KEYS>
__device__ __forceinline__ bool window_sort_tile(unsigned char* __restrict__ lds, uint32_t* __restrict__ s_win,
                                                 const uint2* __restrict__ bucket, uint32_t* __restrict__ out, int n,
                                                 int n_env, uint32_t* __restrict__ obj_last,
                                                 const int32_t* __restrict__ tie = nullptr) {
    constexpr int WAVES = THREADS / WAVE, NW = NB / 2, CH = NW / (WAVES * WAVE), WK = KEYS - 256, ROUNDS = (THREADS * E + WK - 1) / WK;
    static_assert(NB % (2 * WAVES * WAVE) == 0 && E % 8 == 0 && KEYS % THREADS == 0, "shape");
    static_assert(5 * WAVES * 4 <= SORT_MISC_BYTES, "per-wave words");
    uint64_t* s_keys = reinterpret_cast<uint64_t*>(lds);                       // [KEYS]
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(lds + (size_t)KEYS * 8);    // [NW + 2] two counters per word -> two starts
    uint32_t* s_wmin = s_hist + NW + 2;
    uint32_t* s_wmax = s_wmin + WAVES;
    uint32_t* s_wtot = s_wmax + WAVES;
    uint32_t* s_wsq = s_wtot + WAVES;
    uint32_t* s_wbig = s_wsq + WAVES;                                          // [WAVES] largest bucket of the wave's share
    // s_win[w] = first bucket of window w, s_win[ROUNDS + 1 + w] = its first position; entry ROUNDS closes the last window
    // (opaque per call: the kernel's loop over its lists would otherwise hoist every thread-dependent address and bucket
    // index of this function out of it and keep them on the stack)
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    const int lane = t & (WAVE - 1), wave = t / WAVE;
    const int e_used = (n + THREADS - 1) / THREADS;                            // (uniform) key slots of a thread that hold keys

    // the depth bits stay in registers (E per thread); the indices are fetched again when a key is parked (the list is
    // in the cache hierarchy, and 2 E more live registers spill)
    uint32_t d[E];
    uint32_t dmin = 0xffffffffu, dmax = 0u;
#pragma unroll
    for (int e = 0; e < E; ++e) d[e] = gload(&bucket[min(e * THREADS + t, n - 1)].x);
    for (int i = t; i < NW + 2; i += THREADS) s_hist[i] = 0u;
    // window table: no window opened = (bucket NB, position n); window 0 opens at (bucket 0, position 0)
    if (t < 2 * (ROUNDS + 1)) s_win[t] = (t == 0 || t == ROUNDS + 1) ? 0u : (t <= ROUNDS ? (uint32_t)NB : (uint32_t)n);
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const bool in = e * THREADS + t < n;
        dmin = min(dmin, in ? d[e] : 0xffffffffu); dmax = max(dmax, in ? d[e] : 0u);
    }
    dmin = ~wave_inclusive_max(~dmin);
    dmax = wave_inclusive_max(dmax);
    if (lane == WAVE - 1) { s_wmin[wave] = dmin; s_wmax[wave] = dmax; }
    __syncthreads();                            // 1: counters zero, per-wave extremes visible
    uint32_t mn = 0xffffffffu, mx = 0u;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { mn = min(mn, s_wmin[w]); mx = max(mx, s_wmax[w]); }
    const float scale = (float)NB / ((float)(mx - mn) + 1.0f);
    auto bucket_of = [&](uint32_t depth) { return min((uint32_t)((float)(depth - mn) * scale), (uint32_t)(NB - 1)); };
    // Eight atomics of a thread in flight at a time: a key slot beyond the list adds 0 to a word of the thread's own
    // (no exec-masked round trip per key, no pile of same-address atomics from the padding)
    uint32_t sl[E / 4];                         // arrival slot inside the bucket, 8 bits per key (a bucket of > 255 keys rejects the list)
#pragma unroll
    for (int e0 = 0; e0 < E; e0 += 8) {
        sl[e0 / 4] = 0u; sl[e0 / 4 + 1] = 0u;
        if (e0 < e_used) {
            uint32_t old[8], sh[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool in = (e0 + e) * THREADS + t < n;
                const uint32_t b = bucket_of(d[e0 + e]);
                sh[e] = (b & 1u) << 4;
                old[e] = atomicAdd(&s_hist[in ? (b >> 1) : (uint32_t)t], in ? (1u << sh[e]) : 0u);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) sl[(e0 + e) >> 2] |= ((old[e] >> sh[e]) & 0xffu) << (8 * (e & 3));
        }
    }
    __syncthreads();                            // 2: histogram complete
    const int wbase = wave * (WAVE * CH);
    uint32_t tot = 0, sq = 0, big = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const uint32_t hc = s_hist[wbase + c * WAVE + lane], lo = hc & 0xffffu, hi = hc >> 16;
        tot += lo + hi; sq += lo * lo + hi * hi; big = max(big, max(lo, hi));
    }
    tot = wave_inclusive_scan(tot);
    sq = wave_inclusive_scan(sq);
    big = wave_inclusive_max(big);
    if (lane == WAVE - 1) { s_wtot[wave] = tot; s_wsq[wave] = sq; s_wbig[wave] = big; }
    __syncthreads();                            // 3: per-wave totals visible
    uint32_t carry = 0, sqsum = 0, biggest = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const uint32_t tw = s_wtot[w];
        carry += w < wave ? tw : 0u;
        sqsum += s_wsq[w];
        biggest = max(biggest, s_wbig[w]);
    }
    if (biggest > 255u || sqsum > BUCKET_SQ_LIMIT * (uint32_t)n * (uint32_t)((THREADS * E + NB - 1) / NB)) { __syncthreads(); return false; }
    // exclusive scan of the counts -> every half word = its bucket's first position; the bucket whose start is the first at
    // or beyond w WK opens window w (exactly one bucket per window that exists)
    auto opens = [&](uint32_t b, uint32_t start, uint32_t next) {
        // w WK in (start, next]: the NEXT bucket's start is the first at or beyond w WK (a bucket holds < WK keys: one w)
        const uint32_t w = start / (uint32_t)WK + 1u;
        if (w * (uint32_t)WK <= next && w <= (uint32_t)ROUNDS) { s_win[w] = b + 1u; s_win[ROUNDS + 1 + w] = next; }
    };
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const uint32_t word = (uint32_t)(wbase + c * WAVE + lane);
        const uint32_t hc = s_hist[word], lo = hc & 0xffffu, hi = hc >> 16;
        const uint32_t incl = wave_inclusive_scan(lo + hi);
        const uint32_t s_lo = carry + incl - (lo + hi), s_hi = s_lo + lo;
        s_hist[word] = s_lo | (s_hi << 16);
        opens(2u * word, s_lo, s_hi);
        opens(2u * word + 1u, s_hi, s_hi + hi);
        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
    }
    if (t == THREADS - 1) s_hist[NW] = (uint32_t)n | ((uint32_t)n << 16);      // closes the last bucket
    __syncthreads();                            // 4: bucket starts and window table
    uint32_t best = 0;
    for (int w = 0; w < ROUNDS; ++w) {
        const uint32_t b_lo = s_win[w], b_hi = s_win[w + 1];
        const uint32_t p_lo = s_win[ROUNDS + 1 + w], p_hi = min(s_win[ROUNDS + 2 + w], (uint32_t)n);
        if (b_lo >= (uint32_t)NB || p_lo >= (uint32_t)n) break;
        const uint32_t shift = (uint32_t)w * (uint32_t)WK;
        // (opaque copies per round: everything below that depends only on d[] and t is loop-invariant, and the compiler
        // would keep the 32 buckets and 32 load addresses of a thread live across the rounds -- 100 registers to the stack)
        int tt = t;
        asm volatile("" : "+v"(tt));
        constexpr int PBATCH = 8;               // index loads and bucket-start look-ups in flight at a time
#pragma unroll
        for (int e0 = 0; e0 < E; e0 += PBATCH) {
            if (e0 >= e_used) break;
            uint32_t idv[PBATCH], st[PBATCH];
            bool mine[PBATCH];
#pragma unroll
            for (int e = 0; e < PBATCH; ++e) {
                asm volatile("" : "+v"(d[e0 + e]));
                const uint32_t b = bucket_of(d[e0 + e]);
                mine[e] = (e0 + e) * THREADS + tt < n && b >= b_lo && b < b_hi;
                idv[e] = 0u; st[e] = 0u;
                if (mine[e]) { idv[e] = gload(&bucket[(e0 + e) * THREADS + tt].y); st[e] = (s_hist[b >> 1] >> ((b & 1u) << 4)) & 0xffffu; }
            }
#pragma unroll
            for (int e = 0; e < PBATCH; ++e)
                if (mine[e])
                    s_keys[st[e] + ((sl[(e0 + e) >> 2] >> (8 * ((e0 + e) & 3))) & 0xffu) - shift] = ((uint64_t)d[e0 + e] << 32) | idv[e];
        }
        __syncthreads();                        // the window's keys parked in their buckets
        constexpr int EB = PGR_WINDOW_EB, PER = KEYS / THREADS;
#pragma unroll
        for (int e0 = 0; e0 < PER; e0 += EB) {
            uint64_t key[EB];
            uint32_t x0[EB], x1[EB];
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const uint32_t i = shift + (uint32_t)((e0 + e) * THREADS + t);
                key[e] = s_keys[min(max(i, p_lo), p_hi - 1u) - shift];
            }
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const uint32_t b = bucket_of((uint32_t)(key[e] >> 32));
                x0[e] = s_hist[b >> 1]; x1[e] = s_hist[(b >> 1) + 1];
            }
#pragma unroll
            for (int e = 0; e < EB; ++e) {
                const uint32_t i = shift + (uint32_t)((e0 + e) * THREADS + t);
                if (i >= p_lo && i < p_hi) {
                    const uint32_t kd = (uint32_t)(key[e] >> 32), kid = (uint32_t)key[e];
                    const bool odd = bucket_of(kd) & 1u;
                    const uint32_t s0 = odd ? x0[e] >> 16 : x0[e] & 0xffffu, cnt = (odd ? x1[e] & 0xffffu : x0[e] >> 16) - s0;
                    uint32_t rank = i - s0;
                    if (cnt > 1u) {
                        rank = 0;
                        uint32_t same = 0;
                        constexpr uint32_t RANK_BATCH = 4;
                        uint64_t kb[RANK_BATCH];
#pragma unroll
                        for (uint32_t m = 0; m < RANK_BATCH; ++m) kb[m] = s_keys[s0 - shift + min(m, cnt - 1u)];
#pragma unroll
                        for (uint32_t m = 0; m < RANK_BATCH; ++m) {
                            const bool in = m < cnt;
                            rank += (in && kb[m] < key[e]) ? 1u : 0u;
                            same += (in && (uint32_t)(kb[m] >> 32) == kd) ? 1u : 0u;
                        }
                        for (uint32_t j = s0 + RANK_BATCH; j < s0 + cnt; ++j) {
                            const uint64_t kj = s_keys[j - shift];
                            rank += kj < key[e] ? 1u : 0u;
                            same += (uint32_t)(kj >> 32) == kd ? 1u : 0u;
                        }
                        if (tie && same > 1u) {
                            const int32_t mine = gload(tie + kid);
                            rank = 0;
                            for (uint32_t j = s0; j < s0 + cnt; ++j) {
                                const uint64_t kj = s_keys[j - shift];
                                const uint32_t dj = (uint32_t)(kj >> 32);
                                rank += (dj < kd || (dj == kd && (uint32_t)kj != kid && gload(tie + (uint32_t)kj) < mine)) ? 1u : 0u;
                            }
                        }
                    }
                    const uint32_t f = s0 + rank;
                    gstore(out + f, kid);
                    if (n_env >= 0 && (int)kid >= n_env) best = max(best, f + 1u);
                }
            }
        }
        __syncthreads();                        // the image is free for the next window
    }
    if (n_env >= 0) {
        best = wave_inclusive_max(best);
        if (lane == WAVE - 1 && best) gatomic_max(obj_last, best);
    }
    return true;
}

// Lists beyond one LDS sort (> CAP keys): ONE counting-sort pass by coarse depth bucket through the alt buffer
// (L2-resident), cut at the first bucket start at or after every multiple of CAP / 2 -- with no bucket larger than
// CAP / 2 every segment holds < CAP keys, and all of its depths precede the next segment's -- then each segment is
// sorted in LDS like a list of its own and written at its offset.  Returns false (nothing written to out) when a
// coarse bucket piles up or the list needs more than PART_MAX_SEGMENTS segments; the caller then merges LDS-sorted
// chunks through L2.
constexpr int SORT_LARGE_THREADS = 1024;
constexpr int SORT_LARGE_MAX = SORT_LARGE_THREADS * 16;     // 16384: register capacity of a 1024-thread workgroup at 16 keys each
// The open-ended tier sorts up to SORT_LARGE_KEYS keys in one LDS image: 8 B per key + 8192 buckets = 157 KiB of the CU's
// 160.  (Round 3: 4096 buckets under 16384 keys put 2-4 keys into a bucket, and the serial ranking loop inside the buckets
// was 45 % of such a list's cycles -- 54 % of the 5 M-Gaussian scene's keys sit in these lists.  Twice the buckets for
// 2 % fewer keys; lists of 16001..16384 keys take the partition path with the longer ones.)
constexpr int SORT_LARGE_KEYS = 16000;
constexpr int SORT_LARGE_BUCKETS = 8192;
constexpr int PART_BUCKETS = 4096;
constexpr int PART_MAX_SEGMENTS = 63;

template <int THREADS, int E>
__device__ __forceinline__ bool partition_sort_long(unsigned char* __restrict__ lds, uint32_t* __restrict__ s_cut,
                                                    const uint2* __restrict__ bucket, uint2* __restrict__ alt,
                                                    uint32_t* __restrict__ out, int n, int n_env,
                                                    uint32_t* __restrict__ obj_last, const int32_t* __restrict__ tie,
                                                    const uint32_t* __restrict__ inv) {
    constexpr int CAP = SORT_LARGE_KEYS, HALF = CAP / 2, WAVES = THREADS / WAVE, CH = PART_BUCKETS / (WAVES * WAVE);
    static_assert(PART_BUCKETS % (WAVES * WAVE) == 0, "bucket count");
    uint32_t* s_hist = reinterpret_cast<uint32_t*>(lds);         // [PART_BUCKETS] counts -> starts -> cursors
    uint32_t* s_misc = s_hist + PART_BUCKETS;                    // [0] min [1] max [2] largest bucket [4..] wave totals
    const int t = threadIdx.x, lane = t & (WAVE - 1), wave = t / WAVE;
    if ((n + HALF - 1) / HALF > PART_MAX_SEGMENTS) return false;
    uint32_t dmin = 0xffffffffu, dmax = 0u;
    // The three passes over the list read it from global memory (L2) with PB loads of a thread in flight at once, index
    // clamped into the list: a `for (i < n) gload` loop waits for every load before it issues the next one (20..100 serial
    // round trips per pass and thread on a 20..100 k-key list, with one workgroup per CU to cover them).
    constexpr int PB = 8;
    for (int i0 = t; i0 < n; i0 += PB * THREADS) {
        uint32_t dv[PB];
#pragma unroll
        for (int k = 0; k < PB; ++k) dv[k] = gload(bucket + min(i0 + k * THREADS, n - 1)).x;     // (a clamped repeat changes no min / max)
#pragma unroll
        for (int k = 0; k < PB; ++k) { dmin = min(dmin, dv[k]); dmax = max(dmax, dv[k]); }
    }
    for (int i = t; i < PART_BUCKETS; i += THREADS) s_hist[i] = 0u;
    if (t < 4 + WAVES) s_misc[t] = t == 0 ? 0xffffffffu : 0u;
    if (t < PART_MAX_SEGMENTS + 2) s_cut[t] = (uint32_t)n;      // segment g starts at s_cut[g]; n = not opened
    __syncthreads();
#pragma unroll
    for (int m = 1; m < WAVE; m <<= 1) {
        dmin = min(dmin, (uint32_t)__shfl_xor((int)dmin, m));
        dmax = max(dmax, (uint32_t)__shfl_xor((int)dmax, m));
    }
    if (lane == 0) { atomicMin(&s_misc[0], dmin); atomicMax(&s_misc[1], dmax); }
    __syncthreads();
    const uint32_t mn = s_misc[0];
    const float scale = (float)PART_BUCKETS / ((float)(s_misc[1] - mn) + 1.0f);
    auto coarse = [&](uint32_t d) { return min((uint32_t)((float)(d - mn) * scale), (uint32_t)(PART_BUCKETS - 1)); };
    for (int i0 = t; i0 < n; i0 += PB * THREADS) {
        uint32_t dv[PB];
#pragma unroll
        for (int k = 0; k < PB; ++k) dv[k] = gload(bucket + min(i0 + k * THREADS, n - 1)).x;
#pragma unroll
        for (int k = 0; k < PB; ++k)
            if (i0 + k * THREADS < n) atomicAdd(&s_hist[coarse(dv[k])], 1u);
    }
    __syncthreads();
    const int wbase = wave * (WAVE * CH);
    uint32_t tot = 0, big = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const uint32_t h = s_hist[wbase + c * WAVE + lane];
        tot += h; big = max(big, h);
    }
#pragma unroll
    for (int m = 1; m < WAVE; m <<= 1) {
        tot += (uint32_t)__shfl_xor((int)tot, m);
        big = max(big, (uint32_t)__shfl_xor((int)big, m));
    }
    if (lane == 0) { s_misc[4 + wave] = tot; atomicMax(&s_misc[2], big); }
    __syncthreads();
    if (s_misc[2] > (uint32_t)HALF) { __syncthreads(); return false; }
    uint32_t carry = 0;
    for (int w = 0; w < wave; ++w) carry += s_misc[4 + w];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const uint32_t h = s_hist[wbase + c * WAVE + lane];
        const uint32_t incl = wave_inclusive_scan(h);
        s_hist[wbase + c * WAVE + lane] = carry + incl - h;
        carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
    }
    __syncthreads();
    // segment g begins at the smallest bucket start >= g * HALF; bucket sizes <= HALF, so every g up to the last one
    // is hit and consecutive cuts are < CAP apart
    for (int b = t; b < PART_BUCKETS; b += THREADS) {
        const uint32_t s0 = s_hist[b];
        const uint32_t g = (s0 + (uint32_t)HALF - 1u) / (uint32_t)HALF;
        const bool first = b == 0 || (s_hist[b - 1] + (uint32_t)HALF - 1u) / (uint32_t)HALF != g;
        if (first && s0 < (uint32_t)n) atomicMin(&s_cut[g], s0);
    }
    __syncthreads();
    // counting-sort scatter through the alt buffer (the starts become the cursors)
    for (int i0 = t; i0 < n; i0 += PB * THREADS) {
        uint2 kv[PB];
#pragma unroll
        for (int k = 0; k < PB; ++k) kv[k] = gload(bucket + min(i0 + k * THREADS, n - 1));
#pragma unroll
        for (int k = 0; k < PB; ++k)
            if (i0 + k * THREADS < n) gstore(alt + atomicAdd(&s_hist[coarse(kv[k].x)], 1u), kv[k]);
    }
    __threadfence_block();
    __syncthreads();
    for (int g = 0; g <= PART_MAX_SEGMENTS; ++g) {
        const uint32_t a = s_cut[g];
        if (a >= (uint32_t)n) break;
        const int n_seg = (int)(s_cut[g + 1] - a);
        if (!bucket_sort_tile<THREADS, E, SORT_LARGE_BUCKETS, SORT_LARGE_KEYS>(lds, alt + a, out + a, n_seg, n_env, obj_last, a, tie))
            merge_sort_tile<THREADS, E>(reinterpret_cast<uint64_t*>(lds), alt + a, out + a, n_seg, nullptr, n_env, obj_last, a,
                                        tie, inv);
        __syncthreads();
    }
    return true;
}

// One merge round over sorted runs of length `run` held in global memory (L2-resident): src -> dst.
// Each thread produces G consecutive outputs per step; co-ranking by binary search as in the LDS rounds.
template <int THREADS, int G>
__device__ __forceinline__ void merge_round_global(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, int n,
                                                   int run) {
    const int segments = (n + G - 1) / G;
    for (int seg = threadIdx.x; seg < segments; seg += THREADS) {
        const int o_glob = seg * G;
        const int a0 = o_glob / (2 * run) * (2 * run), b0 = a0 + run;
        const int lenA = min(run, n - a0), lenB = max(0, min(run, n - b0));
        const int o = o_glob - a0;
        int lo = o > lenB ? o - lenB : 0, hi = o < lenA ? o : lenA;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (src[a0 + mid - 1] <= src[b0 + o - mid]) lo = mid; else hi = mid - 1;
        }
        int i = lo, j = o - lo;
        uint64_t av = i < lenA ? src[a0 + i] : KEY_INF;
        uint64_t bv = j < lenB ? src[b0 + j] : KEY_INF;
        const int cnt = min(G, n - o_glob);
        for (int e = 0; e < cnt; ++e) {
            const bool takeA = (j >= lenB) || (i < lenA && av <= bv);
            dst[o_glob + e] = takeA ? av : bv;
            if (takeA) { ++i; av = i < lenA ? src[a0 + i] : KEY_INF; }
            else       { ++j; bv = j < lenB ? src[b0 + j] : KEY_INF; }
        }
    }
}

// short tier (<= 2048 keys): at most 1024 buckets under the 2048-key image = 20.7 KiB, SEVEN workgroups per CU (one bucket
// per key: 24.8 KiB, six per CU; C3 sort stage 0.0231 -> 0.0223 ms per view, profiles/r05_sort_tier_shapes_ab.txt)
constexpr int SORT_SHORT_BUCKETS = 1024;
constexpr int SORT_T2_BUCKETS = 3584;                   // 4097..8192-key tier: 7 x 512 buckets (8192 x 8 + 3584 x 4 + 256 B = 80 128 B)
constexpr int SORT_SMALL_MAX = SORT_THREADS * 8;       // 2048 keys, 24 KiB of LDS: six workgroups per CU

struct ObjOut { int n_env; uint32_t* last; const int32_t* tie; const uint32_t* inv; };

// q = sort queue entry (view * tiles + tile, first instance, keys, 0): order_scatter_kernel (composite.hip.h)
__device__ __forceinline__ void sort_item(const BinView* __restrict__ views, int tiles, const uint4 q,
                                          const uint2*& bucket, uint32_t*& out, int& n, ObjOut& oo,
                                          uint64_t** alt = nullptr) {
    const uint32_t view = q.x / (uint32_t)tiles;
    const uint32_t tile = q.x - view * (uint32_t)tiles;
    const BinView& bv = views[view];
    n = (int)q.z;
    bucket = bv.bucket + q.y;
    out = bv.gauss_sorted + q.y;
    oo = ObjOut{bv.n_env, bv.obj_last + tile, bv.tie_index, bv.tie_inv};
    if (alt) *alt = bv.alt + q.y;
}

// q = SEGMENT queue entry (view * tiles + tile, first instance of the list, keys of the segment, first position of the
// segment inside the list): tile_partition_kernel.  The segment's keys sit in the list's alt buffer, depth-partitioned; its
// sorted indices go to the same positions of gauss_sorted.  `pos` = the offset the object marker is counted from.
__device__ __forceinline__ void sort_segment(const BinView* __restrict__ views, int tiles, const uint4 q,
                                             const uint2*& bucket, uint32_t*& out, int& n, ObjOut& oo, uint32_t& pos) {
    const uint32_t view = q.x / (uint32_t)tiles;
    const uint32_t tile = q.x - view * (uint32_t)tiles;
    const BinView& bv = views[view];
    n = (int)q.z;
    pos = q.w;
    bucket = reinterpret_cast<const uint2*>(bv.alt) + q.y + q.w;
    out = bv.gauss_sorted + q.y + q.w;
    oo = ObjOut{bv.n_env, bv.obj_last + tile, bv.tie_index, bv.tie_inv};
}

// grid = n_views * tiles workgroups of 256 (upper bound of the queue's length); lists of 1..2048 entries
__global__ __launch_bounds__(SORT_THREADS) void tile_sort_kernel(const BinView* __restrict__ views, int tiles,
                                                                 const uint4* __restrict__ queue,
                                                                 const uint32_t* __restrict__ n_queue) {
    constexpr int NBS = SORT_SHORT_BUCKETS;     // at most one bucket per key of capacity
    constexpr int NB2 = NBS < SORT_THREADS * 2 ? NBS : SORT_THREADS * 2, NB4 = NBS < SORT_THREADS * 4 ? NBS : SORT_THREADS * 4;
    constexpr int NB8 = NBS < SORT_THREADS * 8 ? NBS : SORT_THREADS * 8;
    static_assert(SORT_THREADS * 9 * 8 <= SORT_THREADS * 8 * 8 + NB8 * 4 + SORT_MISC_BYTES, "the merge sort's padded keys must fit");
    __shared__ __attribute__((aligned(16))) unsigned char lds[SORT_THREADS * 8 * 8 + NB8 * 4 + SORT_MISC_BYTES];
    uint64_t* skeys = reinterpret_cast<uint64_t*>(lds);
    if (blockIdx.x >= *n_queue) return;
    const uint2* bucket; uint32_t* out; int n; ObjOut oo;
    sort_item(views, tiles, queue[blockIdx.x], bucket, out, n, oo);
    if (n <= SORT_THREADS * 2) {
        if (!bucket_sort_tile<SORT_THREADS, 2, NB2>(lds, bucket, out, n, oo.n_env, oo.last, 0u, oo.tie))
            merge_sort_tile<SORT_THREADS, 2>(skeys, bucket, out, n, nullptr, oo.n_env, oo.last, 0u, oo.tie, oo.inv);
    } else if (n <= SORT_THREADS * 4) {
        if (!bucket_sort_tile<SORT_THREADS, 4, NB4>(lds, bucket, out, n, oo.n_env, oo.last, 0u, oo.tie))
            merge_sort_tile<SORT_THREADS, 4>(skeys, bucket, out, n, nullptr, oo.n_env, oo.last, 0u, oo.tie, oo.inv);
    } else {
        if (!bucket_sort_tile<SORT_THREADS, 8, NB8>(lds, bucket, out, n, oo.n_env, oo.last, 0u, oo.tie))
            merge_sort_tile<SORT_THREADS, 8>(skeys, bucket, out, n, nullptr, oo.n_env, oo.last, 0u, oo.tie, oo.inv);
    }
}

#ifndef PGR_T1_WAVES
#define PGR_T1_WAVES 6      // the 512-thread tier: three workgroups per CU (48 KiB of LDS each) need six waves per SIMD
#endif
// Lists longer than 2048: workgroups stride over the queue of their tier.  One launch per tier (THREADS, E):
//   (512, 8)    2049..4096 keys, one bucket per key, 48 KiB of LDS: three workgroups per CU;
//   (512, 16)   4097..8192 keys over SORT_T2_BUCKETS = 3584 buckets, 80 KiB: two per CU (round 5; rounds 2-4 used 1024 x 8
//               with one bucket per key, 96 KiB, one per CU: C3 sort stage 0.0237 -> 0.0229 ms per view, C5 0.0744 ->
//               0.0708, profiles/r05_sort_tier_shapes_ab.txt; the same treatment of the 2049..4096 tier returned nothing);
//   (1024, 16)  LAST, open-ended: up to 16000 keys over 8192 buckets (157 KiB), anything longer depth-partitioned into
//               LDS-sized segments or, failing that, merge-sorted 16384-key chunks merged through L2 between the list
//               and its alt buffer.  Since round 6 only one- and two-view calls send their long lists here; a batch sorts
//               8193..15872 keys with the windowed sort and longer lists through the split pre-pass (below), and this
//               kernel takes what those reject.
// A kernel per tier keeps each one's register budget its own: with the open-ended tier's code in the same kernel the
// 4097..8192 path (a quarter of C3's keys) ran out of the 128 VGPRs a 1024-thread workgroup gets and spilled.
// SEG: behind the tier's own lists the launch also sorts the depth SEGMENTS of longer lists (the split pre-pass's segment
// queue, seg_cap = its capacity): one launch, the segments fill the tail of the tier's own lists
template <int THREADS, int E, bool LAST, int NB = THREADS * E, int MIN_WAVES = (THREADS == 512 ? PGR_T1_WAVES : 4), bool SEG = false>
__global__ __launch_bounds__(THREADS, MIN_WAVES) void tile_sort_long_kernel(const BinView* __restrict__ views, int tiles,
                                                                 const uint4* __restrict__ queue,
                                                                 const uint32_t* __restrict__ n_queue,
                                                                 const uint4* __restrict__ seg_queue = nullptr,
                                                                 const uint32_t* __restrict__ n_seg = nullptr, uint32_t seg_cap = 0) {
    constexpr int CAP = THREADS * E;
    static_assert(!LAST || CAP == SORT_LARGE_MAX, "the open-ended tier");
    // bucket sort image: 8 B per key + 4 B per bucket, or (last tier) keys + 8192 counters = 157 KiB; the merge sort's padded keys fit
    constexpr size_t LDS_BYTES = LAST ? (size_t)SORT_LARGE_KEYS * 8 + SORT_LARGE_BUCKETS * 4 + SORT_MISC_BYTES : (size_t)CAP * 8 + (size_t)NB * 4 + SORT_MISC_BYTES;
    static_assert((size_t)THREADS * (E + 1) * 8 <= LDS_BYTES && LDS_BYTES <= 160 * 1024, "lds");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    __shared__ uint32_t s_cut[LAST ? PART_MAX_SEGMENTS + 3 : 1];
    uint64_t* skeys = reinterpret_cast<uint64_t*>(lds);
    const uint32_t own = *n_queue;
    uint32_t cand = own;
    if constexpr (SEG) cand += min(*n_seg, seg_cap);
    for (uint32_t k = blockIdx.x; k < cand; k += gridDim.x) {
        const uint2* bucket; uint32_t* out; int n; uint64_t* alt; ObjOut oo;
        if constexpr (SEG) {
            static_assert(!SEG || !LAST, "segments fit the tier");
            if (k >= own) {
                uint32_t pos;
                sort_segment(views, tiles, seg_queue[k - own], bucket, out, n, oo, pos);
                if (n > 0) {                     // (n == 0: a reservation that did not fit the queue, see partition_lists)
                    if (!bucket_sort_tile<THREADS, E, NB>(lds, bucket, out, n, oo.n_env, oo.last, pos, oo.tie))
                        merge_sort_tile<THREADS, E>(skeys, bucket, out, n, nullptr, oo.n_env, oo.last, pos, oo.tie, oo.inv);
                }
                __syncthreads();
                continue;
            }
        }
        sort_item(views, tiles, queue[k], bucket, out, n, oo, &alt);
        {
            if constexpr (!LAST) {
                if (!bucket_sort_tile<THREADS, E, NB>(lds, bucket, out, n, oo.n_env, oo.last, 0u, oo.tie))
                    merge_sort_tile<THREADS, E>(skeys, bucket, out, n, nullptr, oo.n_env, oo.last, 0u, oo.tie, oo.inv);
            } else if (n <= SORT_LARGE_KEYS) {
                if (!bucket_sort_tile<THREADS, E, SORT_LARGE_BUCKETS, SORT_LARGE_KEYS>(lds, bucket, out, n, oo.n_env, oo.last, 0u, oo.tie))
                    merge_sort_tile<THREADS, E>(skeys, bucket, out, n, nullptr, oo.n_env, oo.last, 0u, oo.tie, oo.inv);
            } else if (partition_sort_long<THREADS, E>(lds, s_cut, bucket, reinterpret_cast<uint2*>(alt), out, n, oo.n_env,
                                                       oo.last, oo.tie, oo.inv)) {
                // done: depth-partitioned through the alt buffer, every segment sorted in LDS
            } else {
                uint64_t* gk = reinterpret_cast<uint64_t*>(const_cast<uint2*>(bucket));
                for (int c0 = 0; c0 < n; c0 += CAP) {
                    merge_sort_tile<THREADS, E>(skeys, bucket + c0, nullptr, min(CAP, n - c0), gk + c0, -1, nullptr, 0u, oo.tie);
                    __syncthreads();
                }
                uint64_t *src = gk, *dst = alt;
                for (int run = CAP; run < n; run <<= 1) {
                    merge_round_global<THREADS, 16>(src, dst, n, run);
                    __syncthreads();
                    uint64_t* tmp = src; src = dst; dst = tmp;
                }
                for (int i = threadIdx.x; i < n; i += THREADS) out[i] = key_to_pos(oo.inv, (uint32_t)src[i]);
                if (oo.n_env >= 0)
                    mark_last_object<THREADS>([&](int i) { return key_to_pos(oo.inv, (uint32_t)src[i]); }, n, oo.n_env, oo.last);
            }
        }
        __syncthreads();   // LDS reuse across loop iterations
    }
}

// ---- split pre-pass (round 6): lists beyond the windowed sort -> depth segments for the 512 x 16 tier ----------------------
// Rounds 1-5 sorted such a list inside the open-ended kernel: three passes over it (extremes, coarse histogram, counting-sort
// scatter through the alt buffer) and then every segment one after the other in the same 1024-thread workgroup that holds
// its CU alone (partition_sort_long).  The passes need 16 KiB of LDS, not 157: here they run four workgroups to a CU, cut
// the list into segments of < 8192 keys (first bucket start at or beyond every multiple of 4096; no coarse bucket larger
// than that) and hand every segment to the SEGMENT queue of the 512 x 16 tier's kernel (tile_sort_long_kernel<.., SEG>), two
// workgroups per CU, all segments of all lists side by side.  What cannot be split (a piled-up coarse bucket, more than
// PART_MAX_SEGMENTS segments, no room in the queue) goes to the open-ended kernel's queue as before.
constexpr int PART_THREADS = 512;
constexpr int SEG_HALF = 4096;                       // segments hold < 2 * SEG_HALF keys: the 512 x 16 tier's capacity

constexpr size_t PART_LDS_BYTES = (size_t)(PART_BUCKETS + 4 + PART_THREADS / WAVE + PART_MAX_SEGMENTS + 3) * 4;

// the lists first, first + stride, ... of the queue; lds_words: PART_LDS_BYTES of LDS
__device__ __forceinline__ void partition_lists(uint32_t* __restrict__ lds_words, const BinView* __restrict__ views, int tiles,
                                                const uint4* __restrict__ queue, const uint32_t* __restrict__ n_queue,
                                                uint4* __restrict__ seg_queue, uint32_t* __restrict__ n_seg, uint32_t seg_cap,
                                                uint4* __restrict__ open_queue, uint32_t* __restrict__ n_open,
                                                uint32_t first, uint32_t stride) {
    constexpr int THREADS = PART_THREADS, WAVES = THREADS / WAVE, CH = PART_BUCKETS / (WAVES * WAVE), HALF = SEG_HALF, PB = 8;
    static_assert(PART_BUCKETS % (WAVES * WAVE) == 0, "bucket count");
    uint32_t* const s_hist = lds_words;                               // [PART_BUCKETS] counts -> starts -> cursors
    uint32_t* const s_misc = s_hist + PART_BUCKETS;                   // [0] min [1] max [2] largest bucket [3] queue slot [4..] wave totals
    uint32_t* const s_cut = s_misc + 4 + WAVES;                       // [PART_MAX_SEGMENTS + 3] segment g starts at s_cut[g]; n = not opened
    const int t = threadIdx.x, lane = t & (WAVE - 1), wave = t / WAVE;
    const uint32_t cand = *n_queue;
    for (uint32_t k = first; k < cand; k += stride) {
        const uint4 q = queue[k];
        const uint2* bucket; uint32_t* out; int n; uint64_t* alt64; ObjOut oo;
        sort_item(views, tiles, q, bucket, out, n, oo, &alt64);
        uint2* const alt = reinterpret_cast<uint2*>(alt64);
        bool ok = (n + HALF - 1) / HALF <= PART_MAX_SEGMENTS;        // (uniform)
        if (ok) {
            uint32_t dmin = 0xffffffffu, dmax = 0u;
            for (int i0 = t; i0 < n; i0 += PB * THREADS) {
                uint32_t dv[PB];
#pragma unroll
                for (int j = 0; j < PB; ++j) dv[j] = gload(&bucket[min(i0 + j * THREADS, n - 1)].x);   // (a clamped repeat changes no extreme)
#pragma unroll
                for (int j = 0; j < PB; ++j) { dmin = min(dmin, dv[j]); dmax = max(dmax, dv[j]); }
            }
            for (int i = t; i < PART_BUCKETS; i += THREADS) s_hist[i] = 0u;
            if (t < 4 + WAVES) s_misc[t] = t == 0 ? 0xffffffffu : 0u;
            if (t < PART_MAX_SEGMENTS + 3) s_cut[t] = (uint32_t)n;
            __syncthreads();
            dmin = ~wave_inclusive_max(~dmin);
            dmax = wave_inclusive_max(dmax);
            if (lane == WAVE - 1) { atomicMin(&s_misc[0], dmin); atomicMax(&s_misc[1], dmax); }
            __syncthreads();
            const uint32_t mn = s_misc[0];
            const float scale = (float)PART_BUCKETS / ((float)(s_misc[1] - mn) + 1.0f);
            auto coarse = [&](uint32_t d) { return min((uint32_t)((float)(d - mn) * scale), (uint32_t)(PART_BUCKETS - 1)); };
            for (int i0 = t; i0 < n; i0 += PB * THREADS) {
                uint32_t dv[PB];
#pragma unroll
                for (int j = 0; j < PB; ++j) dv[j] = gload(&bucket[min(i0 + j * THREADS, n - 1)].x);
#pragma unroll
                for (int j = 0; j < PB; ++j)
                    if (i0 + j * THREADS < n) atomicAdd(&s_hist[coarse(dv[j])], 1u);
            }
            __syncthreads();
            const int wbase = wave * (WAVE * CH);
            uint32_t tot = 0, big = 0;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const uint32_t h = s_hist[wbase + c * WAVE + lane];
                tot += h; big = max(big, h);
            }
            tot = wave_inclusive_scan(tot);
            big = wave_inclusive_max(big);
            if (lane == WAVE - 1) { s_misc[4 + wave] = tot; atomicMax(&s_misc[2], big); }
            __syncthreads();
            ok = s_misc[2] <= (uint32_t)HALF;
            if (ok) {
                uint32_t carry = 0;
                for (int w = 0; w < wave; ++w) carry += s_misc[4 + w];
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const uint32_t h = s_hist[wbase + c * WAVE + lane];
                    const uint32_t incl = wave_inclusive_scan(h);
                    const uint32_t s0 = carry + incl - h;
                    s_hist[wbase + c * WAVE + lane] = s0;
                    // segment g begins at the smallest bucket start >= g * HALF: bucket sizes <= HALF, so every g up to the
                    // last one is hit and consecutive cuts are < 2 * HALF apart (empty buckets share their successor's start)
                    const uint32_t g = (s0 + (uint32_t)HALF - 1u) / (uint32_t)HALF;
                    if (s0 < (uint32_t)n) atomicMin(&s_cut[g], s0);
                    carry += (uint32_t)__builtin_amdgcn_readlane((int)incl, WAVE - 1);
                }
                __syncthreads();
                // counting-sort scatter through the alt buffer (the starts become the cursors)
                for (int i0 = t; i0 < n; i0 += PB * THREADS) {
                    uint2 kv[PB];
#pragma unroll
                    for (int j = 0; j < PB; ++j) kv[j] = gload(bucket + min(i0 + j * THREADS, n - 1));
#pragma unroll
                    for (int j = 0; j < PB; ++j)
                        if (i0 + j * THREADS < n) gstore(alt + atomicAdd(&s_hist[coarse(kv[j].x)], 1u), kv[j]);
                }
                // the segments: g = 0 .. ceil(last start / HALF), without gaps (starts advance by <= HALF): that is
                // ceil(n / HALF) of them, or one more; entry g = [s_cut[g], s_cut[g + 1]) (s_cut = n beyond the last one)
                const uint32_t g_top = (uint32_t)((n + HALF - 1) / HALF);
                const uint32_t n_segs = g_top + (s_cut[g_top] < (uint32_t)n ? 1u : 0u);
                if (t == 0) s_misc[3] = gatomic_add(n_seg, n_segs);
                __syncthreads();
                const uint32_t slot = s_misc[3];
                ok = slot + n_segs <= seg_cap;
                if ((uint32_t)t < n_segs && slot + (uint32_t)t < seg_cap) {
                    const uint32_t a = s_cut[t], b = s_cut[t + 1];
                    // (no room for all of them: empty entries, and the whole list goes to the open-ended kernel)
                    seg_queue[slot + (uint32_t)t] = make_uint4(q.x, q.y, ok && b > a ? b - a : 0u, a);
                }
            }
        }
        if (!ok && t == 0) open_queue[gatomic_add(n_open, 1u)] = q;
        __syncthreads();                           // LDS reuse across the lists
    }
}

// 8193 .. SORT_WINDOW_MAX keys: the windowed sort; a rejected list joins the open-ended tier's queue (launched afterwards).
// part_blocks workgroups of the same launch run the split pre-pass of the longer lists instead (part_queue; 0 = none):
// that pass is as long as its longest list -- a few hundred workgroups that each walk one list three times -- and left most
// of the chip idle as a launch of its own (C5: 238 us per 32-view batch); here the windowed sort fills the chip beside it.
static_assert(PART_THREADS == SORT_WINDOW_THREADS && PART_LDS_BYTES <= SORT_WINDOW_LDS, "the two share workgroups");
__global__ __launch_bounds__(SORT_WINDOW_THREADS, 4) void tile_sort_window_kernel(const BinView* __restrict__ views, int tiles,
                                                                                   const uint4* __restrict__ queue,
                                                                                   const uint32_t* __restrict__ n_queue,
                                                                                   uint4* __restrict__ open_queue,
                                                                                   uint32_t* __restrict__ n_open,
                                                                                   uint32_t part_blocks = 0,
                                                                                   const uint4* __restrict__ part_queue = nullptr,
                                                                                   const uint32_t* __restrict__ n_part = nullptr,
                                                                                   uint4* __restrict__ seg_queue = nullptr,
                                                                                   uint32_t* __restrict__ n_seg = nullptr,
                                                                                   uint32_t seg_cap = 0) {
    constexpr size_t LDS_BYTES = SORT_WINDOW_LDS;
    static_assert(2 * (LDS_BYTES + 64) <= 160 * 1024, "two workgroups per CU");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    __shared__ uint32_t s_win[2 * (SORT_WINDOW_ROUNDS + 1) + 2];
    // every third workgroup of the launch's front is a pre-pass workgroup: the two kinds are resident side by side from the
    // start (all pre-pass workgroups first would fill the chip alone -- workgroups are dispatched in order)
    // (part_eff: the pre-pass workgroups that exist in THIS grid -- every one of them must be there, they stride over their
    // queue by their own number)
    const uint32_t part_eff = min(part_blocks, (gridDim.x + 2u) / 3u);
    const uint32_t part_before = min(part_eff, (blockIdx.x + 2u) / 3u);           // pre-pass workgroups among [0, blockIdx.x)
    if (blockIdx.x % 3u == 0u && blockIdx.x / 3u < part_eff) {
        partition_lists(reinterpret_cast<uint32_t*>(lds), views, tiles, part_queue, n_part, seg_queue, n_seg, seg_cap, open_queue,
                        n_open, blockIdx.x / 3u, part_eff);
        return;
    }
    if (gridDim.x == part_eff) return;                                            // (no workgroup left for the windowed sort: not launched so)
    const uint32_t cand = *n_queue;
    for (uint32_t k = blockIdx.x - part_before; k < cand; k += gridDim.x - part_eff) {
        const uint4 q = queue[k];
        const uint2* bucket; uint32_t* out; int n; ObjOut oo;
        sort_item(views, tiles, q, bucket, out, n, oo);
        if (!window_sort_tile<SORT_WINDOW_THREADS, SORT_WINDOW_E, SORT_WINDOW_BUCKETS, SORT_WINDOW_KEYS>(
                lds, s_win, bucket, out, n, oo.n_env, oo.last, oo.tie)) {
            if (threadIdx.x == 0) open_queue[gatomic_add(n_open, 1u)] = q;
        }
        __syncthreads();
    }
}

}  // namespace pgr
