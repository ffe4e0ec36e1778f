// tilebin.hip.h -- sync-free tile binning for gfx950: (Gaussian, tile) instances -> per-tile lists
// sorted by (depth bits, Gaussian index).  Replaces SURVEY.md section 8a rows a6-a9 (scan, emission,
// global 44-bit radix sort, range search) of the reference design with a layout that never leaves the
// chip's fast paths:
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
//   tile_sort    one workgroup per tile: the list is sorted by the 64-bit key (depth bits << 32 | index)
//                in LDS (register sorting network + merge-path rounds), which is exactly the order of a
//                stable sort of Gaussian-major emission by (tile, depth) -- the order the compositor's
//                blending depends on.  Tiers: <= 4096 keys (256 threads, 32 KiB), <= 16384 keys (1024
//                threads, 128 KiB of the CU's 160 KiB); longer lists are sorted in place in global memory.
//
// Everything is integer/bit work: bit-exact against the oracle's sorted lists.
#pragma once
#include "pgr_common.h"

namespace pgr {

constexpr int BIN_CHUNK = 4096;        // Gaussians per binning workgroup
constexpr int BIN_THREADS = 1024;
constexpr int BIN_LDS_TILES = 16384;   // tiles histogrammed per LDS pass (64 KiB)
constexpr int SORT_THREADS = 256;

// rect packed as 4 x uint16: minx, miny, maxx, maxy (exclusive); all zero = culled
__device__ __forceinline__ uint2 pack_rect(int minx, int miny, int maxx, int maxy) {
    return make_uint2((uint32_t)minx | ((uint32_t)miny << 16), (uint32_t)maxx | ((uint32_t)maxy << 16));
}

struct BinView {                 // per-view pointers used by the binning kernels (device table)
    const uint2* rects;          // [n] packed tile rectangles
    const float* depth;          // [n]
    uint32_t* tile_count;        // [tiles] zero-filled before bin_count
    uint32_t* rel;               // [chunks, tiles]
    uint2* ranges;               // [tiles]
    uint32_t* counters;          // [0] total instances, [1] overflow flag
    uint2* bucket;               // [max_instances] (depth bits, index), unsorted per tile
    uint32_t* gauss_sorted;      // [max_instances]
};

__global__ __launch_bounds__(BIN_THREADS) void bin_count_kernel(const BinView* __restrict__ views, int n, int grid_x,
                                                                int tiles) {
    extern __shared__ uint32_t hist[];
    const BinView& bv = views[blockIdx.y];
    const int chunk = blockIdx.x;
    const int begin = chunk * BIN_CHUNK, end = min(n, begin + BIN_CHUNK);
    for (int lo = 0; lo < tiles; lo += BIN_LDS_TILES) {
        const int span = min(BIN_LDS_TILES, tiles - lo);
        for (int t = threadIdx.x; t < span; t += BIN_THREADS) hist[t] = 0;
        __syncthreads();
        for (int i = begin + threadIdx.x; i < end; i += BIN_THREADS) {
            const uint2 r = bv.rects[i];
            const int minx = r.x & 0xffff, miny = r.x >> 16, maxx = r.y & 0xffff, maxy = r.y >> 16;
            for (int y = miny; y < maxy; ++y)
                for (int x = minx; x < maxx; ++x) {
                    const int t = y * grid_x + x - lo;
                    if ((unsigned)t < (unsigned)span) atomicAdd(&hist[t], 1u);
                }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < span; t += BIN_THREADS) {
            const uint32_t c = hist[t];
            bv.rel[(size_t)chunk * tiles + lo + t] = c ? atomicAdd(&bv.tile_count[lo + t], c) : 0u;
        }
        __syncthreads();
    }
}

// one workgroup per view
__global__ __launch_bounds__(1024) void tile_scan_kernel(const BinView* __restrict__ views, int tiles,
                                                         uint32_t max_instances) {
    __shared__ uint32_t wave_tot[1024 / WAVE];
    __shared__ uint32_t carry_s;
    const BinView& bv = views[blockIdx.x];
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < tiles; base += 1024) {
        const int idx = base + threadIdx.x;
        const uint32_t v = idx < tiles ? bv.tile_count[idx] : 0u;
        uint32_t s = v;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) {
            const uint32_t t = __shfl_up(s, d, WAVE);
            if (lane >= d) s += t;
        }
        if (lane == WAVE - 1) wave_tot[wid] = s;
        __syncthreads();
        uint32_t wave_prefix = 0;
        for (int w = 0; w < wid; ++w) wave_prefix += wave_tot[w];
        const uint32_t carry = carry_s;
        const uint32_t excl = carry + wave_prefix + s - v;
        if (idx < tiles) bv.ranges[idx] = make_uint2(excl, excl + v);
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wave_prefix + s;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        bv.counters[0] = carry_s;
        bv.counters[1] = carry_s > max_instances ? 1u : 0u;
    }
}

__global__ __launch_bounds__(BIN_THREADS) void bin_scatter_kernel(const BinView* __restrict__ views, int n, int grid_x,
                                                                  int tiles) {
    extern __shared__ uint32_t cursor[];
    const BinView& bv = views[blockIdx.y];
    if (bv.counters[1]) return;   // overflow: reported by the host, nothing may be written past the buffers
    const int chunk = blockIdx.x;
    const int begin = chunk * BIN_CHUNK, end = min(n, begin + BIN_CHUNK);
    for (int lo = 0; lo < tiles; lo += BIN_LDS_TILES) {
        const int span = min(BIN_LDS_TILES, tiles - lo);
        for (int t = threadIdx.x; t < span; t += BIN_THREADS)
            cursor[t] = bv.ranges[lo + t].x + bv.rel[(size_t)chunk * tiles + lo + t];
        __syncthreads();
        for (int i = begin + threadIdx.x; i < end; i += BIN_THREADS) {
            const uint2 r = bv.rects[i];
            const int minx = r.x & 0xffff, miny = r.x >> 16, maxx = r.y & 0xffff, maxy = r.y >> 16;
            if (maxx <= minx || maxy <= miny) continue;
            const uint32_t dbits = __float_as_uint(bv.depth[i]);
            for (int y = miny; y < maxy; ++y)
                for (int x = minx; x < maxx; ++x) {
                    const int t = y * grid_x + x - lo;
                    if ((unsigned)t < (unsigned)span) {
                        const uint32_t slot = atomicAdd(&cursor[t], 1u);
                        bv.bucket[slot] = make_uint2(dbits, (uint32_t)i);
                    }
                }
        }
        __syncthreads();
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

// Sorts n <= THREADS*E keys of `bucket` (global, (depth,idx) pairs) into out[] (indices only).
// skeys must hold THREADS*(E+1) keys.
template <int THREADS, int E>
__device__ __forceinline__ void merge_sort_tile(uint64_t* __restrict__ skeys, const uint2* __restrict__ bucket,
                                                uint32_t* __restrict__ out, int n) {
    const int t = threadIdx.x;
    uint64_t r[E];
    // the list is unordered, so WHICH keys a thread starts with is free: take them coalesced
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = e * THREADS + t;
        uint64_t k = KEY_INF;
        if (i < n) {
            const uint2 v = bucket[i];
            k = ((uint64_t)v.x << 32) | v.y;
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
    for (int i = t; i < n; i += THREADS) out[i] = (uint32_t)skeys[pad_idx<E>(i)];
}

// All-ascending bitonic network on n keys padded virtually to npad (a power of two) with +inf: a
// compare-exchange whose upper index is >= n is a no-op.  Only used, in place in global memory
// (L2-resident), for lists that exceed the LDS tiers.
__device__ __forceinline__ void bitonic_sort_global(uint64_t* keys, int n, int npad, int nthreads) {
    const int half = npad >> 1;
    for (int k = 2, lk = 1; k <= npad; k <<= 1, ++lk) {
        const int hk = k >> 1;
        for (int p = threadIdx.x; p < half; p += nthreads) {
            const int blk = p >> (lk - 1), off = p & (hk - 1);
            const int i = blk * k + off, j = blk * k + (k - 1 - off);
            if (j < n) {
                const uint64_t a = keys[i], b = keys[j];
                if (a > b) { keys[i] = b; keys[j] = a; }
            }
        }
        __syncthreads();
        for (int s = k >> 2, ls = lk - 2; s >= 1; s >>= 1, --ls) {
            for (int p = threadIdx.x; p < half; p += nthreads) {
                const int blk = p >> ls, off = p & (s - 1);
                const int i = blk * 2 * s + off, j = i + s;
                if (j < n) {
                    const uint64_t a = keys[i], b = keys[j];
                    if (a > b) { keys[i] = b; keys[j] = a; }
                }
            }
            __syncthreads();
        }
    }
}

constexpr int SORT_SMALL_MAX = SORT_THREADS * 16;      // 4096 keys, 32 KiB LDS
constexpr int SORT_LARGE_THREADS = 1024;
constexpr int SORT_LARGE_MAX = SORT_LARGE_THREADS * 16; // 16384 keys, 136 KiB LDS with padding

__device__ __forceinline__ bool sort_item(const BinView* __restrict__ views, int tiles,
                                          const uint32_t* __restrict__ work_order, uint32_t index,
                                          const uint2*& bucket, uint32_t*& out, int& n) {
    const uint32_t item = work_order[2 * index] >> 1;   // work items come in (half 0, half 1) pairs
    const uint32_t view = item / (uint32_t)tiles;
    const uint32_t tile = item - view * (uint32_t)tiles;
    const BinView& bv = views[view];
    if (bv.counters[1]) return false;
    const uint2 range = bv.ranges[tile];
    n = (int)(range.y - range.x);
    bucket = bv.bucket + range.x;
    out = bv.gauss_sorted + range.x;
    return n > 0;
}

// grid = n_views * tiles workgroups of 256; lists of 1..4096 entries
__global__ __launch_bounds__(SORT_THREADS) void tile_sort_kernel(const BinView* __restrict__ views, int tiles,
                                                                 const uint32_t* __restrict__ work_order) {
    __shared__ uint64_t skeys[SORT_THREADS * 17];
    const uint2* bucket; uint32_t* out; int n;
    if (!sort_item(views, tiles, work_order, blockIdx.x, bucket, out, n)) return;
    if (n > SORT_SMALL_MAX) return;                      // tile_sort_large_kernel's
    if (n <= SORT_THREADS * 2) merge_sort_tile<SORT_THREADS, 2>(skeys, bucket, out, n);
    else if (n <= SORT_THREADS * 4) merge_sort_tile<SORT_THREADS, 4>(skeys, bucket, out, n);
    else if (n <= SORT_THREADS * 8) merge_sort_tile<SORT_THREADS, 8>(skeys, bucket, out, n);
    else merge_sort_tile<SORT_THREADS, 16>(skeys, bucket, out, n);
}

// Lists longer than 4096: work_order is sorted by descending length class, so they are its first
// n_candidates entries (device counter written by order_scan_kernel); workgroups stride over them.
__global__ __launch_bounds__(SORT_LARGE_THREADS) void tile_sort_large_kernel(const BinView* __restrict__ views, int tiles,
                                                                             const uint32_t* __restrict__ work_order,
                                                                             const uint32_t* __restrict__ n_candidates) {
    __shared__ uint64_t skeys[SORT_LARGE_THREADS * 17];   // 136 KiB of the CU's 160 KiB
    const uint32_t cand = *n_candidates;
    for (uint32_t k = blockIdx.x; k < cand; k += gridDim.x) {
        const uint2* bucket; uint32_t* out; int n;
        const bool ok = sort_item(views, tiles, work_order, k, bucket, out, n);
        if (ok && n > SORT_SMALL_MAX) {
            if (n <= SORT_LARGE_MAX) {
                merge_sort_tile<SORT_LARGE_THREADS, 16>(skeys, bucket, out, n);
            } else {
                uint64_t* gk = reinterpret_cast<uint64_t*>(const_cast<uint2*>(bucket));
                for (int i = threadIdx.x; i < n; i += SORT_LARGE_THREADS) {
                    const uint2 e = bucket[i];
                    gk[i] = ((uint64_t)e.x << 32) | e.y;
                }
                __syncthreads();
                int npad = 1;
                while (npad < n) npad <<= 1;
                bitonic_sort_global(gk, n, npad, SORT_LARGE_THREADS);
                for (int i = threadIdx.x; i < n; i += SORT_LARGE_THREADS) out[i] = (uint32_t)gk[i];
            }
        }
        __syncthreads();   // skeys reuse across loop iterations
    }
}

}  // namespace pgr
