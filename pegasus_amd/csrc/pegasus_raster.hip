// pegasus_raster.hip -- C ABI of libpegasus_raster.so (see include/pegasus_raster.h).
// gfx950 only.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC
//
// Pipeline of one batch of views of one scene (all stages enqueued on the caller's stream, no host
// round trip until the end-of-batch status read):
//   per view : pack_camera, preprocess                           (preprocess.hip.h)
//   batch    : bin_count -> tile_scan -> bin_scatter -> work order -> tile_sort(_large)   (tilebin.hip.h)
//   batch    : composite_wave over every (view, tile, half) work item, longest lists first (composite.hip.h)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "backward.hip.h"
#include "blockcull.hip.h"
#include "knn.hip.h"
#include "compose.hip.h"
#include "composite.hip.h"
#include "pgr_common.h"
#include "preprocess.hip.h"
#include "tilebin.hip.h"

namespace pgr {

static thread_local char g_hip_error[256] = "";

static bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    snprintf(g_hip_error, sizeof(g_hip_error), "%s: %s", what, hipGetErrorString(e));
    return false;
}

static size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// Calls of one or two views (the drop-in GaussianRasterizer: one camera per call) cannot fill the chip; their launches are
// latency-bound and get a few arrangements of their own (fewer, fuller launches).  Results never depend on it.
constexpr int SMALL_BATCH_VIEWS = 2;
#ifndef PGR_PRE_SMALL_SCENE
#define PGR_PRE_SMALL_SCENE 400000
#endif
constexpr int PRE_SMALL_SCENE = PGR_PRE_SMALL_SCENE;   // Gaussians: below this the preprocess spreads a batch's views over gridDim.y ...
constexpr int PRE_VIEW_GROUP = 8;         // ... in groups of this many

// A/B switch for tests and measurements (read per call; results never depend on it)
static bool block_cull_enabled() {
    const char* e = getenv("PGR_BLOCK_CULL");
    return !(e && e[0] == '0');
}


static bool records_enabled() {      // PGR_BIN_RECORDS=0: the scatter walk re-evaluates every candidate (A/B, tests)
    const char* e = getenv("PGR_BIN_RECORDS");
    return !(e && e[0] == '0');
}

// layers > 1 (pgr_forward_layers_async): the view is `layers` stacked copies of the tile grid -- per-(tile, layer) lists
static Layout make_layout(int32_t n, int32_t width, int32_t height, int64_t max_instances, int32_t layers = 1) {
    Layout L{};
    const size_t N = (size_t)(n > 0 ? n : 0), I = (size_t)(max_instances > 0 ? max_instances : 0);
    const int gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE * (layers > 1 ? layers : 1);
    L.tiles = gx * gy;
    L.grid_x = gx;
    L.grid_y = gy;
    L.n_blocks = (int)((N + PRE_BLOCK - 1) / PRE_BLOCK);
    L.n_chunks = (int)((N + BIN_CHUNK - 1) / BIN_CHUNK);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes ? bytes : 1); return o; };
    L.cam = take(sizeof(CameraDev));
    L.counters = take(64);
    L.splats = take(N * 48);
    L.radii = take(N * 4);
    L.rects = take(N * 8);
    L.crects = take(N * 8);
    L.rel = take((size_t)L.n_chunks * L.tiles * 4);
    L.ranges = take((size_t)L.tiles * 8);
    L.bucket = take(I * 8);
    L.alt = take(I * 8);
    L.gauss_sorted = take(I * 4);
    L.total = off;
    L.max_instances = (int64_t)I;
    return L;
}

static int check_scene(const PgrScene* s) {
    if (!s || s->n < 0) return PGR_ERR_INVALID_ARGUMENT;
    if (s->n == 0) return PGR_OK;
    if (!s->means3d || !s->opacities) return PGR_ERR_INVALID_ARGUMENT;
    if ((s->shs == nullptr) == (s->colors_precomp == nullptr)) return PGR_ERR_INVALID_ARGUMENT;
    const bool have_sr = s->scales != nullptr && s->rotations != nullptr;
    if (have_sr == (s->cov3d_precomp != nullptr)) return PGR_ERR_INVALID_ARGUMENT;
    if (s->shs && (s->sh_degree < 0 || s->sh_degree > 3 || s->sh_stride < (s->sh_degree + 1) * (s->sh_degree + 1)))
        return PGR_ERR_INVALID_ARGUMENT;
    if (s->shs_rest && (!s->shs || s->sh_stride < 2)) return PGR_ERR_INVALID_ARGUMENT;   // split layout: dc + at least one more
    return PGR_OK;
}

// Per-view slice of a workspace.
struct ViewWs {
    CameraDev* cam;
    uint32_t* counters;
    float4* splats;   // [n,3] per-Gaussian records
    int32_t* radii;   // per-view home of radii when the caller passes no radii output
    uint2 *rects, *crects;
    uint32_t *tile_count, *rel;
    uint2* ranges;
    uint2* bucket;
    uint64_t* alt;
    uint32_t* gauss_sorted;
    uint32_t* obj_last;   // batch header
};

static ViewWs carve(char* ws, const Layout& L) {
    ViewWs v;
    v.cam = reinterpret_cast<CameraDev*>(ws + L.cam);
    v.counters = reinterpret_cast<uint32_t*>(ws + L.counters);
    v.splats = reinterpret_cast<float4*>(ws + L.splats);
    v.radii = reinterpret_cast<int32_t*>(ws + L.radii);
    v.rects = reinterpret_cast<uint2*>(ws + L.rects);
    v.crects = reinterpret_cast<uint2*>(ws + L.crects);
    v.tile_count = nullptr;   // lives in the batch header (BatchLayout::tile_counts)
    v.rel = reinterpret_cast<uint32_t*>(ws + L.rel);
    v.ranges = reinterpret_cast<uint2*>(ws + L.ranges);
    v.bucket = reinterpret_cast<uint2*>(ws + L.bucket);
    v.alt = reinterpret_cast<uint64_t*>(ws + L.alt);
    v.gauss_sorted = reinterpret_cast<uint32_t*>(ws + L.gauss_sorted);
    v.obj_last = nullptr;
    return v;
}

// Batch header placed in front of the per-view slices.
struct BatchLayout {
    size_t tables, cams, status, tile_counts, order_state, work_order, long_list, tie_inv, vis, obj_u8, views, total;
    int n_groups, vis_words;
    size_t view_table_off, bin_table_off, pre_table_off, tables_bytes;   // inside `tables` (one H2D copy)
    size_t order_slots;
    size_t seg_cap;              // entries of the segment queue (behind the SORT_TIERS tier queues)
    size_t per_view;
};

// Host-side image of the tables + the status words read back, laid out exactly as on the device.
static size_t host_scratch_bytes(int n_views) {
    return align_up((size_t)n_views * sizeof(ViewEntry), 16) + align_up((size_t)n_views * sizeof(BinView), 16) +
           align_up((size_t)n_views * sizeof(PreOut), 16) + (size_t)n_views * 8;
}

static BatchLayout make_batch_layout(const Layout& L, int n_views, size_t n_scene) {
    BatchLayout B{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes ? bytes : 1); return o; };
    B.view_table_off = 0;
    B.bin_table_off = B.view_table_off + align_up((size_t)n_views * sizeof(ViewEntry), 16);
    B.pre_table_off = B.bin_table_off + align_up((size_t)n_views * sizeof(BinView), 16);
    B.tables_bytes = B.pre_table_off + align_up((size_t)n_views * sizeof(PreOut), 16);
    B.tables = take(B.tables_bytes);
    B.cams = take((size_t)n_views * sizeof(CameraDev));
    B.status = take((size_t)n_views * 8);          // per view: [0] listed instances, [1] overflow flag
    B.tile_counts = take((size_t)n_views * L.tiles * 8);   // [tile_count u32 | obj_last u32] x views: one memset
    B.order_state = take(ORDER_STATE_WORDS * 4);
    // NUM_XCD interleaved streams; each holds the items of its band of tile rows for every view
    B.order_slots = (size_t)NUM_XCD * max_band_rows(L.grid_y) * L.grid_x * ITEMS_PER_TILE * n_views;
    B.work_order = take(B.order_slots * 4);
    // the sort queues: one per tier, and the segment queue of the split pre-pass -- a list of n > SORT_WINDOW_MAX keys yields
    // at most n / SEG_HALF + 2 <= n (1 / 4096 + 2 / 15872) = n / 2702 segments, and a view's lists hold max_instances keys
    B.seg_cap = (size_t)n_views * ((size_t)(L.max_instances > 0 ? L.max_instances : 0) / 2702 + 2);
    B.long_list = take(((size_t)n_views * L.tiles * SORT_TIERS + B.seg_cap) * sizeof(uint4));
    B.tie_inv = take((size_t)n_scene * 4);     // inverse of PgrScene::tie_index (filled only when one is given)
    B.n_groups = (int)((n_scene + WAVE - 1) / WAVE);
    B.vis_words = (n_views + 31) / 32;
    B.vis = take((size_t)B.n_groups * B.vis_words * 4);
    B.obj_u8 = take(n_scene);                  // object ids as bytes (fused semantic pass)
    B.views = off;
    B.per_view = align_up(L.total);
    B.total = off + (size_t)n_views * B.per_view;
    return B;
}

static int check_camera(const PgrCamera* cam, const PgrOutputs* out, bool layered = false) {
    if (!cam || !out || cam->image_width <= 0 || cam->image_height <= 0 || !(cam->tanfovx > 0.f) ||
        !(cam->tanfovy > 0.f) || !cam->viewmatrix || !cam->projmatrix || !cam->campos || !cam->bg ||
        // color + depth, or (records-only view) the frame record alone; a layered call writes mask planes only
        (layered ? !out->sem_masks : !((out->color && out->depth) || (out->record && !out->color && !out->depth))) ||
        (cam->depth_mode != PGR_DEPTH_EXPECTED && cam->depth_mode != PGR_DEPTH_NORMALIZED))
        return PGR_ERR_INVALID_ARGUMENT;
    // tile coordinates are packed into 16 bits
    if ((cam->image_width + TILE - 1) / TILE > 0xffff || (cam->image_height + TILE - 1) / TILE > 0xffff)
        return PGR_ERR_INVALID_ARGUMENT;
    return PGR_OK;
}

static int zero_outputs(const PgrOutputs* out, size_t P, hipStream_t stream, int n_masks) {
    if (out->color && !hip_ok(hipMemsetAsync(out->color, 0, 3 * P * sizeof(float), stream), "memset color"))
        return PGR_ERR_LAUNCH_FAILURE;
    if (out->depth && !hip_ok(hipMemsetAsync(out->depth, 0, P * sizeof(float), stream), "memset depth"))
        return PGR_ERR_LAUNCH_FAILURE;
    if (out->sem_masks && n_masks > 0 && !hip_ok(hipMemsetAsync(out->sem_masks, 0, (size_t)n_masks * P, stream), "memset masks"))
        return PGR_ERR_LAUNCH_FAILURE;
    if (out->record) {
        const size_t bytes = align_up(3 * P, 16) + align_up(2 * P, 16) + align_up((size_t)((n_masks + 7) / 8) * P, 16);
        if (!hip_ok(hipMemsetAsync(out->record, 0, bytes, stream), "memset record")) return PGR_ERR_LAUNCH_FAILURE;
    }
    if (out->final_T && !hip_ok(hipMemsetAsync(out->final_T, 0, P * sizeof(float), stream), "memset T"))
        return PGR_ERR_LAUNCH_FAILURE;
    if (out->n_contrib && !hip_ok(hipMemsetAsync(out->n_contrib, 0, P * sizeof(uint32_t), stream), "memset n"))
        return PGR_ERR_LAUNCH_FAILURE;
    if (out->sem_color && !hip_ok(hipMemsetAsync(out->sem_color, 0, 3 * P * sizeof(float), stream), "memset sem"))
        return PGR_ERR_LAUNCH_FAILURE;
    if (out->sem_depth && !hip_ok(hipMemsetAsync(out->sem_depth, 0, P * sizeof(float), stream), "memset semd"))
        return PGR_ERR_LAUNCH_FAILURE;
    return PGR_OK;
}

// The batch header in ONE launch: tile counters | obj_last and the work-order state cleared, the work order invalid, the first
// CAM_PACK_MAX cameras packed, and -- one- and two-view calls -- the pointer tables written from the launch arguments.  A
// single-view call lasts 0.45 ms on the GPU: the H2D copy of its 300 bytes of tables and the camera launch were 10 us of it.
constexpr int HEADER_TABLE_WORDS = 256;
struct HeaderTables { uint32_t w[HEADER_TABLE_WORDS]; };
__global__ __launch_bounds__(256) void batch_header_kernel(uint32_t* __restrict__ zero, size_t n_zero, uint32_t* __restrict__ ff,
                                                           size_t n_ff, CamPack cams, int cam_count, int width, int height,
                                                           CameraDev* __restrict__ cams_out, HeaderTables tables,
                                                           uint32_t* __restrict__ tables_out, int table_words) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_zero; i += stride) gstore(zero + i, 0u);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_ff; i += stride) gstore(ff + i, INVALID_ITEM);
    if ((int)blockIdx.x < cam_count && threadIdx.x < 64)
        pack_camera(cams, (int)blockIdx.x, (int)threadIdx.x, width, height, cams_out + blockIdx.x);
    if (blockIdx.x == gridDim.x - 1)
        for (int i = threadIdx.x; i < table_words; i += blockDim.x) gstore(tables_out + i, tables.w[i]);
}

// The whole hot path for a batch of views of ONE scene.  All views share the image size.
// ev: optional PGR_NUM_STAGES+1 events recorded at the stage boundaries (profiling entry point only).
// host_scratch: NULL = synchronous call (tables staged from pageable memory, stream synchronised at the end,
// num_instances filled).  Non-NULL = pinned host memory of host_scratch_bytes(n_views): nothing blocks, the
// status words land in its tail when the stream reaches them (pgr_batch_status reads them).
static int32_t forward_batch_impl(const PgrScene* scene, int n_views, const PgrCamera* cams, const PgrOutputs* outs,
                                  void* workspace, size_t workspace_bytes, int64_t max_instances,
                                  int64_t* num_instances, hipStream_t stream, hipEvent_t* ev,
                                  void* host_scratch = nullptr, const PgrSemantic* semantic = nullptr,
                                  const PgrPosedObjects* posed = nullptr, const PgrLayers* layers = nullptr,
                                  hipEvent_t status_event = nullptr) {
    auto mark = [&](int k) { if (ev) (void)hipEventRecord(ev[k], stream); };
    // every argument check happens here, before the first enqueue: an early return below this block would leave work
    // on the stream that still reads the (pageable) table staging of the synchronous path
    // (per-Gaussian arrays of an EMPTY scene may be NULL: torch hands out a null pointer for an empty tensor)
    const bool empty = scene && scene->n == 0;
    if (posed && ((!posed->object_id && !empty) || !posed->poses || posed->k_objects <= 0 ||
                  (scene && (scene->cov3d_precomp || scene->shs_rest))))
        return PGR_ERR_INVALID_ARGUMENT;
    if (semantic && ((!semantic->object_id && !empty) || !semantic->colors || semantic->n_env < 0 || semantic->k_objects <= 0))
        return PGR_ERR_INVALID_ARGUMENT;
    if (layers && (semantic || (!layers->layer_id && !empty) || !layers->mask_colors || layers->n_layers <= 0 || layers->n_layers > 4096))
        return PGR_ERR_INVALID_ARGUMENT;
    if (n_views <= 0 || !cams || !outs) return PGR_ERR_INVALID_ARGUMENT;
    if (num_instances) for (int v = 0; v < n_views; ++v) num_instances[v] = 0;
    if (int rc = check_scene(scene)) return rc;
    if (max_instances < 0 || max_instances > 0x7fffffffLL) return PGR_ERR_INVALID_ARGUMENT;
    const int W = cams[0].image_width, H = cams[0].image_height, N = scene->n;
    for (int v = 0; v < n_views; ++v) {
        if (int rc = check_camera(&cams[v], &outs[v], layers != nullptr)) return rc;
        if (cams[v].image_width != W || cams[v].image_height != H) return PGR_ERR_INVALID_ARGUMENT;
        // a semantic descriptor asks for the objects-only image of EVERY view of the batch
        // (a records-only view takes it as the mask planes of its record instead: no image, but then the colours to
        // threshold against must be there)
        if (semantic && !outs[v].sem_color && !(outs[v].record && !outs[v].color && semantic->mask_colors && !outs[v].sem_depth))
            return PGR_ERR_INVALID_ARGUMENT;
        // masks in the compositor's epilogue need the colours to threshold against
        if (!layers && outs[v].sem_masks && !(semantic && semantic->mask_colors)) return PGR_ERR_INVALID_ARGUMENT;
    }
    const int n_layers = layers ? layers->n_layers : 1;
    if (layers && (size_t)n_layers * ((H + TILE - 1) / TILE) > 0xffffu) return PGR_ERR_INVALID_ARGUMENT;   // 16-bit tile rows
    const size_t P = (size_t)W * H;
    // failure after the first enqueue: the synchronous path drains the stream before its staging memory goes away
    auto fail = [&](int32_t rc) {
        if (!host_scratch) (void)hipStreamSynchronize(stream);
        return rc;
    };

    // N == 0: outputs stay zero-filled, no background (SURVEY.md section 8a "Edge cases")
    if (N == 0) {
        for (int v = 0; v < n_views; ++v)
            if (int rc = zero_outputs(&outs[v], P, stream, layers ? n_layers : (semantic && semantic->mask_colors ? semantic->k_objects : 0))) return rc;
        if (status_event && !hip_ok(hipEventRecord(status_event, stream), "record status event")) return PGR_ERR_LAUNCH_FAILURE;
        return PGR_OK;
    }

    if (!workspace) return PGR_ERR_INVALID_ARGUMENT;
    const Layout L = make_layout(N, W, H, max_instances, n_layers);
    const int layer_tiles = L.tiles / n_layers, layer_rows = L.grid_y / n_layers;
    if (layers && layer_tiles > BIN_LDS_TILES) return PGR_ERR_INVALID_ARGUMENT;     // one layer per LDS pass
    const BatchLayout B = make_batch_layout(L, n_views, (size_t)N);
    if (workspace_bytes < B.total) return PGR_ERR_WORKSPACE_TOO_SMALL;
    char* ws = static_cast<char*>(workspace);
    auto* view_table = reinterpret_cast<ViewEntry*>(ws + B.tables + B.view_table_off);
    auto* bin_table = reinterpret_cast<BinView*>(ws + B.tables + B.bin_table_off);
    auto* pre_table = reinterpret_cast<PreOut*>(ws + B.tables + B.pre_table_off);
    auto* cams_dev = reinterpret_cast<CameraDev*>(ws + B.cams);
    auto* status_dev = reinterpret_cast<uint32_t*>(ws + B.status);
    auto* order_state = reinterpret_cast<uint32_t*>(ws + B.order_state);
    auto* sort_queue = reinterpret_cast<uint4*>(ws + B.long_list);
    auto* work_order = reinterpret_cast<uint32_t*>(ws + B.work_order);
    std::vector<ViewWs> vw((size_t)n_views);
    std::vector<char> pageable;
    char* hs = static_cast<char*>(host_scratch);
    if (!hs) {
        pageable.resize(host_scratch_bytes(n_views));
        hs = pageable.data();
    }
    auto* table = reinterpret_cast<ViewEntry*>(hs + B.view_table_off);
    auto* bins = reinterpret_cast<BinView*>(hs + B.bin_table_off);
    auto* pres = reinterpret_cast<PreOut*>(hs + B.pre_table_off);
    auto* h_status = reinterpret_cast<uint32_t*>(hs + B.tables_bytes);
    bool want_aux = false, want_sem = false;
    // the count walk's verdicts live where the sort's outputs will (alt, gauss_sorted: contiguous, dead until the sort)
    const size_t verdict_room = (L.total - L.alt) / (VERDICT_REGION_WORDS * 4);
    const int verdict_groups = layer_tiles <= BIN_LDS_TILES && records_enabled()
                                   ? (int)std::min<size_t>(verdict_room, (size_t)(N + WAVE - 1) / WAVE) : 0;
    for (int v = 0; v < n_views; ++v) {
        vw[v] = carve(ws + B.views + (size_t)v * B.per_view, L);
        vw[v].cam = cams_dev + v;          // cameras of a batch are contiguous: preprocess walks them
        vw[v].counters = status_dev + 2 * v;   // and so are the status words: one D2H copy per batch
        vw[v].tile_count = reinterpret_cast<uint32_t*>(ws + B.tile_counts) + (size_t)v * L.tiles;
        vw[v].obj_last = reinterpret_cast<uint32_t*>(ws + B.tile_counts + (size_t)n_views * L.tiles * 4) + (size_t)v * L.tiles;
        ViewEntry& e = table[v];
        memset(&e, 0, sizeof(e));
        e.cam = vw[v].cam; e.ranges = vw[v].ranges; e.gauss_sorted = vw[v].gauss_sorted; e.splats = vw[v].splats;
        e.out = CompOut{outs[v].color, outs[v].depth, outs[v].final_T, outs[v].n_contrib};
        e.counters = vw[v].counters;
        // fused semantic pass: the same walk also accumulates the objects-only image
        e.sem_color = semantic ? outs[v].sem_color : nullptr;
        e.sem_depth = semantic ? outs[v].sem_depth : nullptr;
        e.obj_last = vw[v].obj_last;
        e.sem_masks = (semantic || layers) ? outs[v].sem_masks : nullptr;
        e.record = layers ? nullptr : outs[v].record;
        want_sem = want_sem || e.sem_color || (semantic && e.record);
        want_aux = want_aux || outs[v].final_T || outs[v].n_contrib;
        bins[v] = BinView{vw[v].crects, vw[v].splats, vw[v].tile_count, vw[v].rel, vw[v].ranges,
                          vw[v].counters, vw[v].bucket, vw[v].gauss_sorted, vw[v].alt, vw[v].obj_last,
                          semantic ? semantic->n_env : -1, scene->tie_index,
                          scene->tie_index ? (scene->tie_inv ? scene->tie_inv : reinterpret_cast<const uint32_t*>(ws + B.tie_inv)) : nullptr,
                          reinterpret_cast<uint2*>(vw[v].alt)};
        // radii and the reference-style 3-sigma rectangles are per-view OUTPUTS: written only when the caller asks for
        // radii (12 N bytes per view the frame path never reads; pgr_workspace_view's `rects` is valid only then)
        pres[v] = PreOut{vw[v].splats, outs[v].radii ? vw[v].rects : nullptr, vw[v].crects, outs[v].radii};
    }
    // the pointer tables: in the header launch's arguments when they are small (one / two views), else one H2D copy
    HeaderTables header_tables;
    int table_words = 0;
    if (n_views <= SMALL_BATCH_VIEWS && B.tables_bytes <= sizeof(header_tables)) {
        memcpy(header_tables.w, hs, B.tables_bytes);
        table_words = (int)(B.tables_bytes / 4);
    } else if (!hip_ok(hipMemcpyAsync(ws + B.tables, hs, B.tables_bytes, hipMemcpyHostToDevice, stream), "memcpy tables"))
        return fail(PGR_ERR_LAUNCH_FAILURE);

    // ---- stage 0: batch header (+ cameras) + per-Gaussian preprocess
    mark(0);
    static_assert(ORDER_STATE_WORDS * 4 <= 256, "order state fits its slot");
    auto camera_pack = [&](int v0, int cnt) {
        CamPack cp;
        for (int k = 0; k < cnt; ++k) {
            const PgrCamera& c = cams[v0 + k];
            cp.view[k] = c.viewmatrix; cp.proj[k] = c.projmatrix; cp.campos[k] = c.campos; cp.bg[k] = c.bg;
            cp.tanfovx[k] = c.tanfovx; cp.tanfovy[k] = c.tanfovy; cp.depth_mode[k] = c.depth_mode;
        }
        return cp;
    };
    {
        const int cnt = std::min(CAM_PACK_MAX, n_views);
        batch_header_kernel<<<256, 256, 0, stream>>>(
            reinterpret_cast<uint32_t*>(ws + B.tile_counts), (B.work_order - B.tile_counts) / 4,
            reinterpret_cast<uint32_t*>(ws + B.work_order), B.order_slots, camera_pack(0, cnt), cnt, W, H, cams_dev, header_tables,
            reinterpret_cast<uint32_t*>(ws + B.tables), table_words);
    }
    if (scene->tie_index && !scene->tie_inv)     // (a per-scene constant: pgr_scene_prepare computes it once)
        invert_tie_index_kernel<<<(N + 255) / 256, 256, 0, stream>>>(N, scene->tie_index,
                                                                      reinterpret_cast<uint32_t*>(ws + B.tie_inv));
    for (int v0 = CAM_PACK_MAX; v0 < n_views; v0 += CAM_PACK_MAX) {
        const int cnt = std::min(CAM_PACK_MAX, n_views - v0);
        pack_camera_kernel<<<cnt, 64, 0, stream>>>(camera_pack(v0, cnt), W, H, cams_dev + v0);
    }
    // which 64-Gaussian blocks can show up in which view: decided by the preprocess waves themselves (conservative;
    // PGR_BLOCK_CULL=0 switches the test off), left in `vis` for the binning walks
    // (one- and two-view calls skip it: bounding the blocks costs their latency-bound preprocess more than the skipped
    // work returns -- 98 -> 88 us for a single view of the 2 M-Gaussian scene)
    // small scene, many views (an objects-only pass, a single object): the batch's views are spread over gridDim.y in groups of
    // PRE_VIEW_GROUP instead of walked by one wave -- 200 k Gaussians are 3 000 waves, a quarter of what the chip holds, and 32
    // views in a row per wave cost 0.31 ms where the arithmetic is 0.1.  Block culling is off there (its words hold 32 views).
    const bool split_views = N <= PRE_SMALL_SCENE && n_views >= 2 * PRE_VIEW_GROUP;
    const int view_groups = split_views ? (n_views + PRE_VIEW_GROUP - 1) / PRE_VIEW_GROUP : 1;
    const int views_per_block = split_views ? PRE_VIEW_GROUP : n_views;
    uint32_t* vis = block_cull_enabled() && n_views > SMALL_BATCH_VIEWS && !split_views ? reinterpret_cast<uint32_t*>(ws + B.vis) : nullptr;
    // one pass over the Gaussians for the whole batch (scene data read once, per-view outputs written)
    const PosedDev pd{posed ? posed->object_id : nullptr, posed ? posed->poses : nullptr, posed ? posed->k_objects : 0};
    const int deg = scene->shs ? scene->sh_degree : 0;
    const LayerDev ld{layers ? layers->layer_id : nullptr, n_layers};
#define PGR_PRE(D, Pz, Ly, Sp) preprocess_batch_kernel<D, Pz, Ly, Sp><<<dim3(L.n_blocks, view_groups), PRE_BLOCK, 0, stream>>>(*scene, cams_dev, pre_table, n_views, pd, vis, B.vis_words, ld, views_per_block)
#define PGR_PRE_DEG(Pz, Ly, Sp) switch (deg) { case 0: PGR_PRE(0, Pz, Ly, Sp); break; case 1: PGR_PRE(1, Pz, Ly, Sp); break; \
                                               case 2: PGR_PRE(2, Pz, Ly, Sp); break; default: PGR_PRE(3, Pz, Ly, Sp); break; }
    // (the split SH layout -- PgrScene::shs_rest, the single-view render() of a model as stored -- has its own kernels for the
    //  plain and the layered call; a posed call takes the concatenated layout: check above)
    if (layers) {
        if (posed) { PGR_PRE_DEG(true, true, false) } else if (scene->shs_rest) { PGR_PRE_DEG(false, true, true) } else { PGR_PRE_DEG(false, true, false) }
    } else {
        if (posed) { PGR_PRE_DEG(true, false, false) } else if (scene->shs_rest) { PGR_PRE_DEG(false, false, true) } else { PGR_PRE_DEG(false, false, false) }
    }
#undef PGR_PRE_DEG
#undef PGR_PRE
    mark(1);
    // ---- stage 1: per-chunk LDS tile histograms + slice reservation, then the tile scan (device only)
    const int grid_x = (W + TILE - 1) / TILE;
    const bool few_views = n_views <= SMALL_BATCH_VIEWS;
    const size_t lds = bin_lds_bytes(layer_tiles, few_views ? BIN_THREADS_SMALL : BIN_THREADS);
    const BinLayers bl{layers ? layers->layer_id : nullptr, layer_tiles, layer_rows, n_layers};
    if (few_views)
        bin_kernel<false, BIN_THREADS_SMALL><<<dim3(L.n_chunks, n_views), BIN_THREADS_SMALL, lds, stream>>>(
            bin_table, N, grid_x, L.tiles, W, H, vis, B.vis_words, verdict_groups, bl);
    else
        bin_kernel<false, BIN_THREADS><<<dim3(L.n_chunks, n_views), BIN_THREADS, lds, stream>>>(
            bin_table, N, grid_x, L.tiles, W, H, vis, B.vis_words, verdict_groups, bl);
    // tile scan -> ranges; the same pass counts the compositor's work items per (XCD stream, length class) and its last
    // workgroup turns the counts into the streams' write cursors
    tile_scan_kernel<<<n_views, 1024, 0, stream>>>(bin_table, L.tiles, (uint32_t)max_instances, L.grid_x, order_state,
                                                   layers ? 1 : 0, status_event ? h_status : nullptr);
    // early status: the scan wrote the status words into the pinned host scratch itself; whoever waits for this event reads
    // them two thirds of a single-view call before its compositor ends
    if (status_event && !hip_ok(hipEventRecord(status_event, stream), "record status event")) return fail(PGR_ERR_LAUNCH_FAILURE);
    mark(2);
    // ---- stage 2: scatter (depth bits, index) into the tiles' slices
    if (few_views)
        bin_kernel<true, BIN_THREADS_SMALL><<<dim3(L.n_chunks, n_views), BIN_THREADS_SMALL, lds, stream>>>(
            bin_table, N, grid_x, L.tiles, W, H, vis, B.vis_words, verdict_groups, bl);
    else
        bin_kernel<true, BIN_THREADS><<<dim3(L.n_chunks, n_views), BIN_THREADS, lds, stream>>>(
            bin_table, N, grid_x, L.tiles, W, H, vis, B.vis_words, verdict_groups, bl);
    mark(3);
    // ---- stage 3: work order (XCD streams, longest lists first) + per-tile (depth, index) sort
    const dim3 og((L.tiles + 255) / 256, n_views);
    const int items = n_views * L.tiles;
    const size_t qs = (size_t)items;              // queue stride
    const bool merge_long = n_views <= SMALL_BATCH_VIEWS;
    order_scatter_kernel<<<og, 256, 0, stream>>>(view_table, L.tiles, L.grid_x, order_state, work_order, sort_queue,
                                                 (uint32_t)items, merge_long ? 1 : 0, layers ? 1 : 0);
    const uint32_t* n_queue = order_state + ORDER_BINS;
    uint32_t* const n_open = order_state + ORDER_BINS + 5;
    if (!merge_long) {
        // 8193..15872 keys: the windowed sort (two workgroups per CU); longer: the split pre-pass, whose depth segments the
        // 512 x 16 tier's kernel sorts from the segment queue; what either rejects joins the open-ended kernel's queue
        // (one launch: its first workgroups run the split pre-pass, the windowed sort fills the chip beside them)
        const int part_blocks = std::min(items, 1024);          // every third workgroup: 3 x part_blocks in all, two thirds sort
        tile_sort_window_kernel<<<3 * part_blocks, SORT_WINDOW_THREADS, 0, stream>>>(
            bin_table, L.tiles, sort_queue + 3 * qs, n_queue + 3, sort_queue + 5 * qs, n_open, (uint32_t)part_blocks,
            sort_queue + 4 * qs, n_queue + 4, sort_queue + SORT_TIERS * qs, order_state + ORDER_SEG_WORD, (uint32_t)B.seg_cap);
    }
    // the open-ended kernel: every long list of a one- / two-view call; in a batch only what the two kernels above rejected
    // (piled-up depths: rare) -- a handful of workgroups then, 512 of them cost 20 us to find an empty queue
    tile_sort_long_kernel<1024, 16, true><<<merge_long ? std::min(items, 512) : 32, 1024, 0, stream>>>(
        bin_table, L.tiles, sort_queue + 5 * qs, n_queue + 5);
    if (!merge_long) {
        // 4097..8192 keys: 512 threads x 16 keys over 3584 buckets = 80 KiB, TWO workgroups per CU (round 5: the position-owned
        // ranking needs no index image, so the bucket count is free; 1024 x 8 over 8192 buckets = 96 KiB held a CU alone);
        // the same launch sorts the split pre-pass's segments behind its own lists
        tile_sort_long_kernel<512, 16, false, SORT_T2_BUCKETS, 4, true><<<std::min(items, 2048), 512, 0, stream>>>(
            bin_table, L.tiles, sort_queue + 2 * qs, n_queue + 2, sort_queue + SORT_TIERS * qs, order_state + ORDER_SEG_WORD,
            (uint32_t)B.seg_cap);
        tile_sort_long_kernel<512, 8, false><<<std::min(items, 2048), 512, 0, stream>>>(
            bin_table, L.tiles, sort_queue + qs, n_queue + 1);
    }
    tile_sort_kernel<<<items, SORT_THREADS, 0, stream>>>(bin_table, L.tiles, sort_queue, n_queue);
    mark(4);
    // ---- stage 4: compositing of every (view, tile, quarter) work item in ONE launch; with `semantic` the same
    // walk also produces the objects-only semantic image
    const uint32_t items_per_view = ITEMS_PER_TILE * (uint32_t)L.tiles;
    const uint32_t slots = (uint32_t)B.order_slots;
    SemanticDev sd{nullptr, nullptr, nullptr, 0, 0, nullptr, 0.0f, layer_tiles};
    if (want_sem) {
        const uint8_t* ids_u8 = semantic->object_id_u8;      // (a per-scene constant: pgr_scene_prepare packs it once)
        if (!ids_u8 && semantic->k_objects <= 255 && semantic->n_env < N) {
            auto* dst = reinterpret_cast<uint8_t*>(ws + B.obj_u8);
            const int n_obj = N - semantic->n_env;
            pack_object_ids_kernel<<<(n_obj + 255) / 256, 256, 0, stream>>>(semantic->object_id, semantic->n_env, N, dst);
            ids_u8 = dst;
        }
        sd = SemanticDev{semantic->object_id, ids_u8, semantic->colors, semantic->n_env, semantic->k_objects,
                         semantic->mask_colors, semantic->mask_threshold, layer_tiles};
    }
    if (layers) {
        sd.mask_colors = layers->mask_colors; sd.mask_thr = layers->mask_threshold; sd.k = n_layers;
        // empty (layer, tile) lists have no work item: their pixels hold the background's verdict
        layer_mask_fill_kernel<<<dim3(LAYER_FILL_BLOCKS, n_layers, n_views), 256, 0, stream>>>(
            view_table, layers->mask_colors, layers->mask_threshold, P, layers->layer_id, N);
        launch_composite<false, false, true>(slots, stream, view_table, items_per_view, work_order, sd);
    } else if (want_aux && want_sem)
        launch_composite<true, true>(slots, stream, view_table, items_per_view, work_order, sd);
    else if (want_aux)
        launch_composite<true, false>(slots, stream, view_table, items_per_view, work_order, sd);
    else if (want_sem)
        launch_composite<false, true>(slots, stream, view_table, items_per_view, work_order, sd);
    else
        launch_composite<false, false>(slots, stream, view_table, items_per_view, work_order, sd);
    mark(5);
    if (!hip_ok(hipGetLastError(), "kernel launch")) return fail(PGR_ERR_LAUNCH_FAILURE);

    if (status_event) return PGR_OK;       // the status words reached the host behind the scan
    // the only host read of the batch: instance counts + overflow flags, after everything is enqueued
    if (!hip_ok(hipMemcpyAsync(h_status, status_dev, (size_t)n_views * 8, hipMemcpyDeviceToHost, stream), "memcpy status"))
        return fail(PGR_ERR_LAUNCH_FAILURE);
    if (host_scratch) return PGR_OK;       // asynchronous: the caller synchronises and calls pgr_batch_status
    if (!hip_ok(hipStreamSynchronize(stream), "sync at batch end")) return PGR_ERR_LAUNCH_FAILURE;
    bool overflow = false;
    for (int v = 0; v < n_views; ++v) {
        if (num_instances) num_instances[v] = (int64_t)h_status[2 * v];
        overflow = overflow || h_status[2 * v + 1];
    }
    return overflow ? PGR_ERR_INSTANCE_OVERFLOW : PGR_OK;
}

}  // namespace pgr

using namespace pgr;

extern "C" {

int32_t pgr_abi_version(void) { return PGR_ABI_VERSION; }
#ifndef PGR_SOURCE_HASH
#define PGR_SOURCE_HASH "unstamped"      // pegasus_amd/build.py passes the hash of the sources it compiled
#endif
const char* pgr_version(void) { return "pegasus_raster 0.9 (gfx950) src " PGR_SOURCE_HASH; }

const char* pgr_status_string(int32_t status) {
    switch (status) {
        case PGR_OK: return "ok";
        case PGR_ERR_INVALID_ARGUMENT: return "invalid argument";
        case PGR_ERR_WORKSPACE_TOO_SMALL: return "workspace too small";
        case PGR_ERR_INSTANCE_OVERFLOW: return "instance buffer overflow";
        case PGR_ERR_LAUNCH_FAILURE: return "HIP launch failure";
        case PGR_ERR_NO_DEVICE: return "no HIP device";
        default: return "unknown status";
    }
}

const char* pgr_last_hip_error(void) { return g_hip_error; }

size_t pgr_batch_workspace_bytes(int32_t n, int32_t width, int32_t height, int64_t max_instances, int32_t n_views) {
    if (n < 0 || width <= 0 || height <= 0 || max_instances < 0 || max_instances > 0x7fffffffLL || n_views <= 0)
        return 0;
    return make_batch_layout(make_layout(n, width, height, max_instances), n_views, (size_t)n).total;
}

size_t pgr_workspace_bytes(int32_t n, int32_t width, int32_t height, int64_t max_instances) {
    return pgr_batch_workspace_bytes(n, width, height, max_instances, 1);
}

int32_t pgr_workspace_view(void* workspace, size_t workspace_bytes, int32_t n, int32_t width, int32_t height,
                           int64_t max_instances, int32_t n_views, int32_t view_index, PgrWorkspaceView* v) {
    if (!workspace || !v || n < 0 || width <= 0 || height <= 0 || max_instances < 0 || n_views <= 0 ||
        view_index < 0 || view_index >= n_views)
        return PGR_ERR_INVALID_ARGUMENT;
    const Layout L = make_layout(n, width, height, max_instances);
    const BatchLayout B = make_batch_layout(L, n_views, (size_t)n);
    if (workspace_bytes < B.total) return PGR_ERR_WORKSPACE_TOO_SMALL;
    const ViewWs w = carve(static_cast<char*>(workspace) + B.views + (size_t)view_index * B.per_view, L);
    v->splats = reinterpret_cast<const float*>(w.splats);
    v->rects = reinterpret_cast<const uint16_t*>(w.rects);
    v->gauss_sorted = w.gauss_sorted;
    v->ranges = reinterpret_cast<const uint32_t*>(w.ranges);
    v->num_instances = w.counters;
    return PGR_OK;
}

int32_t pgr_forward(const PgrScene* scene, const PgrCamera* cam, const PgrOutputs* out, void* workspace,
                    size_t workspace_bytes, int64_t max_instances, int64_t* num_instances, void* stream_v) {
    if (scene && scene->n > 0 && out && !out->radii) return PGR_ERR_INVALID_ARGUMENT;
    return forward_batch_impl(scene, 1, cam, out, workspace, workspace_bytes, max_instances, num_instances,
                              static_cast<hipStream_t>(stream_v), nullptr);
}

int32_t pgr_forward_batch(const PgrScene* scene, int32_t n_views, const PgrCamera* cameras, const PgrOutputs* outs,
                          void* workspace, size_t workspace_bytes, int64_t max_instances_per_view,
                          int64_t* num_instances, void* stream_v) {
    return forward_batch_impl(scene, n_views, cameras, outs, workspace, workspace_bytes, max_instances_per_view,
                              num_instances, static_cast<hipStream_t>(stream_v), nullptr);
}

size_t pgr_host_scratch_bytes(int32_t n_views) { return n_views > 0 ? host_scratch_bytes(n_views) : 0; }

int32_t pgr_forward_batch_async(const PgrScene* scene, int32_t n_views, const PgrCamera* cameras, const PgrOutputs* outs,
                                void* workspace, size_t workspace_bytes, int64_t max_instances_per_view,
                                void* host_scratch, size_t host_scratch_size, void* stream_v) {
    if (!host_scratch || n_views <= 0 || host_scratch_size < host_scratch_bytes(n_views)) return PGR_ERR_INVALID_ARGUMENT;
    if (scene && scene->n == 0) memset(static_cast<char*>(host_scratch), 0, host_scratch_bytes(n_views));
    return forward_batch_impl(scene, n_views, cameras, outs, workspace, workspace_bytes, max_instances_per_view, nullptr,
                              static_cast<hipStream_t>(stream_v), nullptr, host_scratch);
}

int32_t pgr_forward_frames_async(const PgrScene* scene, const PgrSemantic* semantic, int32_t n_views,
                                 const PgrCamera* cameras, const PgrOutputs* outs, void* workspace,
                                 size_t workspace_bytes, int64_t max_instances_per_view, void* host_scratch,
                                 size_t host_scratch_size, void* stream_v) {
    if (!host_scratch || n_views <= 0 || host_scratch_size < host_scratch_bytes(n_views)) return PGR_ERR_INVALID_ARGUMENT;
    if (scene && scene->n == 0) memset(static_cast<char*>(host_scratch), 0, host_scratch_bytes(n_views));
    return forward_batch_impl(scene, n_views, cameras, outs, workspace, workspace_bytes, max_instances_per_view, nullptr,
                              static_cast<hipStream_t>(stream_v), nullptr, host_scratch, semantic);
}

int32_t pgr_forward_posed_async(const PgrScene* scene, const PgrSemantic* semantic, const PgrPosedObjects* posed,
                                int32_t n_views, const PgrCamera* cameras, const PgrOutputs* outs, void* workspace,
                                size_t workspace_bytes, int64_t max_instances_per_view, void* host_scratch,
                                size_t host_scratch_size, void* stream_v) {
    if (!host_scratch || n_views <= 0 || host_scratch_size < host_scratch_bytes(n_views)) return PGR_ERR_INVALID_ARGUMENT;
    if (scene && scene->n == 0) memset(static_cast<char*>(host_scratch), 0, host_scratch_bytes(n_views));
    return forward_batch_impl(scene, n_views, cameras, outs, workspace, workspace_bytes, max_instances_per_view, nullptr,
                              static_cast<hipStream_t>(stream_v), nullptr, host_scratch, semantic, posed);
}

int32_t pgr_forward_posed_early_status(const PgrScene* scene, const PgrSemantic* semantic, const PgrPosedObjects* posed,
                                       int32_t n_views, const PgrCamera* cameras, const PgrOutputs* outs, void* workspace,
                                       size_t workspace_bytes, int64_t max_instances_per_view, void* host_scratch,
                                       size_t host_scratch_size, void* stream_v, void* status_event) {
    if (!host_scratch || !status_event || n_views <= 0 || host_scratch_size < host_scratch_bytes(n_views))
        return PGR_ERR_INVALID_ARGUMENT;
    if (scene && scene->n == 0) memset(static_cast<char*>(host_scratch), 0, host_scratch_bytes(n_views));
    return forward_batch_impl(scene, n_views, cameras, outs, workspace, workspace_bytes, max_instances_per_view, nullptr,
                              static_cast<hipStream_t>(stream_v), nullptr, host_scratch, semantic, posed, nullptr,
                              static_cast<hipEvent_t>(status_event));
}

size_t pgr_layers_workspace_bytes(int32_t n, int32_t width, int32_t height, int64_t max_instances, int32_t n_views,
                                  int32_t n_layers) {
    if (n < 0 || width <= 0 || height <= 0 || max_instances < 0 || max_instances > 0x7fffffffLL || n_views <= 0 ||
        n_layers <= 0 || n_layers > 4096)
        return 0;
    return make_batch_layout(make_layout(n, width, height, max_instances, n_layers), n_views, (size_t)n).total;
}

int32_t pgr_forward_layers_async(const PgrScene* scene, const PgrLayers* layers, const PgrPosedObjects* posed,
                                 int32_t n_views, const PgrCamera* cameras, const PgrOutputs* outs, void* workspace,
                                 size_t workspace_bytes, int64_t max_instances_per_view, void* host_scratch,
                                 size_t host_scratch_size, void* stream_v) {
    if (!layers || !host_scratch || n_views <= 0 || host_scratch_size < host_scratch_bytes(n_views))
        return PGR_ERR_INVALID_ARGUMENT;
    if (scene && scene->n == 0) memset(static_cast<char*>(host_scratch), 0, host_scratch_bytes(n_views));
    return forward_batch_impl(scene, n_views, cameras, outs, workspace, workspace_bytes, max_instances_per_view, nullptr,
                              static_cast<hipStream_t>(stream_v), nullptr, host_scratch, nullptr, posed, layers);
}

size_t pgr_scene_cache_bytes(int32_t n) { return n < 0 ? 0 : align_up((size_t)n * 4) + align_up((size_t)n); }

int32_t pgr_scene_prepare(const PgrScene* scene, const PgrSemantic* semantic, void* cache, size_t cache_bytes,
                          const uint32_t** tie_inv, const uint8_t** object_id_u8, void* stream_v) {
    if (!scene || scene->n < 0 || !tie_inv || !object_id_u8) return PGR_ERR_INVALID_ARGUMENT;
    *tie_inv = nullptr;
    *object_id_u8 = nullptr;
    const int N = scene->n;
    if (N == 0) return PGR_OK;
    if (!cache) return PGR_ERR_INVALID_ARGUMENT;
    if (cache_bytes < pgr_scene_cache_bytes(N)) return PGR_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    char* c = static_cast<char*>(cache);
    if (scene->tie_index) {
        auto* inv = reinterpret_cast<uint32_t*>(c);
        invert_tie_index_kernel<<<(N + 255) / 256, 256, 0, stream>>>(N, scene->tie_index, inv);
        *tie_inv = inv;
    }
    if (semantic && semantic->object_id && semantic->k_objects > 0 && semantic->k_objects <= 255 && semantic->n_env >= 0 &&
        semantic->n_env < N) {
        auto* ids = reinterpret_cast<uint8_t*>(c + align_up((size_t)N * 4));
        const int n_obj = N - semantic->n_env;
        pack_object_ids_kernel<<<(n_obj + 255) / 256, 256, 0, stream>>>(semantic->object_id, semantic->n_env, N, ids);
        *object_id_u8 = ids;
    }
    return hip_ok(hipGetLastError(), "scene_prepare launch") ? PGR_OK : PGR_ERR_LAUNCH_FAILURE;
}

int32_t pgr_batch_status(const void* host_scratch, int32_t n_views, int64_t* num_instances) {
    if (!host_scratch || n_views <= 0) return PGR_ERR_INVALID_ARGUMENT;
    const size_t tables = host_scratch_bytes(n_views) - (size_t)n_views * 8;   // status words follow the tables
    const uint32_t* st = reinterpret_cast<const uint32_t*>(static_cast<const char*>(host_scratch) + tables);
    bool overflow = false;
    for (int v = 0; v < n_views; ++v) {
        if (num_instances) num_instances[v] = (int64_t)st[2 * v];
        overflow = overflow || st[2 * v + 1];
    }
    return overflow ? PGR_ERR_INSTANCE_OVERFLOW : PGR_OK;
}

int32_t pgr_forward_batch_profiled(const PgrScene* scene, const PgrSemantic* semantic, int32_t n_views,
                                   const PgrCamera* cameras, const PgrOutputs* outs, void* workspace,
                                   size_t workspace_bytes, int64_t max_instances_per_view, int64_t* num_instances,
                                   void* stream_v, float* stage_ms) {
    if (!stage_ms) return PGR_ERR_INVALID_ARGUMENT;
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    hipEvent_t ev[PGR_NUM_STAGES + 1];
    for (auto& e : ev)
        if (!hip_ok(hipEventCreate(&e), "hipEventCreate")) return PGR_ERR_LAUNCH_FAILURE;
    for (int k = 0; k < PGR_NUM_STAGES; ++k) stage_ms[k] = 0.f;
    int32_t rc = forward_batch_impl(scene, n_views, cameras, outs, workspace, workspace_bytes, max_instances_per_view,
                                    num_instances, stream, ev, nullptr, semantic);
    if (rc == PGR_OK && scene->n > 0) {
        for (int k = 0; rc == PGR_OK && k < PGR_NUM_STAGES; ++k)
            if (!hip_ok(hipEventElapsedTime(&stage_ms[k], ev[k], ev[k + 1]), "hipEventElapsedTime"))
                rc = PGR_ERR_LAUNCH_FAILURE;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

int32_t pgr_backward(const PgrScene* scene, const PgrCamera* cam, const float* grad_color, const float* grad_depth,
                     const float* final_T, const uint32_t* n_contrib, const int32_t* radii, void* workspace,
                     size_t workspace_bytes, int64_t max_instances, const PgrGradOutputs* grads, float* grad_rows,
                     void* stream_v) {
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    if (int rc = check_scene(scene)) return rc;
    if (scene->shs_rest) return PGR_ERR_INVALID_ARGUMENT;      // the SH gradient is one [n,sh_stride,3] array
    if (!cam || !grads || !grad_color || !final_T || !n_contrib || cam->image_width <= 0 || cam->image_height <= 0)
        return PGR_ERR_INVALID_ARGUMENT;
    const int N = scene->n, W = cam->image_width, H = cam->image_height;
    if (N == 0) return PGR_OK;
    if (!workspace || !grad_rows || !radii) return PGR_ERR_INVALID_ARGUMENT;
    const Layout L = make_layout(N, W, H, max_instances);
    const BatchLayout B = make_batch_layout(L, 1, (size_t)N);
    if (workspace_bytes < B.total) return PGR_ERR_WORKSPACE_TOO_SMALL;
    char* ws = static_cast<char*>(workspace);
    const ViewWs vw = carve(ws + B.views, L);
    const CameraDev* camd = reinterpret_cast<const CameraDev*>(ws + B.cams);
    if (!hip_ok(hipMemsetAsync(grad_rows, 0, (size_t)N * GRAD_ROW * sizeof(float), stream), "memset grad rows"))
        return PGR_ERR_LAUNCH_FAILURE;
    // the forward's work order is still in the workspace (one view: item = 4 * tile + quarter)
    composite_backward_block_kernel<<<4u * (uint32_t)B.order_slots, WAVE, 0, stream>>>(
        camd, vw.ranges, vw.gauss_sorted, vw.splats, final_T, n_contrib, grad_color, grad_depth, grad_rows,
        reinterpret_cast<const uint32_t*>(ws + B.work_order));
    const GradOut go{grads->means2d, grads->means3d, grads->opacities, grads->colors, grads->shs, grads->cov3d,
                     grads->scales, grads->rotations};
    const int blocks = (N + 255) / 256;
    switch (scene->shs ? scene->sh_degree : 0) {
        case 0: preprocess_backward_kernel<0><<<blocks, 256, 0, stream>>>(*scene, camd, radii, grad_rows, go); break;
        case 1: preprocess_backward_kernel<1><<<blocks, 256, 0, stream>>>(*scene, camd, radii, grad_rows, go); break;
        case 2: preprocess_backward_kernel<2><<<blocks, 256, 0, stream>>>(*scene, camd, radii, grad_rows, go); break;
        default: preprocess_backward_kernel<3><<<blocks, 256, 0, stream>>>(*scene, camd, radii, grad_rows, go); break;
    }
    return hip_ok(hipGetLastError(), "backward launch") ? PGR_OK : PGR_ERR_LAUNCH_FAILURE;
}

int32_t pgr_compose_object(int32_t n, const float* xyz, const float* rot, const float* f_rest, int32_t n_rest,
                           int32_t in_rest_stride, const PgrObjectPose* pose, float* out_xyz, float* out_rot,
                           float* out_rest, int32_t out_rest_stride, void* stream_v) {
    static_assert(sizeof(PgrObjectPose) == sizeof(ObjectPoseDev), "pose layout");
    if (n < 0 || !pose || (n > 0 && (!xyz || !out_xyz)) || (n_rest != 0 && n_rest != 3 && n_rest != 8 && n_rest != 15) ||
        (f_rest && n_rest > 0 && (in_rest_stride < 3 * n_rest || out_rest_stride < 3 * n_rest)))
        return PGR_ERR_INVALID_ARGUMENT;
    if (n == 0) return PGR_OK;
    ObjectPoseDev P;
    memcpy(&P, pose, sizeof(P));
    compose_object_kernel<<<(n + 255) / 256, 256, 0, static_cast<hipStream_t>(stream_v)>>>(
        n, xyz, rot, f_rest, n_rest, in_rest_stride, P, out_xyz, out_rot, out_rest, out_rest_stride);
    return hip_ok(hipGetLastError(), "compose_object launch") ? PGR_OK : PGR_ERR_LAUNCH_FAILURE;
}

// one wave that watches both of its clocks for spin_us microseconds: s_memtime ticks once per SHADER cycle, s_memrealtime
// at a constant 100 MHz (MI355X_MICROARCH.md "Per-instruction cycle constants"), so ticks[0] / ticks[1] x 100 MHz is the
// clock the chip ran at while whatever else was resident ran beside it; ticks[2], ticks[3] = the 100 MHz counter at the
// start and at the end (two probes on two streams can be checked for having overlapped)
__global__ void clock_probe_kernel(unsigned long long* __restrict__ ticks, uint32_t spin_us) {
    const unsigned long long r0 = wall_clock64();
    const unsigned long long c0 = __builtin_readcyclecounter();
    unsigned long long r1 = r0;
    while (r1 - r0 < (unsigned long long)spin_us * 100ull) {
        __builtin_amdgcn_s_sleep(8);
        r1 = wall_clock64();
    }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { ticks[0] = c1 - c0; ticks[1] = r1 - r0; ticks[2] = r0; ticks[3] = r1; }
}

int32_t pgr_clock_probe(uint64_t* ticks, uint32_t spin_us, void* stream_v) {
    if (!ticks || spin_us > 1000000u) return PGR_ERR_INVALID_ARGUMENT;
    clock_probe_kernel<<<1, WAVE, 0, static_cast<hipStream_t>(stream_v)>>>(reinterpret_cast<unsigned long long*>(ticks), spin_us);
    return hip_ok(hipGetLastError(), "clock_probe launch") ? PGR_OK : PGR_ERR_LAUNCH_FAILURE;
}

size_t pgr_pose_objects_workspace_bytes(int32_t n_jobs) {
    if (n_jobs <= 0) return 0;
    const size_t launches = ((size_t)n_jobs + POSE_JOBS_PER_LAUNCH - 1) / POSE_JOBS_PER_LAUNCH;
    return launches * (align_up((size_t)POSE_JOBS_PER_LAUNCH * POSE_REDUCE_BLOCKS * 3 * sizeof(double)) +
                       align_up((size_t)POSE_JOBS_PER_LAUNCH * sizeof(ObjectPoseDev)));
}

int32_t pgr_pose_objects(int32_t n_jobs, const PgrPoseJob* jobs, const double* sh_dirs, const double* sh_pinv,
                         void* workspace, size_t workspace_bytes, void* stream_v) {
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    if (n_jobs < 0 || (n_jobs > 0 && !jobs)) return PGR_ERR_INVALID_ARGUMENT;
    if (n_jobs == 0) return PGR_OK;
    for (int k = 0; k < n_jobs; ++k) {
        const PgrPoseJob& j = jobs[k];
        if (j.n < 0 || (j.n > 0 && (!j.src || !j.dst)) || j.kind < PGR_POSE_XYZ || j.kind > PGR_POSE_SH ||
            (j.kind == PGR_POSE_SH && ((j.n_rest != 3 && j.n_rest != 8 && j.n_rest != 15) || !sh_dirs || !sh_pinv)))
            return PGR_ERR_INVALID_ARGUMENT;
    }
    if (!workspace || workspace_bytes < pgr_pose_objects_workspace_bytes(n_jobs)) return PGR_ERR_WORKSPACE_TOO_SMALL;
    char* ws = static_cast<char*>(workspace);
    const size_t part_bytes = align_up((size_t)POSE_JOBS_PER_LAUNCH * POSE_REDUCE_BLOCKS * 3 * sizeof(double));
    const size_t per_launch = part_bytes + align_up((size_t)POSE_JOBS_PER_LAUNCH * sizeof(ObjectPoseDev));
    for (int k0 = 0, launch = 0; k0 < n_jobs; k0 += POSE_JOBS_PER_LAUNCH, ++launch) {
        PoseJobTable T{};
        T.count = std::min(POSE_JOBS_PER_LAUNCH, n_jobs - k0);
        uint32_t blocks = 0;
        bool reduce = false;
        for (int k = 0; k < T.count; ++k) {
            const PgrPoseJob& j = jobs[k0 + k];
            T.job[k] = PoseJobDev{j.src, j.dst, j.R, j.t, j.n, j.kind, j.n_rest, j.about_origin,
                                  j.R_row_stride > 0 ? j.R_row_stride : 3, j.t_stride > 0 ? j.t_stride : 1, blocks};
            blocks += (uint32_t)((j.n + 255) / 256);
            reduce = reduce || (j.kind == PGR_POSE_XYZ && !j.about_origin && j.R && j.n > 0);
        }
        auto* partial = reinterpret_cast<double*>(ws + (size_t)launch * per_launch);
        auto* poses = reinterpret_cast<ObjectPoseDev*>(ws + (size_t)launch * per_launch + part_bytes);
        if (reduce) pose_reduce_kernel<<<dim3(POSE_REDUCE_BLOCKS, T.count), 256, 0, stream>>>(T, partial);
        pose_prepare_kernel<<<T.count, 128, 0, stream>>>(T, partial, sh_dirs, sh_pinv, poses);
        if (blocks) pose_apply_kernel<<<blocks, 256, 0, stream>>>(T, poses);
    }
    return hip_ok(hipGetLastError(), "pose_objects launch") ? PGR_OK : PGR_ERR_LAUNCH_FAILURE;
}

size_t pgr_block_visibility_workspace_bytes(int32_t n, int32_t n_views) {
    if (n < 0 || n_views <= 0) return 0;
    return align_up((size_t)n_views * sizeof(CameraDev)) + align_up((size_t)((n + WAVE - 1) / WAVE) * sizeof(BlockBounds) + 1);
}

int32_t pgr_block_visibility(const PgrScene* scene, int32_t n_views, const PgrCamera* cams, void* workspace,
                             size_t workspace_bytes, uint32_t* vis_words, void* stream_v) {
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    if (int rc = check_scene(scene)) return rc;
    if (n_views <= 0 || !cams) return PGR_ERR_INVALID_ARGUMENT;
    if (scene->n == 0) return PGR_OK;
    if (!workspace || !vis_words) return PGR_ERR_INVALID_ARGUMENT;
    if (workspace_bytes < pgr_block_visibility_workspace_bytes(scene->n, n_views)) return PGR_ERR_WORKSPACE_TOO_SMALL;
    const int W = cams[0].image_width, H = cams[0].image_height;
    for (int v = 0; v < n_views; ++v)
        if (cams[v].image_width != W || cams[v].image_height != H || W <= 0 || H <= 0 || !cams[v].viewmatrix ||
            !cams[v].projmatrix || !cams[v].campos || !cams[v].bg || !(cams[v].tanfovx > 0.f) || !(cams[v].tanfovy > 0.f))
            return PGR_ERR_INVALID_ARGUMENT;
    char* ws = static_cast<char*>(workspace);
    auto* cams_dev = reinterpret_cast<CameraDev*>(ws);
    auto* bounds = reinterpret_cast<BlockBounds*>(ws + align_up((size_t)n_views * sizeof(CameraDev)));
    for (int v0 = 0; v0 < n_views; v0 += CAM_PACK_MAX) {
        CamPack cp;
        const int cnt = std::min(CAM_PACK_MAX, n_views - v0);
        for (int k = 0; k < cnt; ++k) {
            const PgrCamera& c = cams[v0 + k];
            cp.view[k] = c.viewmatrix; cp.proj[k] = c.projmatrix; cp.campos[k] = c.campos; cp.bg[k] = c.bg;
            cp.tanfovx[k] = c.tanfovx; cp.tanfovy[k] = c.tanfovy; cp.depth_mode[k] = 0;
        }
        pack_camera_kernel<<<cnt, 64, 0, stream>>>(cp, W, H, cams_dev + v0);
    }
    const int groups = (scene->n + WAVE - 1) / WAVE, words = (n_views + 31) / 32;
    block_bounds_kernel<<<(groups + 3) / 4, 256, 0, stream>>>(*scene, nullptr, bounds, groups);
    block_cull_kernel<<<dim3((groups + 7) / 8, words), 256, 0, stream>>>(bounds, groups, cams_dev, n_views, words, vis_words);
    return hip_ok(hipGetLastError(), "block_visibility launch") ? PGR_OK : PGR_ERR_LAUNCH_FAILURE;
}

int32_t pgr_mark_visible(int32_t n, const float* means3d, const float* viewmatrix, uint8_t* present, void* stream_v) {
    if (n < 0 || (n > 0 && (!means3d || !viewmatrix || !present))) return PGR_ERR_INVALID_ARGUMENT;
    if (n == 0) return PGR_OK;
    mark_visible_kernel<<<(n + 255) / 256, 256, 0, static_cast<hipStream_t>(stream_v)>>>(n, means3d, viewmatrix,
                                                                                         present);
    return hip_ok(hipGetLastError(), "mark_visible launch") ? PGR_OK : PGR_ERR_LAUNCH_FAILURE;
}

int32_t pgr_color_masks(const float* img_chw, int32_t n_images, int32_t width, int32_t height, const float* colors_k3,
                        int32_t k, float threshold, uint8_t* masks_khw, void* stream_v) {
    if (!img_chw || n_images < 0 || n_images > 65535 || width <= 0 || height <= 0 || k < 0 ||
        (k > 0 && (!colors_k3 || !masks_khw)))
        return PGR_ERR_INVALID_ARGUMENT;
    if (k == 0 || n_images == 0) return PGR_OK;
    const size_t P = (size_t)width * height;
    color_masks_kernel<<<dim3((unsigned)((P + 255) / 256), n_images), 256, 0, static_cast<hipStream_t>(stream_v)>>>(
        img_chw, P, colors_k3, k, threshold, masks_khw);
    return hip_ok(hipGetLastError(), "color_masks launch") ? PGR_OK : PGR_ERR_LAUNCH_FAILURE;
}

int32_t pgr_quantize_frame(const float* img_chw, const float* depth_hw, int32_t width, int32_t height,
                           uint8_t* rgb_hwc, uint16_t* depth_mm_hw, void* stream_v) {
    if (width <= 0 || height <= 0 || ((img_chw == nullptr) != (rgb_hwc == nullptr)) ||
        ((depth_hw == nullptr) != (depth_mm_hw == nullptr)))
        return PGR_ERR_INVALID_ARGUMENT;
    const size_t P = (size_t)width * height;
    quantize_kernel<<<(unsigned)((P + 255) / 256), 256, 0, static_cast<hipStream_t>(stream_v)>>>(
        img_chw, depth_hw, P, rgb_hwc, depth_mm_hw);
    return hip_ok(hipGetLastError(), "quantize launch") ? PGR_OK : PGR_ERR_LAUNCH_FAILURE;
}

int32_t pgr_pack_frames(const float* color_b3hw, const float* depth_bhw, const uint8_t* masks_bkhw, int32_t n_images,
                        int32_t k, int32_t width, int32_t height, uint8_t* rgb_bhwc, uint16_t* depth_mm_bhw,
                        uint8_t* mask_bits_bhwj, void* stream_v) {
    if (n_images < 0 || n_images > 65535 || width <= 0 || height <= 0 || k < 0 ||
        ((color_b3hw == nullptr) != (rgb_bhwc == nullptr)) || ((depth_bhw == nullptr) != (depth_mm_bhw == nullptr)) ||
        ((masks_bkhw == nullptr) != (mask_bits_bhwj == nullptr)) || (masks_bkhw && k == 0))
        return PGR_ERR_INVALID_ARGUMENT;
    if (n_images == 0) return PGR_OK;
    const size_t P = (size_t)width * height;
    pack_frames_kernel<<<dim3((unsigned)((P + 255) / 256), n_images), 256, 0, static_cast<hipStream_t>(stream_v)>>>(
        color_b3hw, depth_bhw, masks_bkhw, P, k, rgb_bhwc, depth_mm_bhw, mask_bits_bhwj);
    return hip_ok(hipGetLastError(), "pack_frames launch") ? PGR_OK : PGR_ERR_LAUNCH_FAILURE;
}

int32_t pgr_frame_record_layout(int32_t width, int32_t height, int32_t k, PgrRecordLayout* layout) {
    if (!layout || width <= 0 || height <= 0 || k < 0) return PGR_ERR_INVALID_ARGUMENT;
    const size_t P = (size_t)width * height;
    layout->off_rgb = 0;
    layout->off_depth = (int64_t)align_up(3 * P, 16);
    layout->off_masks = layout->off_depth + (int64_t)align_up(2 * P, 16);
    layout->bytes = layout->off_masks + (int64_t)align_up((size_t)((k + 7) / 8) * P, 16);
    return PGR_OK;
}

int32_t pgr_pack_records(const float* color_b3hw, const float* depth_bhw, const uint8_t* masks_bkhw, int32_t n_images,
                         int32_t k, int32_t width, int32_t height, uint8_t* records, int64_t record_stride,
                         void* stream_v) {
    PgrRecordLayout L;
    if (pgr_frame_record_layout(width, height, k, &L) != PGR_OK || n_images < 0 || n_images > 65535 || !records ||
        record_stride < L.bytes || (record_stride & 15) || (masks_bkhw && k == 0))
        return PGR_ERR_INVALID_ARGUMENT;
    if (n_images == 0) return PGR_OK;
    const size_t P = (size_t)width * height;
    pack_records_kernel<<<dim3((unsigned)((P + 255) / 256), n_images), 256, 0, static_cast<hipStream_t>(stream_v)>>>(
        color_b3hw, depth_bhw, masks_bkhw, P, k, records, (size_t)record_stride, (size_t)L.off_depth, (size_t)L.off_masks);
    return hip_ok(hipGetLastError(), "pack_records launch") ? PGR_OK : PGR_ERR_LAUNCH_FAILURE;
}

}  // extern "C"





// ---- 3-nearest-neighbour mean squared distance (simple_knn.distCUDA2) ---------------------------------------------
namespace {
struct KnnLayout { size_t grid, count, start, sorted, total; int target; size_t cells; };
KnnLayout knn_layout(int32_t n) {
    KnnLayout K{};
    int target = 1;
    while (target < KNN_MAX_GRID && (double)target * target * target < 0.5 * (double)n) ++target;   // ~2 points per cell
    K.target = target;
    K.cells = (size_t)target * target * target;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes ? bytes : 1); return o; };
    K.grid = take(sizeof(KnnGrid));
    K.count = take(K.cells * 4);
    K.start = take((K.cells + 1) * 4);
    K.sorted = take((size_t)(n > 0 ? n : 1) * 16);
    K.total = off;
    return K;
}
}  // namespace

size_t pgr_knn_workspace_bytes(int32_t n) { return n < 0 ? 0 : knn_layout(n).total; }

int32_t pgr_knn_mean_dist2(int32_t n, const float* xyz, float* out, void* workspace, size_t workspace_bytes,
                           void* stream_v) {
    if (n < 0) return PGR_ERR_INVALID_ARGUMENT;
    if (n == 0) return PGR_OK;
    if (!xyz || !out || !workspace) return PGR_ERR_INVALID_ARGUMENT;
    const KnnLayout K = knn_layout(n);
    if (workspace_bytes < K.total) return PGR_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    char* ws = static_cast<char*>(workspace);
    auto* grid = reinterpret_cast<KnnGrid*>(ws + K.grid);
    auto* count = reinterpret_cast<uint32_t*>(ws + K.count);
    auto* start = reinterpret_cast<uint32_t*>(ws + K.start);
    auto* sorted = reinterpret_cast<float4*>(ws + K.sorted);
    KnnGrid init{};
    for (int a = 0; a < 3; ++a) { init.lo[a] = 0xffffffffu; init.hi[a] = 0u; }
    if (!hip_ok(hipMemcpyAsync(grid, &init, sizeof(init), hipMemcpyHostToDevice, stream), "memcpy knn grid") ||
        !hip_ok(hipMemsetAsync(count, 0, K.cells * 4, stream), "memset knn counts"))
        return PGR_ERR_LAUNCH_FAILURE;
    const int blocks = (n + 255) / 256;
    knn_bbox_kernel<<<blocks, 256, 0, stream>>>(n, xyz, grid);
    knn_grid_kernel<<<1, 1, 0, stream>>>(grid, K.target);
    knn_count_kernel<<<blocks, 256, 0, stream>>>(n, xyz, grid, count);
    knn_scan_kernel<<<1, 1024, 0, stream>>>(grid, count, start);
    // the counts become the cursors
    if (!hip_ok(hipMemcpyAsync(count, start, K.cells * 4, hipMemcpyDeviceToDevice, stream), "memcpy knn cursors"))
        return PGR_ERR_LAUNCH_FAILURE;
    knn_scatter_kernel<<<blocks, 256, 0, stream>>>(n, xyz, grid, count, sorted);
    knn_search_kernel<<<blocks, 256, 0, stream>>>(n, grid, start, sorted, out);
    return hip_ok(hipGetLastError(), "knn launch") ? PGR_OK : PGR_ERR_LAUNCH_FAILURE;
}
