// pegasus_raster.hip -- C ABI of libpegasus_raster.so (see include/pegasus_raster.h).
// gfx950 only.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <rocprim/rocprim.hpp>

#include "binning.hip.h"
#include "composite.hip.h"
#include "pgr_common.h"
#include "preprocess.hip.h"

namespace pgr {

static thread_local char g_hip_error[256] = "";

static bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    snprintf(g_hip_error, sizeof(g_hip_error), "%s: %s", what, hipGetErrorString(e));
    return false;
}

static size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

constexpr size_t SORT_TEMP_FIXED = 32u << 20;  // histograms / look-back state of the device radix sort

static Layout make_layout(int32_t n, int32_t width, int32_t height, int64_t max_instances) {
    Layout L{};
    const size_t N = (size_t)(n > 0 ? n : 0), I = (size_t)(max_instances > 0 ? max_instances : 0);
    const int gx = (width + TILE - 1) / TILE, gy = (height + TILE - 1) / TILE;
    L.tiles = gx * gy;
    L.n_blocks = (int)((N + PRE_BLOCK - 1) / PRE_BLOCK);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes ? bytes : 1); return o; };
    L.cam = take(sizeof(CameraDev));
    L.counters = take(64);
    L.xy = take(N * 8);
    L.depth = take(N * 4);
    L.conic_opacity = take(N * 16);
    L.rgb = take(N * 16);
    L.tiles_touched = take(N * 4);
    L.offsets = take(N * 4);
    L.block_sums = take((size_t)L.n_blocks * 4);
    L.keys_unsorted = take(I * 8);
    L.vals_unsorted = take(I * 4);
    L.keys_sorted = take(I * 8);
    L.vals_sorted = take(I * 4);
    L.ranges = take((size_t)L.tiles * 8);
    L.sort_temp_bytes = SORT_TEMP_FIXED + I * 12;
    L.sort_temp = take(L.sort_temp_bytes);
    L.total = off;
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
    return PGR_OK;
}

}  // namespace pgr

using namespace pgr;

extern "C" {

int32_t pgr_abi_version(void) { return PGR_ABI_VERSION; }
const char* pgr_version(void) { return "pegasus_raster 0.1 (gfx950)"; }

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

size_t pgr_workspace_bytes(int32_t n, int32_t width, int32_t height, int64_t max_instances) {
    if (n < 0 || width <= 0 || height <= 0 || max_instances < 0 || max_instances > 0x7fffffffLL) return 0;
    return make_layout(n, width, height, max_instances).total;
}

int32_t pgr_workspace_view(void* workspace, size_t workspace_bytes, int32_t n, int32_t width, int32_t height,
                           int64_t max_instances, PgrWorkspaceView* v) {
    if (!workspace || !v || n < 0 || width <= 0 || height <= 0 || max_instances < 0) return PGR_ERR_INVALID_ARGUMENT;
    const Layout L = make_layout(n, width, height, max_instances);
    if (workspace_bytes < L.total) return PGR_ERR_WORKSPACE_TOO_SMALL;
    char* w = static_cast<char*>(workspace);
    v->xy = reinterpret_cast<const float*>(w + L.xy);
    v->depth = reinterpret_cast<const float*>(w + L.depth);
    v->conic_opacity = reinterpret_cast<const float*>(w + L.conic_opacity);
    v->rgb = reinterpret_cast<const float*>(w + L.rgb);
    v->tiles_touched = reinterpret_cast<const uint32_t*>(w + L.tiles_touched);
    v->offsets = reinterpret_cast<const uint32_t*>(w + L.offsets);
    v->keys_sorted = reinterpret_cast<const uint64_t*>(w + L.keys_sorted);
    v->gauss_sorted = reinterpret_cast<const uint32_t*>(w + L.vals_sorted);
    v->ranges = reinterpret_cast<const uint32_t*>(w + L.ranges);
    v->num_instances = reinterpret_cast<const uint32_t*>(w + L.counters);
    return PGR_OK;
}

}  // extern "C"

// ev: optional PGR_NUM_STAGES+1 events recorded at the stage boundaries (profiling entry point only)
static int32_t forward_impl(const PgrScene* scene, const PgrCamera* cam, const PgrOutputs* out, void* workspace,
                            size_t workspace_bytes, int64_t max_instances, int64_t* num_instances,
                            hipStream_t stream, hipEvent_t* ev) {
    auto mark = [&](int k) { if (ev) (void)hipEventRecord(ev[k], stream); };
    if (num_instances) *num_instances = 0;
    if (int rc = check_scene(scene)) return rc;
    if (!cam || !out || cam->image_width <= 0 || cam->image_height <= 0 || !(cam->tanfovx > 0.f) ||
        !(cam->tanfovy > 0.f) || !cam->viewmatrix || !cam->projmatrix || !cam->campos || !cam->bg || !out->color ||
        !out->depth || (scene->n > 0 && !out->radii) || max_instances < 0 || max_instances > 0x7fffffffLL)
        return PGR_ERR_INVALID_ARGUMENT;
    const int W = cam->image_width, H = cam->image_height, N = scene->n;
    const size_t P = (size_t)W * H;

    // N == 0: outputs stay zero-filled, no background (SURVEY.md section 8a "Edge cases")
    if (N == 0) {
        if (!hip_ok(hipMemsetAsync(out->color, 0, 3 * P * sizeof(float), stream), "memset color") ||
            !hip_ok(hipMemsetAsync(out->depth, 0, P * sizeof(float), stream), "memset depth"))
            return PGR_ERR_LAUNCH_FAILURE;
        if (out->final_T && !hip_ok(hipMemsetAsync(out->final_T, 0, P * sizeof(float), stream), "memset T"))
            return PGR_ERR_LAUNCH_FAILURE;
        if (out->n_contrib && !hip_ok(hipMemsetAsync(out->n_contrib, 0, P * sizeof(uint32_t), stream), "memset n"))
            return PGR_ERR_LAUNCH_FAILURE;
        return PGR_OK;
    }

    if (!workspace) return PGR_ERR_INVALID_ARGUMENT;
    const Layout L = make_layout(N, W, H, max_instances);
    if (workspace_bytes < L.total) return PGR_ERR_WORKSPACE_TOO_SMALL;
    char* ws = static_cast<char*>(workspace);
    auto* camd = reinterpret_cast<CameraDev*>(ws + L.cam);
    auto* counters = reinterpret_cast<uint32_t*>(ws + L.counters);
    auto* xy = reinterpret_cast<float2*>(ws + L.xy);
    auto* depth = reinterpret_cast<float*>(ws + L.depth);
    auto* conop = reinterpret_cast<float4*>(ws + L.conic_opacity);
    auto* rgbd = reinterpret_cast<float4*>(ws + L.rgb);
    auto* tiles_touched = reinterpret_cast<uint32_t*>(ws + L.tiles_touched);
    auto* offsets = reinterpret_cast<uint32_t*>(ws + L.offsets);
    auto* block_sums = reinterpret_cast<uint32_t*>(ws + L.block_sums);
    auto* keys_u = reinterpret_cast<uint64_t*>(ws + L.keys_unsorted);
    auto* vals_u = reinterpret_cast<uint32_t*>(ws + L.vals_unsorted);
    auto* keys_s = reinterpret_cast<uint64_t*>(ws + L.keys_sorted);
    auto* vals_s = reinterpret_cast<uint32_t*>(ws + L.vals_sorted);
    auto* ranges = reinterpret_cast<uint2*>(ws + L.ranges);

    pack_camera_kernel<<<1, 64, 0, stream>>>(cam->viewmatrix, cam->projmatrix, cam->campos, cam->bg, cam->tanfovx,
                                             cam->tanfovy, W, H, camd);
    mark(0);

    PreOut po{xy, depth, conop, rgbd, tiles_touched, out->radii, block_sums};
    preprocess_kernel<<<L.n_blocks, PRE_BLOCK, 0, stream>>>(*scene, camd, po);
    mark(1);
    scan_block_sums_kernel<<<1, SCAN_THREADS, 0, stream>>>(block_sums, L.n_blocks, counters, (uint32_t)max_instances);

    // Host reads num_rendered here, as the reference does (sizes the sort, reports overflow).
    uint32_t h_counters[2] = {0, 0};
    if (!hip_ok(hipMemcpyAsync(h_counters, counters, sizeof(h_counters), hipMemcpyDeviceToHost, stream), "memcpy") ||
        !hip_ok(hipStreamSynchronize(stream), "sync after scan"))
        return PGR_ERR_LAUNCH_FAILURE;
    const uint32_t total = h_counters[0];
    if (num_instances) *num_instances = (int64_t)total;
    if (h_counters[1] || (int64_t)total > max_instances) return PGR_ERR_INSTANCE_OVERFLOW;

    if (!hip_ok(hipMemsetAsync(ranges, 0, (size_t)L.tiles * sizeof(uint2), stream), "memset ranges"))
        return PGR_ERR_LAUNCH_FAILURE;

    mark(2);
    if (total > 0) {
        emit_kernel<<<L.n_blocks, PRE_BLOCK, 0, stream>>>(N, camd, xy, depth, out->radii, tiles_touched, block_sums,
                                                          counters, offsets, keys_u, vals_u);
        mark(3);
        int tbits = 0;
        while ((1 << tbits) < L.tiles) ++tbits;
        size_t temp_bytes = 0;
        if (!hip_ok(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_u, keys_s, vals_u, vals_s, (size_t)total, 0u,
                                              (unsigned)(32 + tbits), stream),
                    "radix_sort size query"))
            return PGR_ERR_LAUNCH_FAILURE;
        if (temp_bytes > L.sort_temp_bytes) return PGR_ERR_WORKSPACE_TOO_SMALL;
        if (!hip_ok(rocprim::radix_sort_pairs(ws + L.sort_temp, temp_bytes, keys_u, keys_s, vals_u, vals_s,
                                              (size_t)total, 0u, (unsigned)(32 + tbits), stream),
                    "radix_sort_pairs"))
            return PGR_ERR_LAUNCH_FAILURE;
        mark(4);
        tile_ranges_kernel<<<(total + 255) / 256, 256, 0, stream>>>(counters, keys_s, ranges);
        mark(5);
    } else {
        emit_kernel<<<L.n_blocks, PRE_BLOCK, 0, stream>>>(N, camd, xy, depth, out->radii, tiles_touched, block_sums,
                                                          counters, offsets, keys_u, vals_u);
        mark(3); mark(4); mark(5);
    }

    CompOut co{out->color, out->depth, out->final_T, out->n_contrib};
    composite_kernel<<<L.tiles, COMP_THREADS, 0, stream>>>(camd, ranges, vals_s, xy, conop, rgbd, co);
    mark(6);
    if (!hip_ok(hipGetLastError(), "kernel launch")) return PGR_ERR_LAUNCH_FAILURE;
    return PGR_OK;
}

extern "C" {

int32_t pgr_forward(const PgrScene* scene, const PgrCamera* cam, const PgrOutputs* out, void* workspace,
                    size_t workspace_bytes, int64_t max_instances, int64_t* num_instances, void* stream_v) {
    return forward_impl(scene, cam, out, workspace, workspace_bytes, max_instances, num_instances,
                        static_cast<hipStream_t>(stream_v), nullptr);
}

int32_t pgr_forward_profiled(const PgrScene* scene, const PgrCamera* cam, const PgrOutputs* out, void* workspace,
                             size_t workspace_bytes, int64_t max_instances, int64_t* num_instances, void* stream_v,
                             float* stage_ms) {
    if (!stage_ms) return PGR_ERR_INVALID_ARGUMENT;
    hipStream_t stream = static_cast<hipStream_t>(stream_v);
    hipEvent_t ev[PGR_NUM_STAGES + 1];
    for (auto& e : ev)
        if (!hip_ok(hipEventCreate(&e), "hipEventCreate")) return PGR_ERR_LAUNCH_FAILURE;
    for (int k = 0; k < PGR_NUM_STAGES; ++k) stage_ms[k] = 0.f;
    int32_t rc = forward_impl(scene, cam, out, workspace, workspace_bytes, max_instances, num_instances, stream, ev);
    if (rc == PGR_OK && scene->n > 0) {
        if (!hip_ok(hipStreamSynchronize(stream), "sync")) rc = PGR_ERR_LAUNCH_FAILURE;
        for (int k = 0; rc == PGR_OK && k < PGR_NUM_STAGES; ++k)
            if (!hip_ok(hipEventElapsedTime(&stage_ms[k], ev[k], ev[k + 1]), "hipEventElapsedTime"))
                rc = PGR_ERR_LAUNCH_FAILURE;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return rc;
}

int32_t pgr_mark_visible(int32_t n, const float* means3d, const float* viewmatrix, uint8_t* present, void* stream_v) {
    if (n < 0 || (n > 0 && (!means3d || !viewmatrix || !present))) return PGR_ERR_INVALID_ARGUMENT;
    if (n == 0) return PGR_OK;
    mark_visible_kernel<<<(n + 255) / 256, 256, 0, static_cast<hipStream_t>(stream_v)>>>(n, means3d, viewmatrix,
                                                                                         present);
    return hip_ok(hipGetLastError(), "mark_visible launch") ? PGR_OK : PGR_ERR_LAUNCH_FAILURE;
}

int32_t pgr_color_masks(const float* img_chw, int32_t width, int32_t height, const float* colors_k3, int32_t k,
                        float threshold, uint8_t* masks_khw, void* stream_v) {
    if (!img_chw || width <= 0 || height <= 0 || k < 0 || (k > 0 && (!colors_k3 || !masks_khw)))
        return PGR_ERR_INVALID_ARGUMENT;
    if (k == 0) return PGR_OK;
    const size_t P = (size_t)width * height;
    color_masks_kernel<<<(unsigned)((P + 255) / 256), 256, 0, static_cast<hipStream_t>(stream_v)>>>(
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

}  // extern "C"
