/*
 * pegasus_raster.h -- C ABI of libpegasus_raster.so, the MI355X (gfx950) Gaussian-splatting
 * rasterizer that replaces PEGASUS's CUDA extension `diff_gaussian_rasterization._C`.
 *
 * What each entry point replaces.  The reference binds its rasterizer through a PyTorch C++
 * extension that lives in an absent, un-pinned submodule
 * (/root/reference/.gitmodules:1-3 ; installed by /root/reference/setup.sh:19), so the
 * citations are to the reference's CALL SITES of that extension's Python surface:
 *
 *   pgr_forward            <- _C.rasterize_gaussians(...) behind GaussianRasterizer.forward;
 *                             reached from render(): /root/reference/src/gs/render.py:16,57,86,118,
 *                             /root/reference/pegasus.py:271
 *   pgr_mark_visible       <- _C.mark_visible behind GaussianRasterizer.markVisible
 *   pgr_color_masks        <- the colour-distance masks /root/reference/src/gs/render.py:60-63,89-93
 *   pgr_quantize_frame     <- (img*255).astype(uint8), (depth*1000).astype(uint16):
 *                             /root/reference/pegasus.py:347,355
 *   pgr_pack_records       <- the same casts for a whole batch + the K masks as bit planes, ONE record per frame: what the
 *                             writer threads of /root/reference/pegasus.py:346-358 consume / what the gather to the root
 *                             rank carries
 *   pgr_forward_layers_async <- render_silhouette_mask, /root/reference/src/gs/render.py:36-65 (every object rendered
 *                             ALONE and thresholded), for all objects and a batch of cameras in one pipeline pass
 *
 * Conventions
 *   - Every pointer in PgrScene / PgrCamera / PgrOutputs is a DEVICE address of a contiguous
 *     array (fp32 / int32) in the layouts PEGASUS's GaussianModel getters produce
 *     (/root/reference/src/gs/gaussian_model.py:105-128).  The caller (torch) owns all memory.
 *   - The library never allocates or frees device memory and never touches the default stream:
 *     all work is enqueued on `stream` (a hipStream_t passed as void*).
 *   - Re-entrant, no global state.  One call = one stream = one workspace.
 *   - Environment switches for A/B measurements and tests, read per call; results NEVER depend on them:
 *     PGR_BLOCK_CULL=0 (no per-block view culling), PGR_BIN_RECORDS=0 (the scatter walk of the binning evaluates the
 *     tight-list predicate itself instead of reading the count walk's verdicts).
 *   - Return value: 0 = enqueued; negative = PgrStatus.  pgr_status_string() names a code.
 */
#ifndef PEGASUS_RASTER_H
#define PEGASUS_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGR_ABI_VERSION 3
#define PGR_TILE_SIZE 16

typedef enum PgrStatus {
    PGR_OK = 0,
    PGR_ERR_INVALID_ARGUMENT = -1,   /* NULL / inconsistent pointers, bad sizes, bad sh_degree */
    PGR_ERR_WORKSPACE_TOO_SMALL = -2,/* workspace_bytes < pgr_workspace_bytes(...) */
    PGR_ERR_INSTANCE_OVERFLOW = -3,  /* listed instances > max_instances; *num_instances holds the need */
    PGR_ERR_LAUNCH_FAILURE = -4,     /* a HIP call failed; see pgr_last_hip_error */
    PGR_ERR_NO_DEVICE = -5
} PgrStatus;

/* One merged point cloud (environment first, then objects), activated values. */
typedef struct PgrScene {
    int32_t n;                   /* Gaussians */
    const float *means3d;        /* [n,3]  get_xyz */
    const float *opacities;      /* [n]    get_opacity (sigmoid applied) */
    const float *scales;         /* [n,3]  get_scaling (exp applied)      -- or NULL with cov3d_precomp */
    const float *rotations;      /* [n,4]  get_rotation (w,x,y,z), unit   -- or NULL with cov3d_precomp */
    const float *cov3d_precomp;  /* [n,6]  (xx,xy,xz,yy,yz,zz)            -- or NULL */
    const float *shs;            /* [n,sh_stride,3] get_features          -- or NULL with colors_precomp */
    const float *colors_precomp; /* [n,3]                                 -- or NULL */
    int32_t sh_degree;           /* active degree 0..3 */
    int32_t sh_stride;           /* coefficients stored per Gaussian, >= (sh_degree+1)^2 */
    float scale_modifier;
    const int32_t *tie_index;    /* [n] a permutation of 0..n-1, or NULL = the position itself.  The per-tile order is
                                    (depth, index): where two depths are EXACTLY equal the smaller tie_index comes
                                    first.  A scene stored in another order than the caller's (FrameRenderer's Morton
                                    layout) passes the caller's indices here and renders the caller's image bit for bit. */
    const uint32_t *tie_inv;     /* [n] inverse permutation of tie_index (tie_inv[tie_index[i]] = i), or NULL = the library
                                    rebuilds it in its workspace on every call.  A per-SCENE constant: pgr_scene_prepare
                                    computes it once into caller-owned memory. */
    const float *shs_rest;       /* optional: [n,sh_stride-1,3] -- the SH coefficients as PEGASUS's model STORES them, in two
                                    tensors: `shs` is then _features_dc [n,1,3] (coefficient 0 only) and this is
                                    _features_rest (coefficients 1 .. sh_stride-1).  Saves the caller the torch.cat of
                                    get_features (/root/reference/src/gs/gaussian_model.py:118-121: 768 MB moved per
                                    render() of a freshly merged 2 M-Gaussian scene).  Same coefficients, same arithmetic:
                                    results are bit-identical.  Forward entry points without PgrPosedObjects only
                                    (with poses, and in pgr_backward -- whose SH gradient is one [n,sh_stride,3] array --
                                    PGR_ERR_INVALID_ARGUMENT). */
} PgrScene;

/* The depth image's blend rule.  The reference's rasterizer is the absent fork `depth-diff-gaussian-rasterization`
 * (/root/reference/setup.sh:19); the widely used fork of that name writes the un-normalised expected depth, which the
 * in-tree evidence is consistent with (/root/reference/pegasus.py:355 scales it to millimetres as is) -- but the rule
 * itself is unverified (SURVEY.md section 8a "Depth variant"), so it is a switch:
 *   PGR_DEPTH_EXPECTED    depth = sum_i T_i alpha_i z_i                      (default; no background term)
 *   PGR_DEPTH_NORMALIZED  depth = sum_i T_i alpha_i z_i / (1 - T_final)      (sum_i T_i alpha_i = 1 - T_final; 0 where
 *                                                                             nothing was blended) */
typedef enum PgrDepthMode { PGR_DEPTH_EXPECTED = 0, PGR_DEPTH_NORMALIZED = 1 } PgrDepthMode;

/* The fields of GaussianRasterizationSettings that describe one view.  Scalars are host values;
 * the four small tensors stay on the device exactly as PEGASUS's Camera holds them. */
typedef struct PgrCamera {
    int32_t image_width, image_height;
    float tanfovx, tanfovy;
    const float *viewmatrix;     /* device [16], world_view_transform (transposed storage) */
    const float *projmatrix;     /* device [16], full_proj_transform  (transposed storage) */
    const float *campos;         /* device [3] */
    const float *bg;             /* device [3] */
    int32_t depth_mode;          /* PgrDepthMode; applies to depth and sem_depth of this view */
} PgrCamera;

typedef struct PgrOutputs {
    float *color;                /* [3,H,W]  required, except: a layered call writes no image; a RECORDS-ONLY view passes
                                    color = depth = NULL with `record` set and receives the frame record alone (round 6:
                                    a rank of a view-sharded job ships 3.84 MB per 800x800 frame and has no reader for the
                                    25.6 MB of fp32 / mask planes beside it) */
    float *depth;                /* [1,H,W]  required (same exceptions; NULL exactly when color is): sum_i T_i alpha_i z_i, no
                                    bg term; PgrCamera::depth_mode selects the normalised form */
    int32_t *radii;              /* [n]      required */
    float *final_T;              /* [H,W]    optional (NULL) */
    uint32_t *n_contrib;         /* [H,W]    optional (NULL) */
    float *sem_color;            /* [3,H,W]  the objects-only semantic render: REQUIRED on every view of a call that
                                    passes a PgrSemantic (PGR_ERR_INVALID_ARGUMENT otherwise), ignored without one.  A
                                    records-only view (color == NULL, record set) may pass NULL when the descriptor carries
                                    mask_colors: the semantic image then exists only as the record's mask planes */
    float *sem_depth;            /* [1,H,W]  optional, only with sem_color */
    uint8_t *sem_masks;          /* [K,H,W]  optional: the K colour-distance masks of the semantic image
                                    (pgr_color_masks of sem_color against PgrSemantic::mask_colors, bit for bit), written by
                                    the compositor's epilogue from the pixel it holds in registers -- needs
                                    PgrSemantic::mask_colors.  In a LAYERED call (pgr_forward_layers_async) the one output:
                                    [n_layers,H,W], plane k = layer k's image against mask_colors[k]. */
    uint8_t *record;             /* optional: this view's FRAME RECORD (PgrRecordLayout below: uint8 rgb | uint16 depth mm |
                                    mask bit planes, pgr_frame_record_layout(width, height, K) bytes), written by the
                                    compositor's epilogue from the pixel it holds in registers -- bit for bit what
                                    pgr_pack_records makes of color / depth / sem_masks, without the pass that re-reads them.
                                    K = the semantic descriptor's k_objects when it carries mask_colors, else 0 (no mask
                                    section).  What leaves the GPU for a finished frame (gather to the root rank, writers). */
} PgrOutputs;

/* Fused semantic pass: PEGASUS renders the objects alone, painted in flat semantic colours, to derive masks
 * (/root/reference/src/gs/render.py:68-97, colours from pegasus.py:230-232).  With this descriptor the batch
 * call also writes that image (outs[v].sem_color) from the SAME per-tile lists -- the objects-only list of a
 * tile is the scene's list minus the environment entries -- instead of a second preprocess/bin/sort/composite. */
typedef struct PgrSemantic {
    const int32_t *object_id;    /* device [n]: 0 = environment, k = object k (1..k_objects) */
    const float *colors;         /* device [k_objects,3]: the rgb each object's Gaussians carry =
                                    max(C0 * RGB2SH(c_k) + 0.5, 0), evaluated in fp32 in that order */
    int32_t n_env;               /* Gaussians [0, n_env) are the environment */
    int32_t k_objects;
    const uint8_t *object_id_u8; /* device [n - n_env]: object_id[n_env + j] as a byte, or NULL = the library packs it in
                                    its workspace on every call (k_objects <= 255).  Per-scene constant: pgr_scene_prepare */
    const float *mask_colors;    /* device [k_objects,3]: the colours the masks are thresholded against (the c_k of
                                    /root/reference/src/gs/render.py:60-63,89-93), or NULL = no sem_masks output */
    float mask_threshold;        /* L2 distance, the reference's 0.1 */
} PgrSemantic;                   /* checked before anything is enqueued: object_id (unless the scene is empty), colors non-NULL,
                                    n_env >= 0, k_objects > 0, outs[v].sem_color non-NULL for every view, mask_colors
                                    non-NULL if any view passes sem_masks */

/* Device pointers into one view's slice of a workspace, for stage-level parity tests and for backward. */
typedef struct PgrWorkspaceView {
    const float *splats;         /* [n,12] per-Gaussian record: x, y, conic A, B, C, opacity, B/C, B/A, r, g, b, depth
                                    (defined only for Gaussians with a non-empty rectangle) */
    const uint16_t *rects;       /* [n,4] tile rectangle minx,miny,maxx,maxy (max exclusive); zeros = culled.
                                    Written only for views rendered WITH a radii output. */
    const uint32_t *gauss_sorted;/* [num_instances] Gaussian index, tile-major, (depth, index) ascending per tile */
    const uint32_t *ranges;      /* [tiles,2] start,end into gauss_sorted */
    const uint32_t *num_instances; /* [0] listed instances, [1] overflow flag */
} PgrWorkspaceView;

int32_t pgr_abi_version(void);
const char *pgr_version(void);
const char *pgr_status_string(int32_t status);
/* Text of the last HIP error seen by THIS thread inside the library (thread-local, read-only use). */
const char *pgr_last_hip_error(void);

/* Bytes of workspace needed to render one view of `n` Gaussians at width x height with room for
 * `max_instances` (Gaussian,tile) pairs. */
size_t pgr_workspace_bytes(int32_t n, int32_t width, int32_t height, int64_t max_instances);
/* Same for a batch of `n_views` views in flight at once (pgr_forward_batch). */
size_t pgr_batch_workspace_bytes(int32_t n, int32_t width, int32_t height, int64_t max_instances_per_view,
                                 int32_t n_views);

/* Render one view.  `num_instances` (host, optional) receives the number of listed (Gaussian, tile)
 * instances.  Everything is enqueued on `stream`; the call synchronises the stream once, at the end, to
 * read the instance count and the overflow flag (the reference synchronises mid-pipeline to read
 * num_rendered), so that an overflow is reported instead of returned as an image. */
int32_t pgr_forward(const PgrScene *scene, const PgrCamera *camera, const PgrOutputs *out,
                    void *workspace, size_t workspace_bytes, int64_t max_instances,
                    int64_t *num_instances, void *stream);

/* Render `n_views` views of ONE scene (the per-frame loop of /root/reference/pegasus.py:254-325 calls
 * render() once per camera over the same merged cloud; this is that loop as one call).  `cameras` and
 * `outs` are HOST arrays of n_views entries; all views share the image size; outs[v].radii may be NULL.
 * The batch is what fills an MI355X: the compositing of every (view, tile) list is one launch, ordered
 * longest list first, so no view waits on its own slowest tile.  One stream synchronisation per batch, at
 * its end (instance counts + overflow flags); `num_instances` (host, optional) receives n_views counts. */
int32_t pgr_forward_batch(const PgrScene *scene, int32_t n_views, const PgrCamera *cameras,
                          const PgrOutputs *outs, void *workspace, size_t workspace_bytes,
                          int64_t max_instances_per_view, int64_t *num_instances, void *stream);

/* Asynchronous form: enqueues the whole batch on `stream` and returns without synchronising.  `host_scratch`
 * is PINNED host memory of pgr_host_scratch_bytes(n_views) that the library uses to stage its pointer tables and
 * to receive the status words; it must stay untouched until `stream` has passed the call.  After synchronising,
 * pgr_batch_status(host_scratch, n_views, num_instances) returns PGR_OK or PGR_ERR_INSTANCE_OVERFLOW (frames of
 * an overflowed batch are not valid; re-run with a larger capacity).  Lets two batches -- e.g. the scene pass
 * and the semantic pass of a frame set -- run concurrently on two streams with two workspaces. */
size_t pgr_host_scratch_bytes(int32_t n_views);
int32_t pgr_forward_batch_async(const PgrScene *scene, int32_t n_views, const PgrCamera *cameras,
                                const PgrOutputs *outs, void *workspace, size_t workspace_bytes,
                                int64_t max_instances_per_view, void *host_scratch, size_t host_scratch_size,
                                void *stream);
/* Same, with the fused semantic pass (`semantic` may be NULL = plain batch). */
int32_t pgr_forward_frames_async(const PgrScene *scene, const PgrSemantic *semantic, int32_t n_views,
                                 const PgrCamera *cameras, const PgrOutputs *outs, void *workspace,
                                 size_t workspace_bytes, int64_t max_instances_per_view, void *host_scratch,
                                 size_t host_scratch_size, void *stream);

/* Dynamic scenes: per-view rigid poses of the scene's objects, applied INSIDE the preprocess instead of composing a
 * posed copy of the scene per time step (reference: update_object_pose + deepcopy + merge per frame,
 * /root/reference/pegasus.py:254-264,387-390; /root/reference/src/gs/pegasus_setup.py:160-226).  A Gaussian with
 * object_id k > 0 is placed by poses[view][k-1] with the arithmetic of pgr_compose_object (position, orientation);
 * its view-dependent colour is its own SH evaluated in the object's frame (direction R^T d) -- the function the
 * band-rotated coefficients of a composed copy represent.  A batch of time steps then runs like a batch of cameras.
 * Needs scales + rotations (no cov3d_precomp). */
#define PGR_POSE_STRIDE 20
typedef struct PgrPosedObjects {
    const int32_t *object_id;   /* [n] device; 0 = not posed (environment) */
    const float *poses;         /* [n_views, k_objects, PGR_POSE_STRIDE] device:
                                   R[9] row-major, t[3], center[3], q[4] = R as unit quaternion (w,x,y,z), pad */
    int32_t k_objects;
} PgrPosedObjects;
/* pgr_forward_frames_async with per-view object poses (`posed` may be NULL = same as pgr_forward_frames_async). */
int32_t pgr_forward_posed_async(const PgrScene *scene, const PgrSemantic *semantic, const PgrPosedObjects *posed,
                                int32_t n_views, const PgrCamera *cameras, const PgrOutputs *outs, void *workspace,
                                size_t workspace_bytes, int64_t max_instances_per_view, void *host_scratch,
                                size_t host_scratch_size, void *stream);
int32_t pgr_batch_status(const void *host_scratch, int32_t n_views, int64_t *num_instances);
/* pgr_forward_posed_async whose status words reach the host EARLY.  The instance counts and overflow flags are final once the
 * tile scan has run -- a third into a single-view call -- and nothing later changes them: the scan stores them into
 * `host_scratch` itself (which must therefore be device-accessible at its host address: hipHostMalloc memory, what torch's
 * pin_memory() hands out) and `status_event` (a hipEvent_t) is recorded on `stream` right behind it.  After
 * hipEventSynchronize(status_event), pgr_batch_status(host_scratch, ...) is valid while scatter, sort and compositor still
 * run: the caller of a single view (PEGASUS's render(), /root/reference/src/gs/render.py:17-24 -- the upstream rasterizer
 * blocks on its own instance count in the middle of every call the same way) returns to its host code without leaving the
 * GPU idle; the outputs are complete in STREAM order, as for every asynchronous call.  On PGR_ERR_INSTANCE_OVERFLOW the rest
 * of the call does nothing; re-run with a larger capacity. */
int32_t pgr_forward_posed_early_status(const PgrScene *scene, const PgrSemantic *semantic, const PgrPosedObjects *posed,
                                       int32_t n_views, const PgrCamera *cameras, const PgrOutputs *outs, void *workspace,
                                       size_t workspace_bytes, int64_t max_instances_per_view, void *host_scratch,
                                       size_t host_scratch_size, void *stream, void *status_event);

/* Per-SCENE constants the batch calls would otherwise rebuild on every call (round 3: invert_tie_index_kernel and
 * pack_object_ids_kernel, 24 us + 76 MB of traffic per batch of the 2 M-Gaussian scene): the inverse of
 * PgrScene::tie_index and the object ids of the object Gaussians as bytes.  Enqueues the work on `stream`, writes into
 * caller-owned `cache` (device, pgr_scene_cache_bytes(n) bytes) and returns the two device pointers to put into
 * PgrScene::tie_inv / PgrSemantic::object_id_u8 (NULL where the input is absent: scene->tie_index == NULL, semantic ==
 * NULL, or k_objects > 255).  Valid until the scene's tie_index / object ids change. */
size_t pgr_scene_cache_bytes(int32_t n);
int32_t pgr_scene_prepare(const PgrScene *scene, const PgrSemantic *semantic, void *cache, size_t cache_bytes,
                          const uint32_t **tie_inv, const uint8_t **object_id_u8, void *stream);

/* Layered render = silhouette masks (/root/reference/src/gs/render.py:36-65: every object rendered ALONE over an empty
 * environment and thresholded against its semantic colour -- the reference does one deepcopy + merge + render per object
 * and camera).  One pipeline pass renders n_layers images per view: a Gaussian with layer_id k > 0 is composited into
 * image k only (per-(tile, layer) lists: the binning treats the n_layers images as one image of n_layers x grid_y tile
 * rows; projection, lists and blending of a layer are those of rendering its Gaussians alone), and the compositor's
 * epilogue writes outs[v].sem_masks[k-1] = || pixel - mask_colors[k-1] ||_2 <= mask_threshold.  No colour image is
 * written (outs[v].color / depth may be NULL).  layer_id must be non-decreasing along the scene (PEGASUS merges object
 * after object); Gaussians with layer_id 0 are dropped.  `posed` as in pgr_forward_posed_async (may be NULL).
 * EMPTY LAYERS: the plane of a layer no Gaussian carries is all 0, whatever the background -- the reference never renders
 * an object that is not in gs_object_list and leaves its mask column 0 (/root/reference/src/gs/render.py:44-63); an
 * empty scene (n == 0) is the same rule for every layer.  A layer WITH Gaussians of which none reaches a pixel holds the
 * background's verdict || bg - mask_colors[k-1] ||_2 <= mask_threshold there, as a render of that object alone would. */
typedef struct PgrLayers {
    const int32_t *layer_id;     /* device [n] */
    int32_t n_layers;
    const float *mask_colors;    /* device [n_layers,3] */
    float mask_threshold;
} PgrLayers;
size_t pgr_layers_workspace_bytes(int32_t n, int32_t width, int32_t height, int64_t max_instances_per_view,
                                  int32_t n_views, int32_t n_layers);
int32_t pgr_forward_layers_async(const PgrScene *scene, const PgrLayers *layers, const PgrPosedObjects *posed,
                                 int32_t n_views, const PgrCamera *cameras, const PgrOutputs *outs, void *workspace,
                                 size_t workspace_bytes, int64_t max_instances_per_view, void *host_scratch,
                                 size_t host_scratch_size, void *stream);

/* Profiling twin of pgr_forward_batch (bench / rocprof only): records HIP events on `stream` at the
 * stage boundaries (each stage runs for all views before the next starts), synchronises, and writes the
 * elapsed milliseconds of each stage for the whole batch to stage_ms[PGR_NUM_STAGES] in PgrStage order. */
#define PGR_NUM_STAGES 5
typedef enum PgrStage {
    PGR_STAGE_PREPROCESS = 0,  /* camera pack + per-Gaussian projection / EWA / SH */
    PGR_STAGE_BIN_COUNT = 1,   /* per-chunk tile histograms, slice reservation, tile scan */
    PGR_STAGE_BIN_SCATTER = 2, /* (depth, index) pairs into the tiles' slices */
    PGR_STAGE_TILE_SORT = 3,   /* work order + per-tile (depth, index) sort */
    PGR_STAGE_COMPOSITE = 4    /* front-to-back alpha compositing of all views; with a PgrSemantic the same
                                  walk also accumulates the objects-only semantic image */
} PgrStage;
int32_t pgr_forward_batch_profiled(const PgrScene *scene, const PgrSemantic *semantic, int32_t n_views,
                                   const PgrCamera *cameras, const PgrOutputs *outs, void *workspace,
                                   size_t workspace_bytes, int64_t max_instances_per_view,
                                   int64_t *num_instances, void *stream, float *stage_ms);

/* Conservative block visibility (pegasus_amd/csrc/blockcull.hip.h): bit (v % 32) of
 * vis_words[g * ceil(n_views / 32) + v / 32] is CLEAR only if none of the Gaussians [64 g, 64 g + 64) can get a
 * non-zero radius in view v (all behind the near plane, or all tile rectangles empty) -- the test the batch entry
 * points run internally to skip whole waves in the per-Gaussian stages (switch: environment PGR_BLOCK_CULL=0; outputs
 * never depend on it).  A block-granular, many-view relative of markVisible; exposed for the parity tests.
 * vis_words: device [ceil(n / 64) * ceil(n_views / 32)]. */
size_t pgr_block_visibility_workspace_bytes(int32_t n, int32_t n_views);
int32_t pgr_block_visibility(const PgrScene *scene, int32_t n_views, const PgrCamera *cameras, void *workspace,
                             size_t workspace_bytes, uint32_t *vis_words, void *stream);

/* Fill `view` with device pointers into view `view_index` of a workspace laid out for
 * (n,width,height,max_instances,n_views). */
int32_t pgr_workspace_view(void *workspace, size_t workspace_bytes, int32_t n, int32_t width, int32_t height,
                           int64_t max_instances, int32_t n_views, int32_t view_index, PgrWorkspaceView *view);

/* Rigid pose of one object, host struct (copied into the launch).  R row-major, q = R as a unit quaternion
 * (w,x,y,z), D1/D2/D3 = real-SH band rotation matrices (row-major, c' = D c) for the rasterizer's basis. */
typedef struct PgrObjectPose {
    float R[9];
    float t[3];
    float center[3];
    float q[4];
    float D1[9], D2[25], D3[49];
} PgrObjectPose;

/* Scene composition (replaces GaussianModel.apply_transformation + merge_gaussians,
 * /root/reference/src/gs/gaussian_model.py:482-546,584-591, called per frame at pegasus.py:255-264,387-390):
 *   out_xyz[i]  = R (xyz[i] - center) + center + t
 *   out_rot[i]  = q (x) normalise(rot[i])                    (skipped when rot or out_rot is NULL)
 *   out_rest[i] = band-wise D_l * f_rest[i]                  (n_rest in {0,3,8,15} coefficients of 3 floats)
 * Input rows of f_rest are in_rest_stride floats apart, output rows out_rest_stride floats apart, so the output can
 * point INTO a merged [N,16,3] feature tensor (base + 3 floats, stride 48).  In-place operation is allowed. */
int32_t pgr_compose_object(int32_t n, const float *xyz, const float *rot, const float *f_rest, int32_t n_rest,
                           int32_t in_rest_stride, const PgrObjectPose *pose, float *out_xyz, float *out_rot,
                           float *out_rest, int32_t out_rest_stride, void *stream);

/* The pose calls of a whole frame (round 6).  PEGASUS re-poses every object between two frames of a dynamic sequence with
 * three calls per object -- apply_transformation_on_xyz(T), apply_rotation_on_splats(R), apply_rotation_on_sh(R):
 * /root/reference/src/gs/pegasus_setup.py:195-208 -> /root/reference/src/gs/gaussian_model.py:482-546 -- each handed a
 * rotation / translation that already lives on the device.  One JOB per (object, array); R and t stay DEVICE pointers (no
 * host round trip, unlike PgrObjectPose), the cloud's mean, the quaternion of R and the SH band matrices are derived on the
 * device:
 *   PGR_POSE_XYZ   dst[i] = R (src[i] - c) + c + t,  c = the mean of src (about_origin = 0) or 0      src, dst [n,3]
 *   PGR_POSE_ROT   dst[i] = quat(R) (x) normalise(src[i])                              (w,x,y,z)       src, dst [n,4]
 *   PGR_POSE_SH    dst[i] = band-wise D_l(R) src[i],  D_l = pinv(B_l) B_l(R^T d_k)                     src, dst [n,n_rest,3]
 * R = NULL: identity; t = NULL: zero.  R and t may point INTO a 4x4 transform (strides below): PEGASUS builds T on the device and
 * hands its corner and last column over.  src == dst is allowed.  Jobs of one call must not depend on each other.
 * sh_dirs [61,3] / sh_pinv [15,61]: device fp64 tables of the SH sample directions and the three bands' pseudo-inverses
 * (pegasus_amd/sh_rotation.py builds them for the rasterizer's basis; needed only if a job is PGR_POSE_SH).
 * workspace: pgr_pose_objects_workspace_bytes(n_jobs) bytes of device memory. */
typedef enum PgrPoseKind { PGR_POSE_XYZ = 0, PGR_POSE_ROT = 1, PGR_POSE_SH = 2 } PgrPoseKind;
typedef struct PgrPoseJob {
    const float *src;
    float *dst;
    const float *R;              /* device, row-major: R[r][c] at R[r * R_row_stride + c] */
    const float *t;              /* device: t[c] at t[c * t_stride] */
    int32_t n;
    int32_t kind;                /* PgrPoseKind */
    int32_t n_rest;              /* PGR_POSE_SH: coefficients per Gaussian, 3, 8 or 15 */
    int32_t about_origin;        /* PGR_POSE_XYZ */
    int32_t R_row_stride;        /* floats between the rows of R: 3 = a contiguous 3x3, 4 = the corner of a 4x4 */
    int32_t t_stride;            /* floats between the components of t: 1 = contiguous, 4 = the last column of a 4x4 */
} PgrPoseJob;
size_t pgr_pose_objects_workspace_bytes(int32_t n_jobs);
int32_t pgr_pose_objects(int32_t n_jobs, const PgrPoseJob *jobs, const double *sh_dirs, const double *sh_pinv,
                         void *workspace, size_t workspace_bytes, void *stream);

/* Measurement aid (bench.py's roofline.issue_model; replaces nothing of the reference): ONE wave that reads its two clocks
 * around a sleep loop of spin_us microseconds (0: no loop, a time stamp) -- ticks[0] = shader cycles (s_memtime), ticks[1] =
 * 100 MHz ticks (s_memrealtime), ticks[2] / ticks[3] = the 100 MHz counter at its start / end -- so ticks[0] / ticks[1] x
 * 100 MHz is the shader clock under whatever runs beside it on other streams, and stamps taken on the measured stream show
 * whether the two really overlapped.  `ticks`: device uint64[4]. */
int32_t pgr_clock_probe(uint64_t *ticks, uint32_t spin_us, void *stream);

/* Mean squared distance to the 3 nearest neighbours of every point (replaces simple_knn._C.distCUDA2 of the reference's
 * second absent submodule: GaussianModel.create_from_pcd, /root/reference/src/gs/gaussian_model.py:25,147).  Exact
 * 3-NN over a device-built uniform grid; with fewer than 4 points the missing neighbours count as FLT_MAX, as upstream.
 * `xyz` [n,3] and `out` [n] are device pointers; workspace of pgr_knn_workspace_bytes(n) bytes. */
size_t pgr_knn_workspace_bytes(int32_t n);
int32_t pgr_knn_mean_dist2(int32_t n, const float *xyz, float *out, void *workspace, size_t workspace_bytes,
                           void *stream);

/* Gradients returned by pgr_backward (device pointers, any may be NULL = not wanted). */
typedef struct PgrGradOutputs {
    float *means2d;              /* [n,3] screen-space mean, NDC-scaled (what viewspace_points.grad receives) */
    float *means3d;              /* [n,3] */
    float *opacities;            /* [n]   */
    float *colors;               /* [n,3] wrt colors_precomp (or the SH-evaluated rgb) */
    float *shs;                  /* [n,sh_stride,3] */
    float *cov3d;                /* [n,6] wrt the stored covariance parameters */
    float *scales;               /* [n,3] */
    float *rotations;            /* [n,4] */
} PgrGradOutputs;

/* Backward of ONE view rendered by pgr_forward into `workspace` (which must be untouched since, with the same
 * n / image size / max_instances), given dL/dcolor [3,H,W] and optionally dL/ddepth [1,H,W], and the forward's
 * final_T / n_contrib outputs.  Replaces _C.rasterize_gaussians_backward of the reference's extension (used by
 * training only: /root/reference/src/gs/gs_training.py:7,46).  `grad_rows` is scratch of n*12 floats. */
int32_t pgr_backward(const PgrScene *scene, const PgrCamera *camera, const float *grad_color,
                     const float *grad_depth, const float *final_T, const uint32_t *n_contrib,
                     const int32_t *radii, void *workspace, size_t workspace_bytes, int64_t max_instances,
                     const PgrGradOutputs *grads, float *grad_rows, void *stream);

/* present[i] = 1 iff Gaussian i passes the near-plane test of `viewmatrix` (device [16]). */
int32_t pgr_mark_visible(int32_t n, const float *means3d, const float *viewmatrix, uint8_t *present,
                         void *stream);

/* masks[b,k,y,x] = || img[b,:,y,x] - colors[k,:] ||_2 <= threshold   (uint8 0/1); img is a contiguous batch of
 * n_images CHW fp32 images, masks the matching [n_images,k,H,W] batch. */
int32_t pgr_color_masks(const float *img_chw, int32_t n_images, int32_t width, int32_t height,
                        const float *colors_k3, int32_t k, float threshold, uint8_t *masks_khw, void *stream);

/* rgb_hwc = uint8(img*255) (wraps, no clamp), depth_mm = uint16(depth*1000). Either pair may be NULL. */
int32_t pgr_quantize_frame(const float *img_chw, const float *depth_hw, int32_t width, int32_t height,
                           uint8_t *rgb_hwc, uint16_t *depth_mm_hw, void *stream);

/* Batch form of pgr_quantize_frame plus the K per-object masks of every frame as bit planes -- one launch turns a
 * finished batch into what leaves the GPU (the writer threads of /root/reference/pegasus.py:346-358, or the gather of
 * finished frames to the root rank):
 *   rgb_bhwc[b]      = uint8(color[b] * 255)   (wraps, no clamp)        color_b3hw  [n_images,3,H,W]
 *   depth_mm_bhw[b]  = uint16(depth[b] * 1000)                          depth_bhw   [n_images,H,W]
 *   mask_bits_bhwj[b,y,x,j] bit (m % 8) = masks[b,m,y,x] != 0, j = m / 8   masks_bkhw  [n_images,k,H,W] of
 *                      pgr_color_masks; ceil(k/8) bytes per pixel (one byte for the usual k <= 8 objects)
 * Each input/output pair may be NULL (both or neither). */
int32_t pgr_pack_frames(const float *color_b3hw, const float *depth_bhw, const uint8_t *masks_bkhw, int32_t n_images,
                        int32_t k, int32_t width, int32_t height, uint8_t *rgb_bhwc, uint16_t *depth_mm_bhw,
                        uint8_t *mask_bits_bhwj, void *stream);

/* One RECORD per frame -- everything that leaves the GPU for a finished frame, contiguous, so that a batch is one buffer
 * for the disk writers and ONE collective for the gather to the root rank (3.84 MB per 800x800
 * frame at K <= 8):
 *   [0, 3P)                       uint8 rgb HWC   = uint8(color * 255)      (wraps, no clamp; pegasus.py:347)
 *   [off_depth, off_depth + 2P)   uint16 depth mm = uint16(depth * 1000)    (pegasus.py:355), little endian
 *   [off_masks, off_masks + J P)  mask bit planes, J = ceil(k / 8) bytes per pixel: bit (m % 8) of byte m / 8 = mask m
 * with P = width x height and the three offsets / the record size rounded up to 16 bytes (pgr_frame_record_layout).
 * records: device uint8, record b at records + b * record_stride (record_stride >= layout.bytes, a multiple of 16). */
typedef struct PgrRecordLayout {
    int64_t off_rgb, off_depth, off_masks, bytes;
} PgrRecordLayout;
int32_t pgr_frame_record_layout(int32_t width, int32_t height, int32_t k, PgrRecordLayout *layout);
int32_t pgr_pack_records(const float *color_b3hw, const float *depth_bhw, const uint8_t *masks_bkhw, int32_t n_images,
                         int32_t k, int32_t width, int32_t height, uint8_t *records, int64_t record_stride, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PEGASUS_RASTER_H */
