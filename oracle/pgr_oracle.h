/*
 * pgr_oracle.h -- CPU ORACLE for the PEGASUS Gaussian-splatting rasterizer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / reported CPU baseline.  The product (pegasus_amd/) never links, imports
 * or falls back to it.
 *
 * PARITY UNPINNED.  The algorithm lives in a third-party dependency that is ABSENT from
 * /root/reference: the un-initialised, un-pinned git submodule
 *   submodules/gaussian-splatting-pegasus  (/root/reference/.gitmodules:1-3)
 *   -> submodules/depth-diff-gaussian-rasterization (installed by /root/reference/setup.sh:19).
 * The reference ships no tests, golden vectors or fixtures for this path, and
 * nothing of it can be compiled or imported here.  This file therefore restates the PUBLISHED
 * algorithm (3D Gaussian Splatting, Kerbl et al. 2023, cited at /root/reference/README.md:285-299;
 * depth variant = un-normalised expected depth sum(T*alpha*z)) and anchors it on the
 * reference's own call sites:
 *   /root/reference/src/gs/render.py:16-20,57-63,86-93,118-129   (render() dict, CHW layout, masks)
 *   /root/reference/src/gs/gaussian_model.py:105-128              (activations feeding the rasterizer)
 *   /root/reference/pegasus.py:347,355                            (uint8 / uint16-mm quantisation)
 * and on analytic known-answer tests (tests/test_oracle_kat.py).
 *
 * ARITHMETIC CONTRACT.  All arithmetic is IEEE-754 binary32, round-to-nearest-even, compiled
 * with -ffp-contract=off.  Every fused multiply-add is written explicitly as fmaf(); the HIP
 * kernels use the same operation order, so every stage up to and including the sorted
 * instance list is BIT-EXACT between oracle and GPU.  The only non-bit-exact operation is
 * exp() in the compositor (glibc expf here, v_exp_f32 on gfx950); pixels where a
 * threshold decision lies within rounding distance of flipping are reported in `ambig`.
 */
#ifndef PGR_ORACLE_H
#define PGR_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGR_TILE 16                 /* 16x16 pixel tiles */
#define PGR_NEAR_Z 0.2f             /* cull iff view-space z <= 0.2 */
#define PGR_LOWPASS 0.3f            /* screen-space dilation added to cov2D diagonal */
#define PGR_ALPHA_MAX 0.99f
#define PGR_ALPHA_MIN (1.0f / 255.0f)
#define PGR_T_EPS 0.0001f

typedef struct PgrOracleIn {
    int32_t n;                      /* number of Gaussians */
    const float *means3d;           /* [n,3] */
    const float *opacities;         /* [n]   activated (sigmoid applied by caller) */
    const float *scales;            /* [n,3] activated (exp applied), or NULL if cov3d_precomp */
    const float *rotations;         /* [n,4] (w,x,y,z) activated (normalised), or NULL */
    const float *cov3d_precomp;     /* [n,6] (xx,xy,xz,yy,yz,zz) or NULL */
    const float *shs;               /* [n,sh_stride,3] or NULL if colors_precomp */
    const float *colors_precomp;    /* [n,3] or NULL */
    int32_t sh_degree;              /* active degree 0..3 */
    int32_t sh_stride;              /* coefficients stored per Gaussian (16 for max degree 3) */
    float scale_modifier;
    int32_t width, height;
    float tanfovx, tanfovy;
    float viewmatrix[16];           /* element (row r, col c) of the usual matrix at [c*4+r] */
    float projmatrix[16];
    float campos[3];
    float bg[3];
    int32_t cull_mode;              /* 0 = reference lists (every tile of the 3-sigma rectangle);
                                       1 = tight lists: instances that provably cannot reach alpha >= 1/255 at
                                           any pixel of their tile are dropped (pgr_oracle_tile_may_contribute).
                                           Images are bit-identical in both modes; n_contrib indexes the list. */
    /* posed objects (dynamic scenes; include/pegasus_raster.h PgrPosedObjects), all NULL/0 = none:
     * a Gaussian with object_id k > 0 is placed by poses[k-1] exactly as the scene-composition step places it
     * (x' = R (x - center) + center + t, q' = q_R (x) normalise(q)); its colour is its own SH evaluated in the
     * object's frame (direction R^T d), the function the band-rotated coefficients of a composed copy represent. */
    const int32_t *object_id;       /* [n] or NULL */
    const float *poses;             /* [k_objects, PGR_POSE_STRIDE]: R[9] row-major, t[3], center[3], q[4] (w,x,y,z), pad */
    int32_t k_objects;
    const int32_t *tie_index;       /* [n] permutation or NULL: exact depth ties are broken by tie_index instead of the
                                       position (include/pegasus_raster.h PgrScene::tie_index) */
    int32_t depth_mode;             /* 0 = out_depth = sum T alpha z (the default: SURVEY.md section 8a "Depth variant");
                                       1 = that sum / (1 - T_final), 0 where nothing was blended
                                       (include/pegasus_raster.h PgrDepthMode) */
} PgrOracleIn;
#define PGR_POSE_STRIDE 20

typedef struct PgrOracleOut {
    /* per-Gaussian (all caller-allocated, any may be NULL) */
    int32_t *radii;                 /* [n] */
    int32_t *tiles_touched;         /* [n] */
    float *xy;                      /* [n,2] */
    float *depth;                   /* [n]   */
    float *conic_opacity;           /* [n,4] */
    float *rgb;                     /* [n,3] */
    float *cov3d;                   /* [n,6] */
    /* binning */
    int64_t num_instances;          /* OUT: sum tiles_touched */
    uint64_t *keys_sorted;          /* [cap_instances] or NULL */
    uint32_t *gauss_sorted;         /* [cap_instances] or NULL */
    int64_t cap_instances;
    uint32_t *ranges;               /* [tiles,2] start,end or NULL */
    /* image */
    float *out_color;               /* [3,H,W] */
    float *out_depth;               /* [1,H,W] */
    float *final_T;                 /* [H,W] or NULL */
    uint32_t *n_contrib;            /* [H,W] or NULL */
    uint8_t *ambig;                 /* [H,W] or NULL: 1 = a threshold decision is within rounding of flipping */
} PgrOracleOut;

/* returns 0 on success; -1 invalid argument; -2 instance capacity too small (num_instances is still set) */
int pgr_oracle_forward(const PgrOracleIn *in, PgrOracleOut *out, int num_threads);

/* stage entry points used by stage-level parity tests */
int pgr_oracle_preprocess(const PgrOracleIn *in, PgrOracleOut *out, int num_threads);

/* mark_visible: 1 iff view-space z > 0.2 */
int pgr_oracle_mark_visible(int32_t n, const float *means3d, const float *viewmatrix, uint8_t *present);

/* colour-distance masks (reference: src/gs/render.py:60-63,89-93): mask[k,y,x] = ||img[:,y,x]-colors[k]||_2 <= thr */
int pgr_oracle_color_masks(const float *img_chw, int32_t width, int32_t height, const float *colors_k3,
                           int32_t k, float thr, uint8_t *masks_khw);

/* output quantisation (reference: pegasus.py:347,355): rgb (img*255).astype(uint8) wraps, depth (d*1000).astype(uint16) */
int pgr_oracle_quantize(const float *img_chw, const float *depth_hw, int32_t width, int32_t height,
                        uint8_t *rgb_hwc, uint16_t *depth_mm_hw);

/* The tight-list predicate (cull_mode 1): 1 if the splat (xy, conic+opacity) may reach alpha >= 1/255 at some
 * pixel centre of tile (tx,ty), 0 if it provably cannot.  Pure fp32 +,-,*,/ and integer ops: bit-reproducible. */
int pgr_oracle_tile_may_contribute(const float xy[2], const float conic_opacity[4], int32_t tx, int32_t ty,
                                   int32_t width, int32_t height);

const char *pgr_oracle_version(void);

#ifdef __cplusplus
}
#endif
#endif
