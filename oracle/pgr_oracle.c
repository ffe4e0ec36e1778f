/*
 * pgr_oracle.c -- CPU ORACLE (test infrastructure, NOT product code; see pgr_oracle.h header).
 * PARITY UNPINNED: the reference's rasterizer source is an absent, un-pinned submodule
 * (/root/reference/.gitmodules:1-3, /root/reference/setup.sh:19); this restates the published
 * 3DGS forward algorithm, anchored on the reference's call sites cited per function below.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -mfma -fopenmp -shared -fPIC
 */
#if !defined(_OPENMP) && !defined(_POSIX_C_SOURCE)
#define _POSIX_C_SOURCE 199309L      /* clock_gettime for the timing stamps of a build without -fopenmp */
#endif
#include "pgr_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#define pgr_wtime() omp_get_wtime()
#else
#include <time.h>
static double pgr_wtime(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
#endif

const char *pgr_oracle_version(void) { return "pgr-oracle 1.1 (spec rev 2: tight lists)"; }

/* ---- real spherical-harmonics constants (published 3DGS basis) ---- */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

static inline float bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t float_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* 3-term dot product as an fmaf chain in k order: fma(a2,b2, fma(a1,b1, a0*b0)).
 * This is bitwise what v_mfma_f32_*_f32 computes for K=3 with a zero accumulator. */
static inline float dot3_chain(float a0, float b0, float a1, float b1, float a2, float b2)
{
    return fmaf(a2, b2, fmaf(a1, b1, a0 * b0));
}

/* float -> tile index with clamping; equals min(hi, max(0, (int)f)) for every finite f,
 * and is well defined (0) for NaN, unlike a raw C cast. */
static inline int32_t clamp_trunc(float f, int32_t hi)
{
    if (!(f > 0.0f)) return 0;
    if (f >= (float)hi) return hi;
    return (int32_t)f;
}

/* Sigma3D = R diag(mod*s)^2 R^T, quaternion (w,x,y,z) used as given (the caller normalises:
 * /root/reference/src/gs/gaussian_model.py:109-110).  Same construction as the reference's
 * get_covariance (gaussian_model.py:38-42,127-128).  Output (xx,xy,xz,yy,yz,zz). */
static void cov3d_from_scale_rot(const float s[3], float mod, const float q[4], float cov[6])
{
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    float R[3][3];
    R[0][0] = 1.0f - 2.0f * (y * y + z * z);
    R[0][1] = 2.0f * (x * y - r * z);
    R[0][2] = 2.0f * (x * z + r * y);
    R[1][0] = 2.0f * (x * y + r * z);
    R[1][1] = 1.0f - 2.0f * (x * x + z * z);
    R[1][2] = 2.0f * (y * z - r * x);
    R[2][0] = 2.0f * (x * z - r * y);
    R[2][1] = 2.0f * (y * z + r * x);
    R[2][2] = 1.0f - 2.0f * (x * x + y * y);
    const float sx = mod * s[0], sy = mod * s[1], sz = mod * s[2];
    float M[3][3]; /* M[i][k] = R[i][k] * s_k */
    for (int i = 0; i < 3; ++i) {
        M[i][0] = R[i][0] * sx;
        M[i][1] = R[i][1] * sy;
        M[i][2] = R[i][2] * sz;
    }
    cov[0] = dot3_chain(M[0][0], M[0][0], M[0][1], M[0][1], M[0][2], M[0][2]);
    cov[1] = dot3_chain(M[0][0], M[1][0], M[0][1], M[1][1], M[0][2], M[1][2]);
    cov[2] = dot3_chain(M[0][0], M[2][0], M[0][1], M[2][1], M[0][2], M[2][2]);
    cov[3] = dot3_chain(M[1][0], M[1][0], M[1][1], M[1][1], M[1][2], M[1][2]);
    cov[4] = dot3_chain(M[1][0], M[2][0], M[1][1], M[2][1], M[1][2], M[2][2]);
    cov[5] = dot3_chain(M[2][0], M[2][0], M[2][1], M[2][1], M[2][2], M[2][2]);
}

/* 16 real-SH basis values for unit direction (x,y,z); b[k] multiplies coefficient k. */
static void sh_basis(int deg, float x, float y, float z, float b[16])
{
    b[0] = SH_C0;
    if (deg > 0) {
        b[1] = -(SH_C1 * y);
        b[2] = SH_C1 * z;
        b[3] = -(SH_C1 * x);
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z;
            const float xy = x * y, yz = y * z, xz = x * z;
            b[4] = SH_C2[0] * xy;
            b[5] = SH_C2[1] * yz;
            b[6] = SH_C2[2] * (2.0f * zz - xx - yy);
            b[7] = SH_C2[3] * xz;
            b[8] = SH_C2[4] * (xx - yy);
            if (deg > 2) {
                b[9] = SH_C3[0] * y * (3.0f * xx - yy);
                b[10] = SH_C3[1] * xy * z;
                b[11] = SH_C3[2] * y * (4.0f * zz - xx - yy);
                b[12] = SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
                b[13] = SH_C3[4] * x * (4.0f * zz - xx - yy);
                b[14] = SH_C3[5] * z * (xx - yy);
                b[15] = SH_C3[6] * x * (xx - 3.0f * yy);
            }
        }
    }
}


/* ---- tight-list predicate (cull_mode 1) ------------------------------------------------------------
 * A (Gaussian, tile) instance only matters if alpha = min(0.99, op*exp(power)) >= 1/255 at some pixel of
 * the tile.  With q = A dx^2 + 2B dx dy + C dy^2 (power = -q/2) that needs q <= 2 ln(255 op) somewhere.
 * The test below bounds q from BELOW over the continuous pixel rectangle of the tile (exact minimum of the
 * convex form: 0 if the centre is inside, else the smaller of the minima over the two edges facing the centre) and 2 ln(255 op) from
 * ABOVE (exponent + chord of log2 on the mantissa + 0.0861), adds a rounding margin, and keeps the instance
 * unless the lower bound clears the upper bound.  Dropped instances would have been skipped at every pixel,
 * so every pixel's arithmetic is unchanged.  No transcendental: identical bits on CPU and GPU. */
static inline float edge_min_q(float A, float B, float C, float r, float d_fixed, float lo, float hi)
{
    /* minimise over t in [lo,hi]:  A*d^2 + 2*B*d*t + C*t^2   (d = d_fixed, C > 0, r = B / C) */
    float t = -(d_fixed * r);
    t = fminf(hi, fmaxf(lo, t));
    return A * d_fixed * d_fixed + 2.0f * B * d_fixed * t + C * t * t;
}

int pgr_oracle_tile_may_contribute(const float xy[2], const float co[4], int32_t tx, int32_t ty, int32_t W, int32_t H)
{
    const float A = co[0], B = co[1], C = co[2], op = co[3];
    if (op < PGR_ALPHA_MIN) return 0;            /* alpha <= op < 1/255 at every pixel, exactly */
    if (!(A > 0.0f) || !(C > 0.0f)) return 1;    /* degenerate conic: no claim */
    const float x0 = (float)(tx * PGR_TILE), y0 = (float)(ty * PGR_TILE);
    const float x1 = fminf(x0 + (float)(PGR_TILE - 1), (float)(W - 1));
    const float y1 = fminf(y0 + (float)(PGR_TILE - 1), (float)(H - 1));
    const float mx = xy[0], my = xy[1];
    if (mx >= x0 && mx <= x1 && my >= y0 && my <= y1) return 1;
    const float dx0 = x0 - mx, dx1 = x1 - mx, dy0 = y0 - my, dy1 = y1 - my;
    const float rBC = B / C, rBA = B / A;
    /* q grows along every ray from the centre, so its minimum over the rectangle lies on an edge FACING the centre:
     * the vertical edge on the centre's side in x and the horizontal one on its side in y (when the centre is inside
     * the rectangle's x- or y-range that axis has no facing edge; the edge taken then only adds a larger candidate) */
    const float dxn = mx < x0 ? dx0 : dx1, dyn = my < y0 ? dy0 : dy1;
    const float q = fminf(edge_min_q(A, B, C, rBC, dxn, dy0, dy1), edge_min_q(C, B, A, rBA, dyn, dx0, dx1));
    /* upper bound of 2 ln(255 op): 255 op = m 2^e, log2 m <= (m-1) + 0.0861 */
    const float t = 255.0f * op;
    const uint32_t bits = float_to_bits(t);
    const float e = (float)((int32_t)((bits >> 23) & 0xffu) - 127);
    const float m = bits_to_float((bits & 0x007fffffu) | 0x3f800000u);
    const float tau = 1.3862944f * (e + (m - 1.0f) + 0.0861f);
    /* rounding margin: 1e-5 of the largest the three terms get anywhere in the rectangle, + 0.01 */
    const float DX = fmaxf(fabsf(dx0), fabsf(dx1)), DY = fmaxf(fabsf(dy0), fabsf(dy1));
    const float M = A * DX * DX + 2.0f * fabsf(B) * DX * DY + C * DY * DY;
    return q > tau + 0.00001f * M + 0.01f ? 0 : 1;   /* a NaN anywhere keeps the instance */
}

/* ---- preprocess: one Gaussian ---- */
static void preprocess_one(const PgrOracleIn *in, PgrOracleOut *out, int32_t i, int32_t grid_x, int32_t grid_y)
{
    const float *vm = in->viewmatrix, *pm = in->projmatrix;
    if (out->radii) out->radii[i] = 0;
    if (out->tiles_touched) out->tiles_touched[i] = 0;

    float px = in->means3d[3 * i + 0], py = in->means3d[3 * i + 1], pz = in->means3d[3 * i + 2];
    /* posed object: same arithmetic as the scene-composition kernel (pegasus_amd/csrc/compose.hip.h) */
    const float *P = NULL;
    if (in->object_id && in->object_id[i] > 0) {
        P = in->poses + (size_t)PGR_POSE_STRIDE * (size_t)(in->object_id[i] - 1);
        const float dx = px - P[12], dy = py - P[13], dz = pz - P[14];
        px = fmaf(P[2], dz, fmaf(P[1], dy, P[0] * dx)) + P[12] + P[9];
        py = fmaf(P[5], dz, fmaf(P[4], dy, P[3] * dx)) + P[13] + P[10];
        pz = fmaf(P[8], dz, fmaf(P[7], dy, P[6] * dx)) + P[14] + P[11];
    }

    /* view-space position: rows of the usual 4x3 transform, evaluated left to right */
    float tx = vm[0] * px + vm[4] * py + vm[8] * pz + vm[12];
    float ty = vm[1] * px + vm[5] * py + vm[9] * pz + vm[13];
    const float tz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
    if (tz <= PGR_NEAR_Z) return; /* near cull; no x/y frustum test */

    /* homogeneous projection */
    const float hx = pm[0] * px + pm[4] * py + pm[8] * pz + pm[12];
    const float hy = pm[1] * px + pm[5] * py + pm[9] * pz + pm[13];
    const float hw = pm[3] * px + pm[7] * py + pm[11] * pz + pm[15];
    const float p_w = 1.0f / (hw + 0.0000001f);
    const float ndc_x = hx * p_w, ndc_y = hy * p_w;

    /* 3D covariance */
    float cov[6];
    if (in->cov3d_precomp) {
        for (int k = 0; k < 6; ++k) cov[k] = in->cov3d_precomp[6 * i + k];
    } else if (P) {
        const float *q = in->rotations + 4 * i;
        const float inv = 1.0f / fmaxf(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12f);
        const float w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
        const float a = P[15], b = P[16], c = P[17], d = P[18];
        const float qp[4] = {a * w - b * x - c * y - d * z, a * x + b * w + c * z - d * y,
                             a * y - b * z + c * w + d * x, a * z + b * y - c * x + d * w};
        cov3d_from_scale_rot(in->scales + 3 * i, in->scale_modifier, qp, cov);
    } else {
        cov3d_from_scale_rot(in->scales + 3 * i, in->scale_modifier, in->rotations + 4 * i, cov);
    }
    if (out->cov3d) for (int k = 0; k < 6; ++k) out->cov3d[6 * i + k] = cov[k];

    /* EWA projection: cov2D = J W Sigma W^T J^T */
    const float focal_x = (float)in->width / (2.0f * in->tanfovx);
    const float focal_y = (float)in->height / (2.0f * in->tanfovy);
    const float limx = 1.3f * in->tanfovx, limy = 1.3f * in->tanfovy;
    const float txtz = tx / tz, tytz = ty / tz;
    tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
    const float j00 = focal_x / tz;
    const float j02 = -(focal_x * tx) / (tz * tz);
    const float j11 = focal_y / tz;
    const float j12 = -(focal_y * ty) / (tz * tz);
    /* W[r][c] = vm[c*4+r]; T = J*W (2x3), K=2 non-zero terms per entry as an fma chain */
    float T0[3], T1[3];
    for (int c = 0; c < 3; ++c) {
        T0[c] = fmaf(j02, vm[4 * c + 2], j00 * vm[4 * c + 0]);
        T1[c] = fmaf(j12, vm[4 * c + 2], j11 * vm[4 * c + 1]);
    }
    /* U = T * Sigma (2x3) */
    const float S[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
    float U0[3], U1[3];
    for (int c = 0; c < 3; ++c) {
        U0[c] = dot3_chain(T0[0], S[0][c], T0[1], S[1][c], T0[2], S[2][c]);
        U1[c] = dot3_chain(T1[0], S[0][c], T1[1], S[1][c], T1[2], S[2][c]);
    }
    const float c_xx = dot3_chain(U0[0], T0[0], U0[1], T0[1], U0[2], T0[2]) + PGR_LOWPASS;
    const float c_xy = dot3_chain(U0[0], T1[0], U0[1], T1[1], U0[2], T1[2]);
    const float c_yy = dot3_chain(U1[0], T1[0], U1[1], T1[1], U1[2], T1[2]) + PGR_LOWPASS;

    const float det = c_xx * c_yy - c_xy * c_xy;
    if (det == 0.0f) return;
    const float det_inv = 1.0f / det;
    const float con_x = c_yy * det_inv, con_y = -c_xy * det_inv, con_z = c_xx * det_inv;

    const float mid = 0.5f * (c_xx + c_yy);
    const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
    const float lambda1 = mid + disc, lambda2 = mid - disc;
    const float rad_f = ceilf(3.0f * sqrtf(fmaxf(lambda1, lambda2)));
    /* radius is stored as int32; clamp so the cast is defined for absurdly large splats */
    const int32_t radius = rad_f >= 2147483520.0f ? 2147483520 : (int32_t)rad_f;

    const float pix_x = ((ndc_x + 1.0f) * (float)in->width - 1.0f) * 0.5f;
    const float pix_y = ((ndc_y + 1.0f) * (float)in->height - 1.0f) * 0.5f;

    const float rf = (float)radius;
    const int32_t minx = clamp_trunc((pix_x - rf) / (float)PGR_TILE, grid_x);
    const int32_t miny = clamp_trunc((pix_y - rf) / (float)PGR_TILE, grid_y);
    const int32_t maxx = clamp_trunc((pix_x + rf + (float)(PGR_TILE - 1)) / (float)PGR_TILE, grid_x);
    const int32_t maxy = clamp_trunc((pix_y + rf + (float)(PGR_TILE - 1)) / (float)PGR_TILE, grid_y);
    const int32_t w = maxx - minx, h = maxy - miny;
    if (w <= 0 || h <= 0) return;

    /* colour */
    float rgb[3];
    if (in->colors_precomp) {
        rgb[0] = in->colors_precomp[3 * i + 0];
        rgb[1] = in->colors_precomp[3 * i + 1];
        rgb[2] = in->colors_precomp[3 * i + 2];
    } else {
        float dx = px - in->campos[0], dy = py - in->campos[1], dz = pz - in->campos[2];
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        dx = dx / len; dy = dy / len; dz = dz / len;
        if (P) {   /* the object's own frame: R^T d */
            const float ox = fmaf(P[6], dz, fmaf(P[3], dy, P[0] * dx));
            const float oy = fmaf(P[7], dz, fmaf(P[4], dy, P[1] * dx));
            const float oz = fmaf(P[8], dz, fmaf(P[5], dy, P[2] * dx));
            dx = ox; dy = oy; dz = oz;
        }
        float b[16];
        sh_basis(in->sh_degree, dx, dy, dz, b);
        const int ncoef = (in->sh_degree + 1) * (in->sh_degree + 1);
        const float *sh = in->shs + (size_t)i * in->sh_stride * 3;
        for (int c = 0; c < 3; ++c) {
            float acc = b[0] * sh[c];
            for (int k = 1; k < ncoef; ++k) acc = fmaf(b[k], sh[3 * k + c], acc);
            rgb[c] = fmaxf(acc + 0.5f, 0.0f);
        }
    }

    if (out->depth) out->depth[i] = tz;
    if (out->radii) out->radii[i] = radius;
    if (out->xy) { out->xy[2 * i] = pix_x; out->xy[2 * i + 1] = pix_y; }
    if (out->conic_opacity) {
        out->conic_opacity[4 * i + 0] = con_x;
        out->conic_opacity[4 * i + 1] = con_y;
        out->conic_opacity[4 * i + 2] = con_z;
        out->conic_opacity[4 * i + 3] = in->opacities[i];
    }
    if (out->rgb) { out->rgb[3 * i] = rgb[0]; out->rgb[3 * i + 1] = rgb[1]; out->rgb[3 * i + 2] = rgb[2]; }
    if (out->tiles_touched) out->tiles_touched[i] = w * h;
}

static int check_in(const PgrOracleIn *in)
{
    if (!in || in->n < 0 || in->width <= 0 || in->height <= 0) return -1;
    if (in->n > 0) {
        if (!in->means3d || !in->opacities) return -1;
        if ((in->shs == NULL) == (in->colors_precomp == NULL)) return -1;
        const int have_sr = in->scales != NULL && in->rotations != NULL;
        if (have_sr == (in->cov3d_precomp != NULL)) return -1;
        if (in->shs && (in->sh_degree < 0 || in->sh_degree > 3 ||
                        in->sh_stride < (in->sh_degree + 1) * (in->sh_degree + 1))) return -1;
        if (in->object_id && (!in->poses || in->k_objects <= 0 || in->cov3d_precomp)) return -1;
    }
    return 0;
}

int pgr_oracle_preprocess(const PgrOracleIn *in, PgrOracleOut *out, int num_threads)
{
    if (check_in(in) || !out) return -1;
    const int32_t grid_x = (in->width + PGR_TILE - 1) / PGR_TILE;
    const int32_t grid_y = (in->height + PGR_TILE - 1) / PGR_TILE;
    (void)num_threads;
#pragma omp parallel for schedule(static) num_threads(num_threads > 0 ? num_threads : 1)
    for (int32_t i = 0; i < in->n; ++i) preprocess_one(in, out, i, grid_x, grid_y);
    return 0;
}

/* stable LSD radix sort of (key,value) pairs over bits [0,nbits) -- SURVEY.md section 8a row a8.
 * Every pass is parallel over contiguous chunks of the input: per-chunk digit histograms, one exclusive scan in
 * (digit, chunk) order -- which is what keeps the pass stable -- and a scatter in which every chunk owns its cursors.
 * Same permutation as the serial form for any number of threads. */
static void radix_sort_pairs(uint64_t *keys, uint32_t *vals, int64_t n, int nbits, int num_threads)
{
    if (n <= 1) return;
    uint64_t *k2 = (uint64_t *)malloc((size_t)n * sizeof(uint64_t));
    uint32_t *v2 = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    uint64_t *ka = keys, *kb = k2;
    uint32_t *va = vals, *vb = v2;
    int chunks = num_threads < 1 ? 1 : (num_threads > 32 ? 32 : num_threads);    /* a scatter pass is memory-bound long before 32 */
    const char *sort_threads = getenv("PGR_ORACLE_SORT_THREADS");
    if (sort_threads) { const int st = atoi(sort_threads); chunks = st > 0 ? st : 1; }
    if ((int64_t)chunks > n / 4096 + 1) chunks = (int)(n / 4096 + 1);
    int64_t *hist = (int64_t *)malloc((size_t)chunks * 256 * sizeof(int64_t));
    const int64_t per = (n + chunks - 1) / chunks;
    for (int shift = 0; shift < nbits; shift += 8) {
#pragma omp parallel for schedule(static) num_threads(chunks)
        for (int c = 0; c < chunks; ++c) {
            int64_t *h = hist + (size_t)c * 256;
            memset(h, 0, 256 * sizeof(int64_t));
            const int64_t lo = (int64_t)c * per, hi = lo + per < n ? lo + per : n;
            for (int64_t i = lo; i < hi; ++i) h[(ka[i] >> shift) & 0xFF]++;
        }
        int64_t acc = 0;
        for (int b = 0; b < 256; ++b)
            for (int c = 0; c < chunks; ++c) {
                const int64_t v = hist[(size_t)c * 256 + b];
                hist[(size_t)c * 256 + b] = acc;
                acc += v;
            }
#pragma omp parallel for schedule(static) num_threads(chunks)
        for (int c = 0; c < chunks; ++c) {
            int64_t *h = hist + (size_t)c * 256;
            const int64_t lo = (int64_t)c * per, hi = lo + per < n ? lo + per : n;
            for (int64_t i = lo; i < hi; ++i) {
                const int64_t p = h[(ka[i] >> shift) & 0xFF]++;
                kb[p] = ka[i];
                vb[p] = va[i];
            }
        }
        uint64_t *tk = ka; ka = kb; kb = tk;
        uint32_t *tv = va; va = vb; vb = tv;
    }
    if (ka != keys) {
        memcpy(keys, ka, (size_t)n * sizeof(uint64_t));
        memcpy(vals, va, (size_t)n * sizeof(uint32_t));
    }
    free(hist);
    free(k2);
    free(v2);
}

/* ---- compositor: one pixel ---- */
typedef struct { float x, y, hx, ny, hz, op, r, g, b, depth; } Splat;

static void composite_tile(const PgrOracleIn *in, PgrOracleOut *out, const float *xy, const float *conop,
                           const float *rgb, const float *depth, const uint32_t *gauss_sorted,
                           uint32_t start, uint32_t end, int32_t tile_x, int32_t tile_y, Splat *scratch)
{
    const int32_t W = in->width, H = in->height;
    const uint32_t cnt = end - start;
    for (uint32_t j = 0; j < cnt; ++j) {
        const uint32_t g = gauss_sorted[start + j];
        Splat *s = &scratch[j];
        s->x = xy[2 * g]; s->y = xy[2 * g + 1];
        /* exact power-of-two / sign scalings of the conic */
        s->hx = -0.5f * conop[4 * g + 0];
        s->ny = -conop[4 * g + 1];
        s->hz = -0.5f * conop[4 * g + 2];
        s->op = conop[4 * g + 3];
        s->r = rgb[3 * g]; s->g = rgb[3 * g + 1]; s->b = rgb[3 * g + 2];
        s->depth = depth[g];
    }
    for (int32_t ly = 0; ly < PGR_TILE; ++ly) {
        const int32_t py = tile_y * PGR_TILE + ly;
        if (py >= H) break;
        for (int32_t lx = 0; lx < PGR_TILE; ++lx) {
            const int32_t px = tile_x * PGR_TILE + lx;
            if (px >= W) break;
            const float pxf = (float)px, pyf = (float)py;
            float T = 1.0f, Cr = 0.0f, Cg = 0.0f, Cb = 0.0f, D = 0.0f;
            uint32_t last = 0;
            uint8_t amb = 0;
            /* Ambiguity tracking.  exp() is the one operation that is not bit-identical between this
             * oracle (glibc expf) and the GPU (v_exp_f32 of power*log2e): the relative difference of alpha is
             * bounded by ALPHA_RELERR.  T inherits a relative error that grows by alpha*err/(1-alpha) per
             * blend; a pixel is flagged when a threshold comparison lies inside those bounds. */
            const float ALPHA_RELERR = 1.0e-6f;
            float T_relerr = 0.0f;
            for (uint32_t j = 0; j < cnt; ++j) {
                const Splat *s = &scratch[j];
                const float dx = s->x - pxf, dy = s->y - pyf;
                /* power = -0.5*(A dx^2 + C dy^2) - B dx dy, evaluated as
                 *   fma(dx, fma(hx,dx, ny*dy), (hz*dy)*dy)                                     */
                const float power = fmaf(dx, fmaf(s->hx, dx, s->ny * dy), (s->hz * dy) * dy);
                if (power > 0.0f) continue;
                const float e = expf(power);
                const float araw = s->op * e;
                const float alpha = fminf(PGR_ALPHA_MAX, araw);
                if (fabsf(araw - PGR_ALPHA_MIN) <= ALPHA_RELERR * PGR_ALPHA_MIN) amb = 1;
                if (alpha < PGR_ALPHA_MIN) continue;
                const float test_T = fmaf(-alpha, T, T);
                const float step_err = araw < PGR_ALPHA_MAX ? alpha * ALPHA_RELERR / (1.0f - alpha) : 0.0f;
                if (fabsf(test_T - PGR_T_EPS) <= (T_relerr + step_err + 2.0e-7f) * PGR_T_EPS) amb = 1;
                if (test_T < PGR_T_EPS) break; /* this entry is NOT blended */
                const float w = alpha * T;
                Cr = fmaf(s->r, w, Cr);
                Cg = fmaf(s->g, w, Cg);
                Cb = fmaf(s->b, w, Cb);
                D = fmaf(s->depth, w, D);
                T = test_T;
                T_relerr += step_err + 1.0e-7f;
                last = j + 1;
            }
            const size_t pix = (size_t)py * W + px, P = (size_t)W * H;
            if (out->out_color) {
                out->out_color[0 * P + pix] = fmaf(T, in->bg[0], Cr);
                out->out_color[1 * P + pix] = fmaf(T, in->bg[1], Cg);
                out->out_color[2 * P + pix] = fmaf(T, in->bg[2], Cb);
            }
            if (out->out_depth) {
                /* depth rule switch: un-normalised expected depth, or divided by the blended weights' total 1 - T */
                float d = D;
                if (in->depth_mode == 1) {
                    const float wsum = 1.0f - T;
                    d = wsum > 0.0f ? D / wsum : 0.0f;
                }
                out->out_depth[pix] = d;
            }
            if (out->final_T) out->final_T[pix] = T;
            if (out->n_contrib) out->n_contrib[pix] = last;
            if (out->ambig) out->ambig[pix] = amb;
        }
    }
}

int pgr_oracle_forward(const PgrOracleIn *in, PgrOracleOut *out, int num_threads)
{
    if (check_in(in) || !out) return -1;
    const int32_t W = in->width, H = in->height, n = in->n;
    const int32_t grid_x = (W + PGR_TILE - 1) / PGR_TILE, grid_y = (H + PGR_TILE - 1) / PGR_TILE;
    const int32_t tiles = grid_x * grid_y;
    const size_t P = (size_t)W * H;
    if (num_threads < 1) num_threads = 1;

    /* outputs start zero-filled; with n == 0 they STAY zero (no background) */
    if (out->out_color) memset(out->out_color, 0, 3 * P * sizeof(float));
    if (out->out_depth) memset(out->out_depth, 0, P * sizeof(float));
    if (out->final_T) memset(out->final_T, 0, P * sizeof(float));
    if (out->n_contrib) memset(out->n_contrib, 0, P * sizeof(uint32_t));
    if (out->ambig) memset(out->ambig, 0, P);
    if (out->ranges) memset(out->ranges, 0, (size_t)tiles * 2 * sizeof(uint32_t));
    out->num_instances = 0;
    if (n == 0) return 0;

    /* private per-Gaussian arrays where the caller did not ask for them */
    PgrOracleOut o = *out;
    float *xy = o.xy ? o.xy : (float *)calloc((size_t)n * 2, 4);
    float *depth = o.depth ? o.depth : (float *)calloc((size_t)n, 4);
    float *conop = o.conic_opacity ? o.conic_opacity : (float *)calloc((size_t)n * 4, 4);
    float *rgb = o.rgb ? o.rgb : (float *)calloc((size_t)n * 3, 4);
    int32_t *radii = o.radii ? o.radii : (int32_t *)calloc((size_t)n, 4);
    int32_t *tt = o.tiles_touched ? o.tiles_touched : (int32_t *)calloc((size_t)n, 4);
    o.xy = xy; o.depth = depth; o.conic_opacity = conop; o.rgb = rgb; o.radii = radii; o.tiles_touched = tt;

    const int timing = getenv("PGR_ORACLE_TIMING") != NULL;
    double t_[8]; int ti_ = 0;
    t_[ti_++] = pgr_wtime();
    int rc = pgr_oracle_preprocess(in, &o, num_threads);
    t_[ti_++] = pgr_wtime();

    /* a6: inclusive scan of the per-Gaussian instance counts (cull_mode 1: only the tiles that may contribute) */
    int64_t total = 0;
    int64_t *offs = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    int32_t *kept = (int32_t *)malloc((size_t)n * sizeof(int32_t));
#pragma omp parallel for schedule(static) num_threads(num_threads)
    for (int32_t i = 0; i < n; ++i) {
        kept[i] = tt[i];
        if (in->cull_mode == 1 && tt[i] > 0) {
            const float rf = (float)radii[i], px = xy[2 * i], py = xy[2 * i + 1];
            const int32_t minx = clamp_trunc((px - rf) / (float)PGR_TILE, grid_x);
            const int32_t miny = clamp_trunc((py - rf) / (float)PGR_TILE, grid_y);
            const int32_t maxx = clamp_trunc((px + rf + (float)(PGR_TILE - 1)) / (float)PGR_TILE, grid_x);
            const int32_t maxy = clamp_trunc((py + rf + (float)(PGR_TILE - 1)) / (float)PGR_TILE, grid_y);
            int32_t c = 0;
            for (int32_t y = miny; y < maxy; ++y)
                for (int32_t x = minx; x < maxx; ++x)
                    c += pgr_oracle_tile_may_contribute(xy + 2 * i, conop + 4 * i, x, y, W, H);
            kept[i] = c;
        }
    }
    /* emission order = ascending tie index (the position itself without one): the stable sort below then breaks
     * exact depth ties by it */
    if (in->tie_index) {
        int32_t *seq = (int32_t *)malloc((size_t)n * sizeof(int32_t));
        for (int32_t i = 0; i < n; ++i) seq[i] = -1;
        for (int32_t i = 0; i < n; ++i) {
            const int32_t k = in->tie_index[i];
            if (k >= 0 && k < n) seq[k] = i;
        }
        for (int32_t j = 0; j < n; ++j) {
            const int32_t i = seq[j];
            if (i < 0) { rc = -1; continue; }       /* not a permutation */
            total += kept[i]; offs[i] = total;
        }
        free(seq);
    } else {
        for (int32_t i = 0; i < n; ++i) { total += kept[i]; offs[i] = total; }
    }
    out->num_instances = total;
    t_[ti_++] = pgr_wtime();

    uint64_t *keys = NULL;
    uint32_t *vals = NULL;
    uint32_t *ranges = NULL;
    if (rc == 0 && total > 0) {
        if ((out->keys_sorted || out->gauss_sorted) && out->cap_instances < total) rc = -2;
    }
    if (rc == 0 && total > 0) {
        keys = out->keys_sorted ? out->keys_sorted : (uint64_t *)malloc((size_t)total * 8);
        vals = out->gauss_sorted ? out->gauss_sorted : (uint32_t *)malloc((size_t)total * 4);
        /* a7: emission, Gaussian-major, then tile row-major; recompute the rectangle exactly as preprocess did */
#pragma omp parallel for schedule(static) num_threads(num_threads)
        for (int32_t i = 0; i < n; ++i) {
            if (kept[i] == 0) continue;
            const float rf = (float)radii[i], px = xy[2 * i], py = xy[2 * i + 1];
            const int32_t minx = clamp_trunc((px - rf) / (float)PGR_TILE, grid_x);
            const int32_t miny = clamp_trunc((py - rf) / (float)PGR_TILE, grid_y);
            const int32_t maxx = clamp_trunc((px + rf + (float)(PGR_TILE - 1)) / (float)PGR_TILE, grid_x);
            const int32_t maxy = clamp_trunc((py + rf + (float)(PGR_TILE - 1)) / (float)PGR_TILE, grid_y);
            int64_t off = offs[i] - kept[i];
            const uint64_t dbits = float_to_bits(depth[i]);
            for (int32_t y = miny; y < maxy; ++y)
                for (int32_t x = minx; x < maxx; ++x) {
                    if (in->cull_mode == 1 && !pgr_oracle_tile_may_contribute(xy + 2 * i, conop + 4 * i, x, y, W, H))
                        continue;
                    keys[off] = ((uint64_t)(uint32_t)(y * grid_x + x) << 32) | dbits;
                    vals[off] = (uint32_t)i;
                    ++off;
                }
        }
        t_[ti_++] = pgr_wtime();
        /* a8: stable sort over bits [0, 32 + ceil(log2 tiles)) */
        int tbits = 0;
        while ((1 << tbits) < tiles) ++tbits;
        radix_sort_pairs(keys, vals, total, 32 + tbits, num_threads);

        t_[ti_++] = pgr_wtime();
        /* a9: tile ranges */
        ranges = out->ranges ? out->ranges : (uint32_t *)calloc((size_t)tiles * 2, 4);
#pragma omp parallel for schedule(static) num_threads(num_threads)
        for (int64_t k = 0; k < total; ++k) {
            const uint32_t t = (uint32_t)(keys[k] >> 32);
            if (k == 0 || (uint32_t)(keys[k - 1] >> 32) != t) ranges[2 * t] = (uint32_t)k;
            if (k == total - 1 || (uint32_t)(keys[k + 1] >> 32) != t) ranges[2 * t + 1] = (uint32_t)(k + 1);
        }
    }
    free(offs);
    free(kept);

    t_[ti_++] = pgr_wtime();
    /* a10: compositor, every tile (empty tiles produce bg colour, zero depth, T = 1) */
    if (rc == 0) {
        uint32_t maxlen = 0;
        if (ranges) for (int32_t t = 0; t < tiles; ++t) {
            const uint32_t l = ranges[2 * t + 1] - ranges[2 * t];
            if (l > maxlen) maxlen = l;
        }
#pragma omp parallel num_threads(num_threads)
        {
            Splat *scratch = (Splat *)malloc(((size_t)maxlen + 1) * sizeof(Splat));
#pragma omp for schedule(dynamic, 4)
            for (int32_t t = 0; t < tiles; ++t) {
                const uint32_t s = ranges ? ranges[2 * t] : 0, e = ranges ? ranges[2 * t + 1] : 0;
                composite_tile(in, out, xy, conop, rgb, depth, vals, s, e, t % grid_x, t / grid_x, scratch);
            }
            free(scratch);
        }
    }

    t_[ti_++] = pgr_wtime();
    if (timing) {
        fprintf(stderr, "pgr_oracle_forward (%d threads):", num_threads);
        for (int k = 1; k < ti_; ++k) fprintf(stderr, " %.3f", t_[k] - t_[k - 1]);
        fprintf(stderr, " s  [preprocess, count+scan, emission, sort, ranges, composite]\n");
    }
    if (keys && keys != out->keys_sorted) free(keys);
    if (vals && vals != out->gauss_sorted) free(vals);
    if (ranges && ranges != out->ranges) free(ranges);
    if (!out->xy) free(xy);
    if (!out->depth) free(depth);
    if (!out->conic_opacity) free(conop);
    if (!out->rgb) free(rgb);
    if (!out->radii) free(radii);
    if (!out->tiles_touched) free(tt);
    return rc;
}

int pgr_oracle_mark_visible(int32_t n, const float *means3d, const float *vm, uint8_t *present)
{
    if (n < 0 || (n > 0 && (!means3d || !vm || !present))) return -1;
    for (int32_t i = 0; i < n; ++i) {
        const float px = means3d[3 * i], py = means3d[3 * i + 1], pz = means3d[3 * i + 2];
        const float tz = vm[2] * px + vm[6] * py + vm[10] * pz + vm[14];
        present[i] = tz > PGR_NEAR_Z ? 1 : 0;
    }
    return 0;
}

/* reference: /root/reference/src/gs/render.py:60-63 and :89-93 -- np.linalg.norm(img - c, axis=2) <= 0.1.
 * numpy evaluates the float32 difference, squares, sums over the 3 channels in order, sqrt. */
int pgr_oracle_color_masks(const float *img, int32_t W, int32_t H, const float *colors, int32_t k, float thr,
                           uint8_t *masks)
{
    if (!img || !colors || !masks || W <= 0 || H <= 0 || k < 0) return -1;
    const size_t P = (size_t)W * H;
    for (int32_t c = 0; c < k; ++c)
        for (size_t p = 0; p < P; ++p) {
            const float d0 = img[p] - colors[3 * c], d1 = img[P + p] - colors[3 * c + 1],
                        d2 = img[2 * P + p] - colors[3 * c + 2];
            const float dist = sqrtf(d0 * d0 + d1 * d1 + d2 * d2);
            masks[(size_t)c * P + p] = dist <= thr ? 1 : 0;
        }
    return 0;
}

/* reference: /root/reference/pegasus.py:347 (rgb*255 -> uint8, no clamp) and :355 (depth*1000 -> uint16 mm) */
int pgr_oracle_quantize(const float *img, const float *depth, int32_t W, int32_t H, uint8_t *rgb_hwc,
                        uint16_t *depth_mm)
{
    if (W <= 0 || H <= 0) return -1;
    const size_t P = (size_t)W * H;
    for (size_t p = 0; p < P; ++p) {
        if (img && rgb_hwc)
            for (int c = 0; c < 3; ++c) {
                float v = img[(size_t)c * P + p] * 255.0f;
                v = fminf(fmaxf(v, -2147483520.0f), 2147483520.0f);
                rgb_hwc[3 * p + c] = (uint8_t)((int32_t)v & 0xFF);
            }
        if (depth && depth_mm) {
            float v = depth[p] * 1000.0f;
            v = fminf(fmaxf(v, -2147483520.0f), 2147483520.0f);
            depth_mm[p] = (uint16_t)((int32_t)v & 0xFFFF);
        }
    }
    return 0;
}
