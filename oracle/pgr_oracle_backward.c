/*
 * pgr_oracle_backward.c -- CPU ORACLE of the rasterizer's BACKWARD pass (test infrastructure, NOT product code;
 * see pgr_oracle.h).  PARITY UNPINNED, as for the forward: the reference's differentiable rasterizer is the
 * absent, un-pinned submodule (/root/reference/.gitmodules:1-3, /root/reference/setup.sh:19); the only in-tree
 * user is training (/root/reference/src/gs/gs_training.py:7,46).  This file differentiates the forward that
 * pgr_oracle.c restates; it is pinned by finite differences of a dense float64 forward (tests/test_backward.py).
 *
 * Conventions (the published 3DGS backward, which the drop-in surface has to reproduce for training code):
 *   - the 0.99 clamp of alpha is NOT differentiated (alpha = o*G is used for d/do, d/dG even when clamped);
 *   - the screen-space mean gradient dL_dmean2D is returned in NDC-scaled units (pixel gradient * 0.5*W, 0.5*H),
 *     which is what densification statistics read from viewspace_points.grad;
 *   - colours clamped at 0 by max(.,0) pass no gradient;
 *   - view-space x/y clamped to 1.3*tanfov pass no gradient through the clamped coordinate of the Jacobian.
 * Gradients are accumulated in double precision; inputs are the float32 quantities of the forward.
 */
#include "pgr_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static const double SH_C0 = 0.28209479177387814;
static const double SH_C1 = 0.4886025119029199;
static const double SH_C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792,
                                0.5462742152960396};
static const double SH_C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
                                -0.4570457994644658, 1.445305721320277, -0.5900435899266435};

/* basis values b[16] and their partial derivatives wrt the (unit) direction components */
static void sh_basis_grad(int deg, double x, double y, double z, double b[16], double bx[16], double by[16], double bz[16])
{
    for (int k = 0; k < 16; ++k) b[k] = bx[k] = by[k] = bz[k] = 0.0;
    b[0] = SH_C0;
    if (deg < 1) return;
    b[1] = -SH_C1 * y; by[1] = -SH_C1;
    b[2] = SH_C1 * z;  bz[2] = SH_C1;
    b[3] = -SH_C1 * x; bx[3] = -SH_C1;
    if (deg < 2) return;
    const double xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[4] = SH_C2[0] * xy;                  bx[4] = SH_C2[0] * y;  by[4] = SH_C2[0] * x;
    b[5] = SH_C2[1] * yz;                  by[5] = SH_C2[1] * z;  bz[5] = SH_C2[1] * y;
    b[6] = SH_C2[2] * (2 * zz - xx - yy);  bx[6] = SH_C2[2] * -2 * x; by[6] = SH_C2[2] * -2 * y; bz[6] = SH_C2[2] * 4 * z;
    b[7] = SH_C2[3] * xz;                  bx[7] = SH_C2[3] * z;  bz[7] = SH_C2[3] * x;
    b[8] = SH_C2[4] * (xx - yy);           bx[8] = SH_C2[4] * 2 * x; by[8] = SH_C2[4] * -2 * y;
    if (deg < 3) return;
    b[9] = SH_C3[0] * y * (3 * xx - yy);   bx[9] = SH_C3[0] * 6 * xy; by[9] = SH_C3[0] * (3 * xx - 3 * yy);
    b[10] = SH_C3[1] * xy * z;             bx[10] = SH_C3[1] * yz; by[10] = SH_C3[1] * xz; bz[10] = SH_C3[1] * xy;
    b[11] = SH_C3[2] * y * (4 * zz - xx - yy);
    bx[11] = SH_C3[2] * -2 * xy; by[11] = SH_C3[2] * (4 * zz - xx - 3 * yy); bz[11] = SH_C3[2] * 8 * yz;
    b[12] = SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy);
    bx[12] = SH_C3[3] * -6 * xz; by[12] = SH_C3[3] * -6 * yz; bz[12] = SH_C3[3] * (6 * zz - 3 * xx - 3 * yy);
    b[13] = SH_C3[4] * x * (4 * zz - xx - yy);
    bx[13] = SH_C3[4] * (4 * zz - 3 * xx - yy); by[13] = SH_C3[4] * -2 * xy; bz[13] = SH_C3[4] * 8 * xz;
    b[14] = SH_C3[5] * z * (xx - yy);      bx[14] = SH_C3[5] * 2 * xz; by[14] = SH_C3[5] * -2 * yz; bz[14] = SH_C3[5] * (xx - yy);
    b[15] = SH_C3[6] * x * (xx - 3 * yy);  bx[15] = SH_C3[6] * (3 * xx - 3 * yy); by[15] = SH_C3[6] * -6 * xy;
}

/* ---- per-pixel backward of the compositor (reverse traversal of the tile's list) ---- */
static void composite_backward(const PgrOracleIn *in, const float *xy, const float *conop, const float *rgb,
                               const float *depth, const uint32_t *gauss_sorted, const uint32_t *ranges,
                               const float *final_T, const uint32_t *n_contrib, const float *g_color,
                               const float *g_depth, double *g_xy, double *g_conic, double *g_opacity, double *g_rgb,
                               double *g_z)
{
    const int32_t W = in->width, H = in->height;
    const int32_t grid_x = (W + PGR_TILE - 1) / PGR_TILE;
    const size_t P = (size_t)W * H;
    for (int32_t py = 0; py < H; ++py)
        for (int32_t px = 0; px < W; ++px) {
            const size_t pix = (size_t)py * W + px;
            const uint32_t tile = (uint32_t)((py / PGR_TILE) * grid_x + px / PGR_TILE);
            const uint32_t start = ranges[2 * tile];
            const uint32_t last = n_contrib[pix];
            if (last == 0) continue;
            const double gC[3] = {g_color[pix], g_color[P + pix], g_color[2 * P + pix]};
            const double gD = g_depth ? g_depth[pix] : 0.0;
            double T = final_T[pix];
            double S[3] = {T * in->bg[0], T * in->bg[1], T * in->bg[2]};   /* colour behind the current entry */
            double SD = 0.0;
            for (int64_t j = (int64_t)last - 1; j >= 0; --j) {
                const uint32_t g = gauss_sorted[start + j];
                const double dx = (double)xy[2 * g] - (double)px, dy = (double)xy[2 * g + 1] - (double)py;
                const double A = conop[4 * g], B = conop[4 * g + 1], C = conop[4 * g + 2], o = conop[4 * g + 3];
                /* the forward's float32 decisions are reproduced with its own float32 arithmetic */
                const float dxf = xy[2 * g] - (float)px, dyf = xy[2 * g + 1] - (float)py;
                const float powf_ = fmaf(dxf, fmaf(-0.5f * conop[4 * g], dxf, -conop[4 * g + 1] * dyf),
                                         (-0.5f * conop[4 * g + 2] * dyf) * dyf);
                if (powf_ > 0.0f) continue;
                const float alphaf = fminf(PGR_ALPHA_MAX, conop[4 * g + 3] * expf(powf_));
                if (alphaf < PGR_ALPHA_MIN) continue;
                const double power = -0.5 * (A * dx * dx + C * dy * dy) - B * dx * dy;
                const double G = exp(power);
                const double alpha = fmin(0.99, o * G);
                T = T / (1.0 - alpha);                      /* transmittance in FRONT of this entry */
                const double w = alpha * T;
                const double c[3] = {rgb[3 * g], rgb[3 * g + 1], rgb[3 * g + 2]};
                const double z = depth[g];
                double dL_dalpha = 0.0;
                for (int ch = 0; ch < 3; ++ch) {
                    dL_dalpha += gC[ch] * (T * c[ch] - S[ch] / (1.0 - alpha));
                    g_rgb[3 * g + ch] += w * gC[ch];
                }
                dL_dalpha += gD * (T * z - SD / (1.0 - alpha));
                g_z[g] += w * gD;
                for (int ch = 0; ch < 3; ++ch) S[ch] += w * c[ch];
                SD += w * z;
                const double dL_dG = o * dL_dalpha;
                g_opacity[g] += G * dL_dalpha;
                const double dL_dpower = G * dL_dG;
                g_conic[3 * g + 0] += -0.5 * dx * dx * dL_dpower;
                g_conic[3 * g + 1] += -dx * dy * dL_dpower;
                g_conic[3 * g + 2] += -0.5 * dy * dy * dL_dpower;
                g_xy[2 * g + 0] += -(A * dx + B * dy) * dL_dpower;
                g_xy[2 * g + 1] += -(C * dy + B * dx) * dL_dpower;
            }
        }
}

typedef struct PgrOracleGrads {
    float *means2d;     /* [n,3] NDC-scaled screen gradient (z component 0) */
    float *means3d;     /* [n,3] */
    float *opacities;   /* [n]   */
    float *colors;      /* [n,3] gradient wrt the per-Gaussian rgb (after SH evaluation / of colors_precomp) */
    float *shs;         /* [n,sh_stride,3] or NULL */
    float *cov3d;       /* [n,6] */
    float *scales;      /* [n,3] or NULL */
    float *rotations;   /* [n,4] or NULL */
} PgrOracleGrads;

int pgr_oracle_backward(const PgrOracleIn *in, const float *g_color, const float *g_depth, PgrOracleGrads *gr,
                        int num_threads)
{
    if (!in || !g_color || !gr || in->n < 0) return -1;
    const int32_t n = in->n, W = in->width, H = in->height;
    const size_t P = (size_t)W * H;
    const int32_t tiles = ((W + PGR_TILE - 1) / PGR_TILE) * ((H + PGR_TILE - 1) / PGR_TILE);
    if (n == 0) return 0;

    /* forward, keeping every intermediate */
    PgrOracleOut f;
    memset(&f, 0, sizeof(f));
    f.radii = calloc(n, 4); f.tiles_touched = calloc(n, 4); f.xy = calloc((size_t)n * 2, 4); f.depth = calloc(n, 4);
    f.conic_opacity = calloc((size_t)n * 4, 4); f.rgb = calloc((size_t)n * 3, 4); f.cov3d = calloc((size_t)n * 6, 4);
    PgrOracleIn in0 = *in;
    in0.cull_mode = 0;
    pgr_oracle_preprocess(&in0, &f, num_threads);
    int64_t cap = 0;
    for (int32_t i = 0; i < n; ++i) cap += f.tiles_touched[i];
    f.keys_sorted = malloc((size_t)(cap + 1) * 8); f.gauss_sorted = malloc((size_t)(cap + 1) * 4); f.cap_instances = cap + 1;
    f.ranges = calloc((size_t)tiles * 2, 4);
    f.out_color = malloc(3 * P * 4); f.out_depth = malloc(P * 4); f.final_T = malloc(P * 4); f.n_contrib = malloc(P * 4);
    int rc = pgr_oracle_forward(&in0, &f, num_threads);
    if (rc) return rc;

    double *g_xy = calloc((size_t)n * 2, 8), *g_conic = calloc((size_t)n * 3, 8), *g_op = calloc(n, 8);
    double *g_rgb = calloc((size_t)n * 3, 8), *g_z = calloc(n, 8);
    composite_backward(in, f.xy, f.conic_opacity, f.rgb, f.depth, f.gauss_sorted, f.ranges, f.final_T, f.n_contrib,
                       g_color, g_depth, g_xy, g_conic, g_op, g_rgb, g_z);

    const float *vm = in->viewmatrix, *pm = in->projmatrix;
    const double fx = (double)W / (2.0 * in->tanfovx), fy = (double)H / (2.0 * in->tanfovy);
    const double limx = 1.3 * in->tanfovx, limy = 1.3 * in->tanfovy;
    for (int32_t i = 0; i < n; ++i) {
        if (gr->means2d) { gr->means2d[3 * i] = gr->means2d[3 * i + 1] = gr->means2d[3 * i + 2] = 0.f; }
        if (gr->means3d) { gr->means3d[3 * i] = gr->means3d[3 * i + 1] = gr->means3d[3 * i + 2] = 0.f; }
        if (gr->opacities) gr->opacities[i] = 0.f;
        if (gr->colors) { gr->colors[3 * i] = gr->colors[3 * i + 1] = gr->colors[3 * i + 2] = 0.f; }
        if (gr->cov3d) for (int k = 0; k < 6; ++k) gr->cov3d[6 * i + k] = 0.f;
        if (gr->scales) for (int k = 0; k < 3; ++k) gr->scales[3 * i + k] = 0.f;
        if (gr->rotations) for (int k = 0; k < 4; ++k) gr->rotations[4 * i + k] = 0.f;
        if (gr->shs) for (int k = 0; k < in->sh_stride * 3; ++k) gr->shs[(size_t)i * in->sh_stride * 3 + k] = 0.f;
        if (f.radii[i] <= 0) continue;

        const double p[3] = {in->means3d[3 * i], in->means3d[3 * i + 1], in->means3d[3 * i + 2]};
        double gp[3] = {0, 0, 0};

        /* ---- screen position: pix = ((ndc+1) S - 1)/2, ndc = h_xy / (h_w + 1e-7) */
        const double gndc[2] = {g_xy[2 * i] * 0.5 * W, g_xy[2 * i + 1] * 0.5 * H};
        if (gr->means2d) { gr->means2d[3 * i] = (float)gndc[0]; gr->means2d[3 * i + 1] = (float)gndc[1]; }
        {
            const double hx = pm[0] * p[0] + pm[4] * p[1] + pm[8] * p[2] + pm[12];
            const double hy = pm[1] * p[0] + pm[5] * p[1] + pm[9] * p[2] + pm[13];
            const double hw = pm[3] * p[0] + pm[7] * p[1] + pm[11] * p[2] + pm[15];
            const double mw = 1.0 / (hw + 0.0000001);
            for (int k = 0; k < 3; ++k)
                gp[k] += gndc[0] * (pm[4 * k + 0] * mw - hx * mw * mw * pm[4 * k + 3]) +
                         gndc[1] * (pm[4 * k + 1] * mw - hy * mw * mw * pm[4 * k + 3]);
        }

        /* ---- view-space position */
        double t[3];
        for (int r = 0; r < 3; ++r) t[r] = vm[r] * p[0] + vm[4 + r] * p[1] + vm[8 + r] * p[2] + vm[12 + r];
        double gt[3] = {0, 0, g_z[i]};                       /* depth output = t_z */

        /* ---- conic -> cov2D (a,b,c) */
        const double S3[3][3] = {{f.cov3d[6 * i], f.cov3d[6 * i + 1], f.cov3d[6 * i + 2]},
                                 {f.cov3d[6 * i + 1], f.cov3d[6 * i + 3], f.cov3d[6 * i + 4]},
                                 {f.cov3d[6 * i + 2], f.cov3d[6 * i + 4], f.cov3d[6 * i + 5]}};
        const double txtz = t[0] / t[2], tytz = t[1] / t[2];
        const double cx = fmin(limx, fmax(-limx, txtz)) * t[2], cy = fmin(limy, fmax(-limy, tytz)) * t[2];
        const double xmul = (txtz < -limx || txtz > limx) ? 0.0 : 1.0, ymul = (tytz < -limy || tytz > limy) ? 0.0 : 1.0;
        const double j00 = fx / t[2], j02 = -fx * cx / (t[2] * t[2]), j11 = fy / t[2], j12 = -fy * cy / (t[2] * t[2]);
        double T0[3], T1[3];
        for (int k = 0; k < 3; ++k) {
            T0[k] = j00 * vm[4 * k + 0] + j02 * vm[4 * k + 2];
            T1[k] = j11 * vm[4 * k + 1] + j12 * vm[4 * k + 2];
        }
        double a = 0, b = 0, c = 0;
        for (int r = 0; r < 3; ++r)
            for (int s = 0; s < 3; ++s) {
                a += T0[r] * S3[r][s] * T0[s];
                b += T0[r] * S3[r][s] * T1[s];
                c += T1[r] * S3[r][s] * T1[s];
            }
        a += PGR_LOWPASS; c += PGR_LOWPASS;
        const double det = a * c - b * b, d2 = 1.0 / (det * det);
        const double gA = g_conic[3 * i], gB = g_conic[3 * i + 1], gC = g_conic[3 * i + 2];
        const double ga = d2 * (-c * c * gA + b * c * gB - b * b * gC);
        const double gb = d2 * (2 * b * c * gA - (det + 2 * b * b) * gB + 2 * a * b * gC);
        const double gc = d2 * (-b * b * gA + a * b * gB - a * a * gC);

        /* ---- cov2D -> cov3D (6 stored parameters) and T */
        double gS[6];
        gS[0] = ga * T0[0] * T0[0] + gb * T0[0] * T1[0] + gc * T1[0] * T1[0];
        gS[3] = ga * T0[1] * T0[1] + gb * T0[1] * T1[1] + gc * T1[1] * T1[1];
        gS[5] = ga * T0[2] * T0[2] + gb * T0[2] * T1[2] + gc * T1[2] * T1[2];
        gS[1] = 2 * ga * T0[0] * T0[1] + gb * (T0[0] * T1[1] + T0[1] * T1[0]) + 2 * gc * T1[0] * T1[1];
        gS[2] = 2 * ga * T0[0] * T0[2] + gb * (T0[0] * T1[2] + T0[2] * T1[0]) + 2 * gc * T1[0] * T1[2];
        gS[4] = 2 * ga * T0[1] * T0[2] + gb * (T0[1] * T1[2] + T0[2] * T1[1]) + 2 * gc * T1[1] * T1[2];
        if (gr->cov3d) for (int k = 0; k < 6; ++k) gr->cov3d[6 * i + k] = (float)gS[k];
        double gT0[3], gT1[3];
        for (int k = 0; k < 3; ++k) {
            double s0 = 0, s1 = 0;
            for (int s = 0; s < 3; ++s) { s0 += S3[k][s] * T0[s]; s1 += S3[k][s] * T1[s]; }
            gT0[k] = 2 * ga * s0 + gb * s1;
            gT1[k] = 2 * gc * s1 + gb * s0;
        }
        double gj00 = 0, gj02 = 0, gj11 = 0, gj12 = 0;
        for (int k = 0; k < 3; ++k) {
            gj00 += gT0[k] * vm[4 * k + 0]; gj02 += gT0[k] * vm[4 * k + 2];
            gj11 += gT1[k] * vm[4 * k + 1]; gj12 += gT1[k] * vm[4 * k + 2];
        }
        const double tz2 = 1.0 / (t[2] * t[2]), tz3 = tz2 / t[2];
        gt[0] += xmul * -fx * tz2 * gj02;
        gt[1] += ymul * -fy * tz2 * gj12;
        gt[2] += -fx * tz2 * gj00 - fy * tz2 * gj11 + 2 * fx * cx * tz3 * gj02 + 2 * fy * cy * tz3 * gj12;
        /* (a clamped coordinate is c = +-lim * t_z: its own t_z dependence) */
        if (xmul == 0.0) gt[2] += -fx * tz2 * gj02 * (cx / t[2]);
        if (ymul == 0.0) gt[2] += -fy * tz2 * gj12 * (cy / t[2]);
        for (int k = 0; k < 3; ++k) gp[k] += vm[4 * k + 0] * gt[0] + vm[4 * k + 1] * gt[1] + vm[4 * k + 2] * gt[2];

        /* ---- colour */
        if (gr->colors) for (int ch = 0; ch < 3; ++ch) gr->colors[3 * i + ch] = (float)g_rgb[3 * i + ch];
        if (in->shs) {
            double d[3] = {p[0] - in->campos[0], p[1] - in->campos[1], p[2] - in->campos[2]};
            const double len = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            const double u[3] = {d[0] / len, d[1] / len, d[2] / len};
            double bb[16], bx[16], by[16], bz[16];
            sh_basis_grad(in->sh_degree, u[0], u[1], u[2], bb, bx, by, bz);
            const int nc = (in->sh_degree + 1) * (in->sh_degree + 1);
            const float *sh = in->shs + (size_t)i * in->sh_stride * 3;
            double gu[3] = {0, 0, 0};
            for (int ch = 0; ch < 3; ++ch) {
                double acc = 0;
                for (int k = 0; k < nc; ++k) acc += bb[k] * sh[3 * k + ch];
                const double gcol = (acc + 0.5 < 0.0) ? 0.0 : g_rgb[3 * i + ch];      /* clamped at 0: no gradient */
                for (int k = 0; k < nc; ++k) {
                    if (gr->shs) gr->shs[((size_t)i * in->sh_stride + k) * 3 + ch] = (float)(bb[k] * gcol);
                    gu[0] += gcol * sh[3 * k + ch] * bx[k];
                    gu[1] += gcol * sh[3 * k + ch] * by[k];
                    gu[2] += gcol * sh[3 * k + ch] * bz[k];
                }
            }
            /* u = d/|d|:  dL/dd = (gu - u (u.gu)) / |d| */
            const double dot = u[0] * gu[0] + u[1] * gu[1] + u[2] * gu[2];
            for (int k = 0; k < 3; ++k) gp[k] += (gu[k] - u[k] * dot) / len;
        }

        /* ---- cov3D -> scale, rotation */
        if (in->scales && in->rotations && (gr->scales || gr->rotations)) {
            const double q[4] = {in->rotations[4 * i], in->rotations[4 * i + 1], in->rotations[4 * i + 2],
                                 in->rotations[4 * i + 3]};
            const double r = q[0], x = q[1], y = q[2], z = q[3];
            const double R[3][3] = {{1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)},
                                    {2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)},
                                    {2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)}};
            const double s[3] = {in->scale_modifier * (double)in->scales[3 * i], in->scale_modifier * (double)in->scales[3 * i + 1],
                                 in->scale_modifier * (double)in->scales[3 * i + 2]};
            /* Sigma = M M^T, M_ik = R_ik s_k.  Full symmetric gradient: off-diagonals carry half the stored one. */
            const double Gf[3][3] = {{gS[0], 0.5 * gS[1], 0.5 * gS[2]}, {0.5 * gS[1], gS[3], 0.5 * gS[4]},
                                     {0.5 * gS[2], 0.5 * gS[4], gS[5]}};
            double gM[3][3];
            for (int a_ = 0; a_ < 3; ++a_)
                for (int k = 0; k < 3; ++k) {
                    double acc = 0;
                    for (int m = 0; m < 3; ++m) acc += 2 * Gf[a_][m] * R[m][k] * s[k];
                    gM[a_][k] = acc;
                }
            double gR[3][3];
            for (int k = 0; k < 3; ++k) {
                double acc = 0;
                for (int a_ = 0; a_ < 3; ++a_) { acc += gM[a_][k] * R[a_][k]; gR[a_][k] = gM[a_][k] * s[k]; }
                if (gr->scales) gr->scales[3 * i + k] = (float)(acc * in->scale_modifier);
            }
            if (gr->rotations) {
                const double gq_r = 2 * (-z * gR[0][1] + y * gR[0][2] + z * gR[1][0] - x * gR[1][2] - y * gR[2][0] + x * gR[2][1]);
                const double gq_x = 2 * (y * gR[0][1] + z * gR[0][2] + y * gR[1][0] - 2 * x * gR[1][1] - r * gR[1][2] +
                                         z * gR[2][0] + r * gR[2][1] - 2 * x * gR[2][2]);
                const double gq_y = 2 * (-2 * y * gR[0][0] + x * gR[0][1] + r * gR[0][2] + x * gR[1][0] + z * gR[1][2] -
                                         r * gR[2][0] + z * gR[2][1] - 2 * y * gR[2][2]);
                const double gq_z = 2 * (-2 * z * gR[0][0] - r * gR[0][1] + x * gR[0][2] + r * gR[1][0] - 2 * z * gR[1][1] +
                                         y * gR[1][2] + x * gR[2][0] + y * gR[2][1]);
                gr->rotations[4 * i] = (float)gq_r; gr->rotations[4 * i + 1] = (float)gq_x;
                gr->rotations[4 * i + 2] = (float)gq_y; gr->rotations[4 * i + 3] = (float)gq_z;
            }
        }

        if (gr->opacities) gr->opacities[i] = (float)g_op[i];
        if (gr->means3d) for (int k = 0; k < 3; ++k) gr->means3d[3 * i + k] = (float)gp[k];
    }

    free(g_xy); free(g_conic); free(g_op); free(g_rgb); free(g_z);
    free(f.radii); free(f.tiles_touched); free(f.xy); free(f.depth); free(f.conic_opacity); free(f.rgb); free(f.cov3d);
    free(f.keys_sorted); free(f.gauss_sorted); free(f.ranges); free(f.out_color); free(f.out_depth); free(f.final_T);
    free(f.n_contrib);
    return 0;
}
