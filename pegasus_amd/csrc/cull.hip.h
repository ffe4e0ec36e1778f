// cull.hip.h -- the "can this splat reach alpha >= 1/255 anywhere in this pixel rectangle?" predicate.
// Used by the binning kernels (tile rectangle: tight lists) and by the compositor (quarter-tile rectangle:
// per-wave skip mask).  Conservative by construction, bit-reproducible (no transcendental), and mirrored
// operation-for-operation by the oracle (pgr_oracle_tile_may_contribute).
#pragma once
#include "pgr_common.h"

namespace pgr {

// ---- tight-list predicate ----------------------------------------------------------------------
// A (Gaussian, tile) instance only matters if alpha = min(0.99, op*exp(power)) >= 1/255 at some pixel of the
// tile.  With q = A dx^2 + 2B dx dy + C dy^2 (power = -q/2) that needs q <= 2 ln(255 op) somewhere.  The
// test bounds q from BELOW over the tile's continuous pixel rectangle (0 if the centre is inside, else the
// smaller of the minima over the two edges facing the centre) and 2 ln(255 op) from ABOVE (exponent + chord of
// log2 on the mantissa + 0.0861), adds a rounding margin, and keeps the instance unless the lower bound
// clears the upper bound.  Dropped instances would be skipped at every pixel, so no pixel's arithmetic
// changes (oracle: pgr_oracle_tile_may_contribute, same operation order, bit-identical decisions; measured
// on the C3 scene: 55-63 % of the 3-sigma-rectangle instances survive).
__device__ __forceinline__ float edge_min_q(float A, float B, float C, float r, float d_fixed, float lo, float hi) {
    // minimise over t in [lo,hi]:  A*d^2 + 2*B*d*t + C*t^2   with r = B / C precomputed per Gaussian
    float t = -(d_fixed * r);
    t = fminf(hi, fmaxf(lo, t));
    return A * d_fixed * d_fixed + 2.0f * B * d_fixed * t + C * t * t;
}

struct CullSplat { float mx, my, A, B, C, rBC, rBA, tau; uint32_t flags; };   // flags: 1 = never, 2 = always

// rBC = B / C and rBA = B / A are computed once per Gaussian and view by the preprocess (IEEE divisions, the same
// expressions as here) and ride in the record's last two floats: the binning walks and every quarter-tile walk of the
// compositor rebuild this record per use.
__device__ __forceinline__ CullSplat make_cull_splat(float2 xy, float4 co, float rBC, float rBA) {
    CullSplat s;
    s.mx = xy.x; s.my = xy.y; s.A = co.x; s.B = co.y; s.C = co.z;
    s.flags = (co.w < ALPHA_MIN ? 1u : 0u)                      // alpha <= op < 1/255 at every pixel, exactly
            | ((!(co.x > 0.0f) || !(co.z > 0.0f)) ? 2u : 0u);   // degenerate conic: no claim
    s.rBC = rBC;
    s.rBA = rBA;
    const float t = 255.0f * co.w;
    const uint32_t bits = __float_as_uint(t);
    const float e = (float)((int)((bits >> 23) & 0xffu) - 127);
    const float m = __uint_as_float((bits & 0x007fffffu) | 0x3f800000u);
    s.tau = 1.3862944f * (e + (m - 1.0f) + 0.0861f);
    return s;
}

__device__ __forceinline__ CullSplat make_cull_splat(float2 xy, float4 co) {
    return make_cull_splat(xy, co, co.y / co.z, co.y / co.x);
}

// rectangle of pixel CENTRES [x0,x1] x [y0,y1] (inclusive)
__device__ __forceinline__ bool rect_may_contribute(const CullSplat& s, float x0, float y0, float x1, float y1) {
    if (s.flags & 1u) return false;
    if (s.flags & 2u) return true;
    if (s.mx >= x0 && s.mx <= x1 && s.my >= y0 && s.my <= y1) return true;
    const float dx0 = x0 - s.mx, dx1 = x1 - s.mx, dy0 = y0 - s.my, dy1 = y1 - s.my;
    // q grows along every ray from the centre: its minimum over the rectangle lies on an edge FACING the centre
    // (an axis whose range contains the centre has no facing edge; the one taken then only adds a larger candidate)
    const float dxn = s.mx < x0 ? dx0 : dx1, dyn = s.my < y0 ? dy0 : dy1;
    const float q = fminf(edge_min_q(s.A, s.B, s.C, s.rBC, dxn, dy0, dy1), edge_min_q(s.C, s.B, s.A, s.rBA, dyn, dx0, dx1));
    const float DX = fmaxf(fabsf(dx0), fabsf(dx1)), DY = fmaxf(fabsf(dy0), fabsf(dy1));
    const float M = s.A * DX * DX + 2.0f * fabsf(s.B) * DX * DY + s.C * DY * DY;
    return !(q > s.tau + 0.00001f * M + 0.01f);       // a NaN anywhere keeps the instance
}

__device__ __forceinline__ bool tile_may_contribute(const CullSplat& s, int tx, int ty, int W, int H) {
    const float x0 = (float)(tx * TILE), y0 = (float)(ty * TILE);
    const float x1 = fminf(x0 + (float)(TILE - 1), (float)(W - 1));
    const float y1 = fminf(y0 + (float)(TILE - 1), (float)(H - 1));
    return rect_may_contribute(s, x0, y0, x1, y1);
}


}  // namespace pgr
