// compose.hip.h -- scene composition: rigid pose of one object applied to its Gaussians, written straight
// into the merged scene buffers.
//
// Reference behaviour (host round trips + 6 x vstack per frame):
//   xyz       x' = R (x - mean) + mean + t           /root/reference/src/gs/gaussian_model.py:482-497
//   rotation  q' = quat(R * Rot(normalise(q)))       gaussian_model.py:499-505 (GPU -> CPU -> scipy -> GPU)
//   SH        bands 1..3 multiplied by D_l(R)        gaussian_model.py:507-546 (e3nn Wigner-D)
//   merge     6 x torch.vstack                       gaussian_model.py:584-591, pegasus.py:255-264
// Here: one HBM-bound pass, 208 B read + 208 B written per Gaussian, no host involvement.
#pragma once
#include "pgr_common.h"

namespace pgr {

struct ObjectPoseDev {
    float R[9];        // row-major rotation
    float t[3];
    float center[3];   // rotation centre (the object's cloud mean)
    float q[4];        // R as a unit quaternion (w,x,y,z)
    float D1[9], D2[25], D3[49];   // real-SH band rotations, row-major, c' = D c
};

__global__ __launch_bounds__(256) void compose_object_kernel(int n, const float* __restrict__ xyz,
                                                             const float* __restrict__ rot,
                                                             const float* __restrict__ f_rest, int n_rest,
                                                             int in_rest_stride, ObjectPoseDev P,
                                                             float* __restrict__ out_xyz, float* __restrict__ out_rot,
                                                             float* __restrict__ out_rest, int out_rest_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // position: rotate about the cloud mean, then translate
    const float dx = xyz[3 * i + 0] - P.center[0], dy = xyz[3 * i + 1] - P.center[1], dz = xyz[3 * i + 2] - P.center[2];
    out_xyz[3 * i + 0] = fmaf(P.R[2], dz, fmaf(P.R[1], dy, P.R[0] * dx)) + P.center[0] + P.t[0];
    out_xyz[3 * i + 1] = fmaf(P.R[5], dz, fmaf(P.R[4], dy, P.R[3] * dx)) + P.center[1] + P.t[1];
    out_xyz[3 * i + 2] = fmaf(P.R[8], dz, fmaf(P.R[7], dy, P.R[6] * dx)) + P.center[2] + P.t[2];
    // orientation: q' = q_R (x) normalise(q), Hamilton product, (w,x,y,z)
    if (rot && out_rot) {
        const float4 q = reinterpret_cast<const float4*>(rot)[i];
        const float inv = 1.0f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
        const float w = q.x * inv, x = q.y * inv, y = q.z * inv, z = q.w * inv;
        const float a = P.q[0], b = P.q[1], c = P.q[2], d = P.q[3];
        reinterpret_cast<float4*>(out_rot)[i] = make_float4(a * w - b * x - c * y - d * z, a * x + b * w + c * z - d * y,
                                                            a * y - b * z + c * w + d * x, a * z + b * y - c * x + d * w);
    }
    // SH bands 1..3 (coefficient-major, RGB-minor): c'_l = D_l c_l for each colour channel
    if (f_rest && out_rest && n_rest > 0) {
        const float* src = f_rest + (size_t)i * in_rest_stride;
        float* dst = out_rest + (size_t)i * out_rest_stride;
        float c[45];
        const int nf = 3 * n_rest;
#pragma unroll
        for (int k = 0; k < 45; ++k) c[k] = k < nf ? src[k] : 0.0f;
        if (n_rest >= 3) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    float acc = 0.0f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc = fmaf(P.D1[3 * r + k], c[3 * k + ch], acc);
                    dst[3 * r + ch] = acc;
                }
        }
        if (n_rest >= 8) {
#pragma unroll
            for (int r = 0; r < 5; ++r)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    float acc = 0.0f;
#pragma unroll
                    for (int k = 0; k < 5; ++k) acc = fmaf(P.D2[5 * r + k], c[3 * (3 + k) + ch], acc);
                    dst[3 * (3 + r) + ch] = acc;
                }
        }
        if (n_rest >= 15) {
#pragma unroll
            for (int r = 0; r < 7; ++r)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    float acc = 0.0f;
#pragma unroll
                    for (int k = 0; k < 7; ++k) acc = fmaf(P.D3[7 * r + k], c[3 * (8 + k) + ch], acc);
                    dst[3 * (8 + r) + ch] = acc;
                }
        }
    }
}

}  // namespace pgr
