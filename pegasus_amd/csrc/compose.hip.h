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

// ---- pgr_pose_objects (round 6): the pose calls of a whole frame in three launches, poses taken from DEVICE tensors --------
// PEGASUS moves every object between frames with three calls per object (pegasus_setup.py:195-208: apply_transformation_on_xyz,
// apply_rotation_on_splats, apply_rotation_on_sh), each handed a rotation / translation that already lives on the device.  The
// per-call form above needs the pose on the HOST (quaternion, SH band matrices, the cloud's mean): a device round trip per
// call.  Here a frame's calls are JOBS -- one per (object, array) -- whose R / t stay device pointers:
//   pose_reduce_kernel    partial sums of the positions of every job that rotates about its cloud's mean (fixed order: the
//                         mean is the same bits on every run)
//   pose_prepare_kernel   one workgroup per job: mean, quaternion of R (Shepperd's branches, fp64), SH band matrices
//                         D_l = pinv(B_l) B_l(R^T d) over the caller's sample directions (pegasus_amd/sh_rotation.py's
//                         construction, fp64) -> an ObjectPoseDev per job in the workspace
//   pose_apply_kernel     one thread per Gaussian and job: compose_object_kernel's arithmetic on the job's array
constexpr int POSE_JOBS_PER_LAUNCH = 16;
constexpr int POSE_REDUCE_BLOCKS = 32;       // partial sums per job
constexpr int POSE_SH_DIRS = 61;             // sample directions of the SH band matrices (sh_rotation.py)
enum PoseKind { POSE_XYZ = 0, POSE_ROT = 1, POSE_SH = 2 };

struct PoseJobDev {
    const float* src;        // xyz [n,3] | rot [n,4] | f_rest [n,n_rest,3]
    float* dst;
    const float* R;          // device [9] row-major, or NULL = identity
    const float* t;          // device [3], or NULL (POSE_XYZ only)
    int32_t n, kind, n_rest, about_origin, R_row_stride, t_stride;
    uint32_t first_block;    // of pose_apply_kernel
};
struct PoseJobTable { PoseJobDev job[POSE_JOBS_PER_LAUNCH]; int32_t count; };

__global__ __launch_bounds__(256) void pose_reduce_kernel(PoseJobTable T, double* __restrict__ partial) {
    __shared__ double s[3][256];
    const PoseJobDev& j = T.job[blockIdx.y];
    double ax = 0.0, ay = 0.0, az = 0.0;
    if (j.kind == POSE_XYZ && !j.about_origin && j.R) {
        const int per = (j.n + POSE_REDUCE_BLOCKS - 1) / POSE_REDUCE_BLOCKS;
        const int lo = blockIdx.x * per, hi = min(j.n, lo + per);
        for (int i = lo + threadIdx.x; i < hi; i += 256) {
            ax += (double)j.src[3 * i]; ay += (double)j.src[3 * i + 1]; az += (double)j.src[3 * i + 2];
        }
    }
    s[0][threadIdx.x] = ax; s[1][threadIdx.x] = ay; s[2][threadIdx.x] = az;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s[0][threadIdx.x] += s[0][threadIdx.x + w]; s[1][threadIdx.x] += s[1][threadIdx.x + w]; s[2][threadIdx.x] += s[2][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x < 3) partial[((size_t)blockIdx.y * POSE_REDUCE_BLOCKS + blockIdx.x) * 3 + threadIdx.x] = s[threadIdx.x][0];
}

// the rasterizer's real-SH basis, bands 1..3 (preprocess.hip.h sh_basis), in fp64: b[0..14]
__device__ __forceinline__ void sh_bands_f64(double x, double y, double z, double* b) {
    const double xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    b[0] = -(0.4886025119029199 * y); b[1] = 0.4886025119029199 * z; b[2] = -(0.4886025119029199 * x);
    b[3] = 1.0925484305920792 * xy; b[4] = -1.0925484305920792 * yz; b[5] = 0.31539156525252005 * (2.0 * zz - xx - yy);
    b[6] = -1.0925484305920792 * xz; b[7] = 0.5462742152960396 * (xx - yy);
    b[8] = -0.5900435899266435 * y * (3.0 * xx - yy); b[9] = 2.890611442640554 * xy * z;
    b[10] = -0.4570457994644658 * y * (4.0 * zz - xx - yy); b[11] = 0.3731763325901154 * z * (2.0 * zz - 3.0 * xx - 3.0 * yy);
    b[12] = -0.4570457994644658 * x * (4.0 * zz - xx - yy); b[13] = 1.445305721320277 * z * (xx - yy);
    b[14] = -0.5900435899266435 * x * (xx - 3.0 * yy);
}

// sh_dirs [61,3] unit sample directions, sh_pinv [15,61]: rows 0..2 = pinv(B_1), 3..7 = pinv(B_2), 8..14 = pinv(B_3)
__global__ __launch_bounds__(128) void pose_prepare_kernel(PoseJobTable T, const double* __restrict__ partial,
                                                           const double* __restrict__ sh_dirs, const double* __restrict__ sh_pinv,
                                                           ObjectPoseDev* __restrict__ poses) {
    __shared__ double s_brot[POSE_SH_DIRS][15];
    __shared__ double s_R[9];
    const PoseJobDev& j = T.job[blockIdx.x];
    ObjectPoseDev& P = poses[blockIdx.x];
    const int t = threadIdx.x;
    if (t < 9) s_R[t] = j.R ? (double)j.R[(t / 3) * j.R_row_stride + t % 3] : ((t % 4 == 0) ? 1.0 : 0.0);
    __syncthreads();
    if (t < 9) P.R[t] = (float)s_R[t];
    if (t < 3) {
        P.t[t] = j.t ? j.t[t * j.t_stride] : 0.0f;
        double c = 0.0;
        if (j.kind == POSE_XYZ && !j.about_origin && j.R) {
            for (int b = 0; b < POSE_REDUCE_BLOCKS; ++b) c += partial[((size_t)blockIdx.x * POSE_REDUCE_BLOCKS + b) * 3 + t];
            c /= (double)max(j.n, 1);
        }
        P.center[t] = (float)c;
    }
    if (j.kind == POSE_ROT && t == 0) {
        // unit quaternion (w, x, y, z) of R: the largest of trace and diagonal picks the branch
        const double m00 = s_R[0], m11 = s_R[4], m22 = s_R[8], tr = m00 + m11 + m22;
        double w, x, y, z;
        if (tr >= m00 && tr >= m11 && tr >= m22) {
            w = 1.0 + tr; x = s_R[7] - s_R[5]; y = s_R[2] - s_R[6]; z = s_R[3] - s_R[1];
        } else if (m00 >= m11 && m00 >= m22) {
            x = 1.0 - tr + 2.0 * m00; y = s_R[3] + s_R[1]; z = s_R[6] + s_R[2]; w = s_R[7] - s_R[5];
        } else if (m11 >= m22) {
            y = 1.0 - tr + 2.0 * m11; x = s_R[3] + s_R[1]; z = s_R[7] + s_R[5]; w = s_R[2] - s_R[6];
        } else {
            z = 1.0 - tr + 2.0 * m22; x = s_R[6] + s_R[2]; y = s_R[7] + s_R[5]; w = s_R[3] - s_R[1];
        }
        const double inv = 1.0 / sqrt(w * w + x * x + y * y + z * z);
        P.q[0] = (float)(w * inv); P.q[1] = (float)(x * inv); P.q[2] = (float)(y * inv); P.q[3] = (float)(z * inv);
    }
    if (j.kind == POSE_SH) {
        if (t < POSE_SH_DIRS) {
            // rows: Y(R^T d_k), (R^T d)^T = d^T R
            const double dx = sh_dirs[3 * t], dy = sh_dirs[3 * t + 1], dz = sh_dirs[3 * t + 2];
            sh_bands_f64(dx * s_R[0] + dy * s_R[3] + dz * s_R[6], dx * s_R[1] + dy * s_R[4] + dz * s_R[7],
                         dx * s_R[2] + dy * s_R[5] + dz * s_R[8], s_brot[t]);
        }
        __syncthreads();
        if (t < 83) {
            int i, c, off, dim;                       // entry (i, c) of band matrix l; off = the band's first coefficient
            if (t < 9) { i = t / 3; c = t % 3; off = 0; dim = 3; }
            else if (t < 34) { i = (t - 9) / 5; c = (t - 9) % 5; off = 3; dim = 5; }
            else { i = (t - 34) / 7; c = (t - 34) % 7; off = 8; dim = 7; }
            double acc = 0.0;
            for (int k = 0; k < POSE_SH_DIRS; ++k) acc += sh_pinv[(size_t)(off + i) * POSE_SH_DIRS + k] * s_brot[k][off + c];
            float* D = dim == 3 ? P.D1 : (dim == 5 ? P.D2 : P.D3);
            D[i * dim + c] = (float)acc;
        }
    }
}

__global__ __launch_bounds__(256) void pose_apply_kernel(PoseJobTable T, const ObjectPoseDev* __restrict__ poses) {
    int ji = 0;
#pragma unroll
    for (int k = 1; k < POSE_JOBS_PER_LAUNCH; ++k) ji += (k < T.count && blockIdx.x >= T.job[k].first_block) ? 1 : 0;
    const PoseJobDev& j = T.job[ji];
    const ObjectPoseDev& P = poses[ji];
    const int i = (int)(blockIdx.x - j.first_block) * 256 + (int)threadIdx.x;
    if (i >= j.n) return;
    if (j.kind == POSE_XYZ) {
        const float dx = j.src[3 * i + 0] - P.center[0], dy = j.src[3 * i + 1] - P.center[1], dz = j.src[3 * i + 2] - P.center[2];
        j.dst[3 * i + 0] = fmaf(P.R[2], dz, fmaf(P.R[1], dy, P.R[0] * dx)) + P.center[0] + P.t[0];
        j.dst[3 * i + 1] = fmaf(P.R[5], dz, fmaf(P.R[4], dy, P.R[3] * dx)) + P.center[1] + P.t[1];
        j.dst[3 * i + 2] = fmaf(P.R[8], dz, fmaf(P.R[7], dy, P.R[6] * dx)) + P.center[2] + P.t[2];
    } else if (j.kind == POSE_ROT) {
        const float4 q = reinterpret_cast<const float4*>(j.src)[i];
        const float inv = 1.0f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
        const float w = q.x * inv, x = q.y * inv, y = q.z * inv, z = q.w * inv;
        const float a = P.q[0], b = P.q[1], c = P.q[2], d = P.q[3];
        reinterpret_cast<float4*>(j.dst)[i] = make_float4(a * w - b * x - c * y - d * z, a * x + b * w + c * z - d * y,
                                                           a * y - b * z + c * w + d * x, a * z + b * y - c * x + d * w);
    } else {
        const int nf = 3 * j.n_rest;
        const float* src = j.src + (size_t)i * nf;
        float* dst = j.dst + (size_t)i * nf;
        float c[45];
#pragma unroll
        for (int k = 0; k < 45; ++k) c[k] = k < nf ? src[k] : 0.0f;
        if (j.n_rest >= 3) {
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
        if (j.n_rest >= 8) {
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
        if (j.n_rest >= 15) {
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
