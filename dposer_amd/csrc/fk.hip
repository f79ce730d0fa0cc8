// Body-model kernels: rotation conversions and forward kinematics (gfx950).
//
//   rot6d_to_rotmat  -- reference lib/utils/transforms.py:227-235
//   rodrigues        -- smplx==0.1.28 lbs.py batch_rodrigues            (un-vendored dependency)
//   fk_joints        -- smplx lbs.py batch_rigid_transform + the pose assembly of
//                       SMPLX.forward (reference call site lib/body_model/body_model.py:68-88)
//
// These are HBM-bound (FK joints-only: 252 B in + 264 B out and ~3 kFLOP per pose), so the design
// is: one lane = one pose (the whole kinematic chain runs in that lane's registers, the parents
// table is a compile-time constant, no cross-lane traffic at all), and every global access goes
// through an LDS transpose so that HBM only ever sees fully coalesced 16-byte-per-lane streams.
// A lane-per-joint + wave-shuffle chain was rejected: it needs ~12 shuffles per tree level per
// pose and is instruction-bound far below the HBM roofline.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "../../include/dposer_hip.h"
#include "common.h"
#include "gemm_api.h"
#include "kernels_api.h"

#pragma clang fp contract(off)

#define FK_HIP_LAUNCH(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return dposer_set_error(DPOSER_ERR_HIP, std::string(__func__) + ": " + #expr + ": " + hipGetErrorString(_e)); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// LDS-transposed streaming: a block of NT lanes owns NT consecutive items of IN / OUT floats.
// ------------------------------------------------------------------------------------------------
template <int NT> __device__ __forceinline__ void stream_in(const float* __restrict__ g, float* lds, int64_t item0, int64_t n_items, int width) {
    // copy items [item0, item0+NT) x width floats; global side contiguous, LDS row stride = width (+1 if even)
    const int64_t base = item0 * width;
    const int64_t limit = n_items * (int64_t)width;
    const int total = NT * width;
    const int stride = width | 1;
    for (int i = threadIdx.x; i < total; i += NT) {
        const int64_t gi = base + i;
        const float v = gi < limit ? g[gi] : 0.f;
        lds[(i / width) * stride + (i % width)] = v;
    }
}
template <int NT> __device__ __forceinline__ void stream_out(float* __restrict__ g, const float* lds, int64_t item0, int64_t n_items, int width) {
    const int64_t base = item0 * width;
    const int64_t limit = n_items * (int64_t)width;
    const int total = NT * width;
    const int stride = width | 1;
    for (int i = threadIdx.x; i < total; i += NT) {
        const int64_t gi = base + i;
        if (gi < limit) g[gi] = lds[(i / width) * stride + (i % width)];
    }
}

#include "rot_dev.h"

__global__ void __launch_bounds__(256) k_rodrigues(const float* __restrict__ aa, float* __restrict__ out, int64_t n) {
    __shared__ float lds[256 * 9];
    const int64_t item0 = (int64_t)blockIdx.x * 256;
    stream_in<256>(aa, lds, item0, n, 3);
    __syncthreads();
    const float* r = lds + threadIdx.x * 3;
    const Mat3 R = rodrigues(r[0], r[1], r[2]);
    __syncthreads();
    float* o = lds + threadIdx.x * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) o[i] = R.m[i];
    __syncthreads();
    stream_out<256>(out, lds, item0, n, 9);
}

// lib/utils/transforms.py:227-235 (F.normalize eps = 1e-12; R = stack(b1, b2, b3, dim=-1))
__global__ void __launch_bounds__(256) k_rot6d(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
    __shared__ float lds[256 * 9];
    const int64_t item0 = (int64_t)blockIdx.x * 256;
    stream_in<256>(in, lds, item0, n, 6);
    __syncthreads();
    const float* r = lds + threadIdx.x * 7;   // row stride 6|1 = 7
    const float a1x = r[0], a2x = r[1], a1y = r[2], a2y = r[3], a1z = r[4], a2z = r[5];   // view(-1,3,2)
    __syncthreads();
    const float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
    const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
    const float d = b1x * a2x + b1y * a2y + b1z * a2z;
    const float ux = a2x - d * b1x, uy = a2y - d * b1y, uz = a2z - d * b1z;
    const float n2 = fmaxf(sqrtf(ux * ux + uy * uy + uz * uz), 1e-12f);
    const float b2x = ux / n2, b2y = uy / n2, b2z = uz / n2;
    const float b3x = b1y * b2z - b1z * b2y, b3y = b1z * b2x - b1x * b2z, b3z = b1x * b2y - b1y * b2x;
    float* o = lds + threadIdx.x * 9;
    o[0] = b1x; o[1] = b2x; o[2] = b3x;
    o[3] = b1y; o[4] = b2y; o[5] = b3y;
    o[6] = b1z; o[7] = b2z; o[8] = b3z;
    __syncthreads();
    stream_out<256>(out, lds, item0, n, 9);
}

// Rotation matrix -> axis-angle through the unit quaternion: the route torchgeometry.rotation_matrix_to_angle_axis takes
// (rotation_matrix_to_quaternion, quaternion_to_angle_axis), which lib/utils/transforms.py:197-224 calls for rot6d -> axis-angle.
// Quaternion by the four-case rule (pivot on the largest of trace, m00, m11, m22: one square root, the other three components
// from off-diagonal sums / differences, so a near-zero component keeps full relative accuracy); angle-axis with the sign
// convention of quaternion_to_angle_axis (q and -q give the same vector, angle in [0, pi]); NaNs zeroed as transforms.py:223.
// FROM6D: Gram-Schmidt of the 6-D representation first (k_rot6d), one kernel.
__device__ __forceinline__ void rotmat_to_aa(const float (&m)[9], float (&aa)[3]) {
    const float t = m[0] + m[4] + m[8];
    float qw, qx, qy, qz;
    if (t > 0.f) {
        const float S = sqrtf(t + 1.0f) * 2.0f;
        qw = 0.25f * S; qx = (m[7] - m[5]) / S; qy = (m[2] - m[6]) / S; qz = (m[3] - m[1]) / S;
    } else if (m[0] > m[4] && m[0] > m[8]) {
        const float S = sqrtf(1.0f + m[0] - m[4] - m[8]) * 2.0f;
        qw = (m[7] - m[5]) / S; qx = 0.25f * S; qy = (m[1] + m[3]) / S; qz = (m[2] + m[6]) / S;
    } else if (m[4] > m[8]) {
        const float S = sqrtf(1.0f + m[4] - m[0] - m[8]) * 2.0f;
        qw = (m[2] - m[6]) / S; qx = (m[1] + m[3]) / S; qy = 0.25f * S; qz = (m[5] + m[7]) / S;
    } else {
        const float S = sqrtf(1.0f + m[8] - m[0] - m[4]) * 2.0f;
        qw = (m[3] - m[1]) / S; qx = (m[2] + m[6]) / S; qy = (m[5] + m[7]) / S; qz = 0.25f * S;
    }
    const float sin2 = qx * qx + qy * qy + qz * qz;
    const float sn = sqrtf(sin2);
    const float two_theta = 2.0f * (qw < 0.f ? atan2f(-sn, -qw) : atan2f(sn, qw));
    const float k = sin2 > 0.f ? two_theta / sn : 2.0f;
    aa[0] = qx * k; aa[1] = qy * k; aa[2] = qz * k;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        if (aa[i] != aa[i]) aa[i] = 0.f;
}
template <bool FROM6D> __global__ void __launch_bounds__(256) k_to_axis_angle(const float* __restrict__ in, float* __restrict__ out, int64_t n) {
    constexpr int W = FROM6D ? 6 : 9;
    __shared__ float lds[256 * (W | 1)];
    const int64_t item0 = (int64_t)blockIdx.x * 256;
    stream_in<256>(in, lds, item0, n, W);
    __syncthreads();
    const float* r = lds + threadIdx.x * (W | 1);
    float m[9];
    if constexpr (FROM6D) {
        const float a1x = r[0], a2x = r[1], a1y = r[2], a2y = r[3], a1z = r[4], a2z = r[5];
        const float n1 = fmaxf(sqrtf(a1x * a1x + a1y * a1y + a1z * a1z), 1e-12f);
        const float b1x = a1x / n1, b1y = a1y / n1, b1z = a1z / n1;
        const float d = b1x * a2x + b1y * a2y + b1z * a2z;
        const float ux = a2x - d * b1x, uy = a2y - d * b1y, uz = a2z - d * b1z;
        const float n2 = fmaxf(sqrtf(ux * ux + uy * uy + uz * uz), 1e-12f);
        const float b2x = ux / n2, b2y = uy / n2, b2z = uz / n2;
        m[0] = b1x; m[1] = b2x; m[2] = b1y * b2z - b1z * b2y;
        m[3] = b1y; m[4] = b2y; m[5] = b1z * b2x - b1x * b2z;
        m[6] = b1z; m[7] = b2z; m[8] = b1x * b2y - b1y * b2x;
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) m[i] = r[i];
    }
    __syncthreads();
    float aa[3];
    rotmat_to_aa(m, aa);
    float* o = lds + threadIdx.x * 3;
    o[0] = aa[0]; o[1] = aa[1]; o[2] = aa[2];
    __syncthreads();
    stream_out<256>(out, lds, item0, n, 3);
}
extern "C" int dposer_rotmat_to_axis_angle(const float* rotmat, float* axis_angle, int64_t n, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(rotmat && axis_angle && n >= 0, "bad argument");
    if (n == 0) return DPOSER_OK;
    hipLaunchKernelGGL(k_to_axis_angle<false>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, rotmat, axis_angle, n);
    FK_HIP_LAUNCH(hipGetLastError());
    return DPOSER_OK;
}
extern "C" int dposer_rot6d_to_axis_angle(const float* rot6d, float* axis_angle, int64_t n, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(rot6d && axis_angle && n >= 0, "bad argument");
    if (n == 0) return DPOSER_OK;
    hipLaunchKernelGGL(k_to_axis_angle<true>, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, rot6d, axis_angle, n);
    FK_HIP_LAUNCH(hipGetLastError());
    return DPOSER_OK;
}

extern "C" int dposer_rodrigues(const float* aa, float* rotmat, int64_t n, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(aa && rotmat && n >= 0, "bad argument");
    if (n == 0) return DPOSER_OK;
    hipLaunchKernelGGL(k_rodrigues, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, aa, rotmat, n);
    FK_HIP_LAUNCH(hipGetLastError());
    return DPOSER_OK;
}
extern "C" int dposer_rot6d_to_rotmat(const float* rot6d, float* rotmat, int64_t n, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(rot6d && rotmat && n >= 0, "bad argument");
    if (n == 0) return DPOSER_OK;
    hipLaunchKernelGGL(k_rot6d, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, rot6d, rotmat, n);
    FK_HIP_LAUNCH(hipGetLastError());
    return DPOSER_OK;
}

// ------------------------------------------------------------------------------------------------
// Rest shape: blend shapes + joint regression (smplx lbs.py blend_shapes / vertices2joints; include/dposer_hip.h)
// ------------------------------------------------------------------------------------------------
// HBM-bound on its output (B x V*3 floats): one thread owns one vertex coordinate i, keeps its L blend-shape coefficients in
// registers (one 4*L-byte contiguous read, L2-resident across the pose chunks) and walks a chunk of poses; the shape
// coefficients are wave-uniform, i.e. scalar loads.  The joint part (J*3 coordinates) runs as extra tiles of the same grid
// on the pre-regressed directions.
constexpr int SB_L = 32;         // shape coefficients per register pass (smplx: 10 + 10 = one pass; full spaces of 300 + 100 loop)
constexpr int SB_POSES = 8;      // poses per block
__global__ void __launch_bounds__(256) k_shape_blend(const float* __restrict__ vt, const float* __restrict__ sd, const float* __restrict__ jt,
                                                      const float* __restrict__ jd, const float* __restrict__ shape, float* __restrict__ vs,
                                                      float* __restrict__ jr, int V3, int J3, int L, int64_t B, int v_tiles) {
    const bool joints = (int)blockIdx.x >= v_tiles;
    const int n = joints ? J3 : V3;
    const int i = (joints ? blockIdx.x - v_tiles : blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const float* dir = (joints ? jd : sd) + (int64_t)i * L;
    const float base = (joints ? jt : vt)[i];
    float* out = joints ? jr : vs;
    const int64_t b0 = (int64_t)blockIdx.y * SB_POSES;
    float acc[SB_POSES];
#pragma unroll
    for (int k = 0; k < SB_POSES; ++k) acc[k] = 0.f;
    for (int l0 = 0; l0 < L; l0 += SB_L) {
        float c[SB_L];
#pragma unroll
        for (int l = 0; l < SB_L; ++l) c[l] = l0 + l < L ? dir[l0 + l] : 0.f;
#pragma unroll
        for (int k = 0; k < SB_POSES; ++k) {
            if (b0 + k >= B) break;
            const float* sh = shape + (b0 + k) * L + l0;       // sum_l shape[l] * dir[l] in index order (torch.einsum's reduction
#pragma unroll                                                 // order is not part of any contract; fp32 agreement is ~1e-7)
            for (int l = 0; l < SB_L; ++l)
                if (l0 + l < L) acc[k] += sh[l] * c[l];
        }
    }
#pragma unroll
    for (int k = 0; k < SB_POSES; ++k)
        if (b0 + k < B) out[(b0 + k) * n + i] = base + acc[k];
}
// d_shape[b][l] = sum_i dv[b][i] * sd[i][l]: stage 1 = per (pose, coordinate chunk, block of 32 coefficients) partial sums (fixed
// order => deterministic)
constexpr int SBB_CHUNK = 4096;
__global__ void __launch_bounds__(256) k_shape_blend_bwd_part(const float* __restrict__ sd, const float* __restrict__ dv, float* __restrict__ part,
                                                               int V3, int L, int n_chunks) {
    __shared__ float red[4][SB_L];
    const int64_t b = blockIdx.y;
    const int i0 = blockIdx.x * SBB_CHUNK;
    const int l0 = blockIdx.z * SB_L;
    float acc[SB_L];
#pragma unroll
    for (int l = 0; l < SB_L; ++l) acc[l] = 0.f;
    for (int i = i0 + threadIdx.x; i < i0 + SBB_CHUNK && i < V3; i += 256) {
        const float g = dv[b * V3 + i];
        const float* d = sd + (int64_t)i * L + l0;
#pragma unroll
        for (int l = 0; l < SB_L; ++l)
            if (l0 + l < L) acc[l] += g * d[l];
    }
#pragma unroll
    for (int l = 0; l < SB_L; ++l) {
        float v = acc[l];
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][l] = v;
    }
    __syncthreads();
    if (threadIdx.x < SB_L && l0 + threadIdx.x < L)
        part[(b * n_chunks + blockIdx.x) * L + l0 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// stage 2: chunks in order, then the joint term
__global__ void k_shape_blend_bwd_final(const float* __restrict__ part, const float* __restrict__ jd, const float* __restrict__ dj, float* __restrict__ d_shape,
                                        int J3, int L, int n_chunks, int64_t B) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * L) return;
    const int64_t b = t / L;
    const int l = (int)(t % L);
    float acc = 0.f;
    for (int c = 0; c < n_chunks; ++c) acc += part[(b * n_chunks + c) * L + l];
    if (dj)
        for (int i = 0; i < J3; ++i) acc += dj[b * J3 + i] * jd[(int64_t)i * L + l];
    d_shape[t] = acc;
}
extern "C" int dposer_shape_blend_forward(const float* v_template, const float* shapedirs, const float* j_template, const float* jdirs,
                                          const float* shape, float* v_shaped, float* j_rest, int32_t num_vertices, int32_t num_joints,
                                          int32_t num_shape, int64_t batch, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(v_template && shapedirs && j_template && jdirs && shape && v_shaped && j_rest, "null argument");
    DP_CHECK_ARG(num_shape >= 1 && num_shape <= 1024, "num_shape must be in 1..1024");
    DP_CHECK_ARG(num_vertices > 0 && num_joints > 0 && batch >= 0, "bad size");
    if (batch == 0) return DPOSER_OK;
    const int V3 = num_vertices * 3, J3 = num_joints * 3;
    const int v_tiles = (int)ceil_div(V3, 256), j_tiles = (int)ceil_div(J3, 256);
    DP_CHECK_ARG(ceil_div(batch, SB_POSES) <= 65535, "batch too large for one launch");
    hipLaunchKernelGGL(k_shape_blend, dim3(v_tiles + j_tiles, (unsigned)ceil_div(batch, SB_POSES)), dim3(256), 0, (hipStream_t)stream, v_template,
                       shapedirs, j_template, jdirs, shape, v_shaped, j_rest, V3, J3, num_shape, batch, v_tiles);
    FK_HIP_LAUNCH(hipGetLastError());
    return DPOSER_OK;
}
extern "C" int64_t dposer_shape_blend_scratch_floats(int32_t num_vertices, int32_t num_shape, int64_t batch) {
    return batch * ceil_div((int64_t)num_vertices * 3, SBB_CHUNK) * num_shape;
}
extern "C" int dposer_shape_blend_backward(const float* shapedirs, const float* jdirs, const float* d_v_shaped, const float* d_j_rest,
                                           float* d_shape, float* scratch, int32_t num_vertices, int32_t num_joints, int32_t num_shape,
                                           int64_t batch, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(shapedirs && jdirs && d_v_shaped && d_shape && scratch, "null argument");
    DP_CHECK_ARG(num_shape >= 1 && num_shape <= 1024, "num_shape must be in 1..1024");
    if (batch == 0) return DPOSER_OK;
    DP_CHECK_ARG(batch <= 65535, "batch too large for one launch");
    const int V3 = num_vertices * 3, J3 = num_joints * 3;
    const int n_chunks = (int)ceil_div(V3, SBB_CHUNK);
    hipLaunchKernelGGL(k_shape_blend_bwd_part, dim3(n_chunks, (unsigned)batch, (unsigned)ceil_div(num_shape, SB_L)), dim3(256), 0, (hipStream_t)stream, shapedirs, d_v_shaped, scratch, V3,
                       num_shape, n_chunks);
    FK_HIP_LAUNCH(hipGetLastError());
    hipLaunchKernelGGL(k_shape_blend_bwd_final, dim3((unsigned)ceil_div(batch * num_shape, 128)), dim3(128), 0, (hipStream_t)stream, scratch, jdirs,
                       d_j_rest, d_shape, J3, num_shape, n_chunks, batch);
    FK_HIP_LAUNCH(hipGetLastError());
    return DPOSER_OK;
}

// ------------------------------------------------------------------------------------------------
// Forward kinematics
// ------------------------------------------------------------------------------------------------
struct KinSMPL {
    static constexpr int J = 24;
    static constexpr int P[24] = {-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21};
};
struct KinSMPLH {
    static constexpr int J = 52;
    static constexpr int P[52] = {-1, 0,  0,  0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  9,  9,  12, 13, 14, 16, 17, 18, 19, 20, 22, 23, 20,
                                  25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 21, 37, 38, 21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50};
};
struct KinSMPLX {
    static constexpr int J = 55;
    static constexpr int P[55] = {-1, 0,  0,  0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  9,  9,  12, 13, 14, 16, 17, 18, 19, 15, 15, 15, 20, 25, 26,
                                  20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38, 21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53};
};

struct Xf {        // 3x4 rigid transform
    float r[9];
    float t[3];
};

constexpr int FK_MAX_SEG = 8;
struct FkArgs {
    const float* seg[FK_MAX_SEG];   // pose segments [B][seg_joints*3]; null => zeros (identity rotations)
    int seg_first[FK_MAX_SEG];      // first joint of each segment
    int seg_joints[FK_MAX_SEG];
    int nseg;
    const float* j_rest;            // [J][3] or [B][J][3]
    int j_rest_batched;
    const float* transl;            // [B][3] or null
    float* joints;                  // [B][joints_ld] rows, first n_out*3 floats written
    int64_t joints_ld;              // floats per pose row in `joints` (>= n_out*3)
    float* rel;                     // [B][n_out][12] or null
    float* pf;                      // pose feature (R_i - I, i >= 1) as FT32 [Bpad][pf_K] or null   (smplx lbs.py: pose_feature)
    int pf_K;
    int n_out;                      // joints [0, n_out) are computed and written
    int64_t B;
};

// does joint I have a child among the joints [0, NOUT)?  (compile-time walk of the parents table)
template <typename Kin> constexpr bool fk_has_child_below(int I, int NOUT) {
    for (int c = I + 1; c < NOUT && c < Kin::J; ++c)
        if (Kin::P[c] == I) return true;
    return false;
}

// One joint of the chain, with the joint index (and therefore its parent) a compile-time constant
// so that the per-lane transforms G[] stay in registers.
// NOUT > 0: "joints only" specialisation (n_out == NOUT, no pose feature, no skinning transforms): everything optional is
// compiled out and a joint without a child below NOUT needs no rotation at all -- its position is parent * rest offset
// (5 of the 22 body joints of SMPL-X: feet, head, wrists).
template <typename Kin, int I, int NOUT>
__device__ __forceinline__ void fk_step(Xf (&G)[Kin::J], const float* pose, const float* jr, float* row, const float (&tr)[3],
                                        const FkArgs& a, int64_t b, int n_out) {
    constexpr bool LEAN = NOUT > 0;
    if (LEAN ? I >= NOUT : I >= n_out) return;
    constexpr int PI = Kin::P[I] < 0 ? 0 : Kin::P[I];
    constexpr bool NEED_R = !LEAN || fk_has_child_below<Kin>(I, NOUT);
    Mat3 R;
    if constexpr (NEED_R) R = rodrigues(pose[3 * I], pose[3 * I + 1], pose[3 * I + 2]);
    float rel[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) rel[k] = jr[3 * I + k] - (I > 0 ? jr[3 * PI + k] : 0.f);
    if (I == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) G[0].r[k] = R.m[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) G[0].t[k] = rel[k];
    } else {
        const Xf Pm = G[PI];
        // G_i = G_parent @ [R rel; 0 1]
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if constexpr (NEED_R) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    G[I].r[3 * r + c] = fmaf(Pm.r[3 * r], R.m[c], fmaf(Pm.r[3 * r + 1], R.m[3 + c], Pm.r[3 * r + 2] * R.m[6 + c]));
            }
            G[I].t[r] = fmaf(Pm.r[3 * r], rel[0], fmaf(Pm.r[3 * r + 1], rel[1], fmaf(Pm.r[3 * r + 2], rel[2], Pm.t[r])));
        }
    }
    // posed joint (+ transl) overwrites the (already consumed) axis-angle of joint I in the lane's LDS row
#pragma unroll
    for (int k = 0; k < 3; ++k) row[3 * I + k] = G[I].t[k] + tr[k];
    if constexpr (LEAN) return;
    if (a.pf && I > 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k)
            a.pf[FT<float>::index(b, (I - 1) * 9 + k, a.pf_K)] = (b < a.B) ? R.m[k] - ((k % 4 == 0) ? 1.0f : 0.0f) : 0.f;
    }
    // skinning transform A_I = [R_I | t_I - R_I J_I] (rest pose removed), 3 x float4 per joint
    if (a.rel && b < a.B) {
        const float jx = jr[3 * I], jy = jr[3 * I + 1], jz = jr[3 * I + 2];
        f32x4* q = reinterpret_cast<f32x4*>(a.rel + (b * n_out + I) * 12);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            f32x4 v;
            v[0] = G[I].r[3 * r]; v[1] = G[I].r[3 * r + 1]; v[2] = G[I].r[3 * r + 2];
            v[3] = G[I].t[r] - (G[I].r[3 * r] * jx + G[I].r[3 * r + 1] * jy + G[I].r[3 * r + 2] * jz);
            q[r] = v;
        }
    }
}
template <typename Kin, int NOUT, int... Is>
__device__ __forceinline__ void fk_chain(std::integer_sequence<int, Is...>, Xf (&G)[Kin::J], const float* pose, const float* jr, float* row,
                                         const float (&tr)[3], const FkArgs& a, int64_t b, int n_out) {
    (fk_step<Kin, Is, NOUT>(G, pose, jr, row, tr, a, b, n_out), ...);
}

// One lane = one pose.  NT lanes per block; pose rows and outputs are staged through LDS.
// LDS row of a pose = 3*n_out floats (+1 pad to make the stride odd => conflict-free per-lane rows): only the
// joints that are actually evaluated are staged, so a 22-joint body query needs 17 KiB per 64 poses, not 42.
template <typename Kin, int NT, int NOUT = 0> __global__ void __launch_bounds__(NT) k_fk_joints(FkArgs a) {
    constexpr int J = Kin::J;
    extern __shared__ float lds[];
    const int64_t item0 = (int64_t)blockIdx.x * NT;
    const int n_out = a.n_out;
    const int ROW = (n_out * 3) | 1;
    for (int sg = 0; sg < FK_MAX_SEG; ++sg) {
        if (sg >= a.nseg) break;
        const int first = a.seg_first[sg];
        if (first >= n_out) break;
        const int gw = a.seg_joints[sg] * 3;                                    // floats per pose in global memory
        const int width = (first + a.seg_joints[sg] <= n_out ? a.seg_joints[sg] : n_out - first) * 3;   // floats staged
        const int col0 = first * 3;
        const float* g = a.seg[sg];
        if (!g) {                                                               // absent segment = zeros (identity rotations)
            for (int e = threadIdx.x; e < NT * width; e += NT) lds[(e / width) * ROW + col0 + (e % width)] = 0.f;
            continue;
        }
        const bool full = (item0 + NT <= a.B) && (width == gw);
        if (full) {
            // the block's NT poses are one contiguous, 16-byte aligned span of NT*gw floats: batches of 8 float4 loads
            // in flight per lane, then scattered into the per-pose LDS rows
            const f32x4* g4 = reinterpret_cast<const f32x4*>(g + item0 * gw);
            const int n4 = NT * gw / 4;
            // element e = 4*i4 lives at (row, col) = (e / gw, e % gw); consecutive batches advance e by 4*NT:
            // (row, col) += ((4*NT) / gw, (4*NT) % gw) with one carry -- no division inside the loop
            const int drow = (4 * NT) / gw, dcol = (4 * NT) % gw;
            int rw = (4 * threadIdx.x) / gw, col = (4 * threadIdx.x) % gw;
            for (int base = 0; base < n4; base += NT * 8) {
                f32x4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i4 = base + k * NT + threadIdx.x;
                    if (i4 < n4) v[k] = g4[i4];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i4 = base + k * NT + threadIdx.x;
                    if (i4 < n4) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool wrap = col + r >= gw;
                            lds[(rw + (wrap ? 1 : 0)) * ROW + col0 + col + r - (wrap ? gw : 0)] = v[k][r];
                        }
                    }
                    rw += drow; col += dcol;
                    if (col >= gw) { col -= gw; ++rw; }
                }
            }
        } else {
            int rw = threadIdx.x / width, col = threadIdx.x % width;
            while (rw < NT) {
                const int64_t bb = item0 + rw;
                lds[rw * ROW + col0 + col] = (bb < a.B) ? g[bb * gw + col] : 0.f;
                col += NT;
                while (col >= width) { col -= width; ++rw; }
            }
        }
    }
    __syncthreads();
    const int64_t b = item0 + threadIdx.x;
    float* row = lds + threadIdx.x * ROW;   // this lane's private LDS row: pose in, posed joints out (in place)
    const float* jr = a.j_rest_batched ? a.j_rest + (b < a.B ? b : 0) * (int64_t)J * 3 : a.j_rest;
    float tr[3] = {0.f, 0.f, 0.f};
    if (a.transl && b < a.B) { tr[0] = a.transl[b * 3]; tr[1] = a.transl[b * 3 + 1]; tr[2] = a.transl[b * 3 + 2]; }
    Xf G[J];
    fk_chain<Kin, NOUT>(std::make_integer_sequence<int, J>{}, G, row, jr, row, tr, a, b, n_out);
    __syncthreads();
    // ---- posed joints: LDS rows (stride ROW) -> coalesced global stream ----
    {
        const int width = n_out * 3;
        if (item0 + NT <= a.B && a.joints_ld == width) {
            f32x4* o4 = reinterpret_cast<f32x4*>(a.joints + item0 * width);
            const int n4 = NT * width / 4;
            const int drow = (4 * NT) / width, dcol = (4 * NT) % width;
            int rw = (4 * threadIdx.x) / width, col = (4 * threadIdx.x) % width;
            for (int i4 = threadIdx.x; i4 < n4; i4 += NT) {
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool wrap = col + r >= width;
                    v[r] = lds[(rw + (wrap ? 1 : 0)) * ROW + col + r - (wrap ? width : 0)];
                }
                o4[i4] = v;
                rw += drow; col += dcol;
                if (col >= width) { col -= width; ++rw; }
            }
        } else {
            int r = threadIdx.x / width, col = threadIdx.x % width;
            while (r < NT) {
                const int64_t bb = item0 + r;
                if (bb < a.B) a.joints[bb * a.joints_ld + col] = lds[r * ROW + col];
                col += NT;
                while (col >= width) { col -= width; ++r; }
            }
        }
    }
}

// Joints-only body query (22 joints, segments = [root | 21 body joints]) on FULL blocks of 64 poses, round 3: the pose rows enter LDS by
// DMA (global_load_lds: the LDS image IS the global layout -- [64][63] body + [64][3] root, both with an odd row stride, i.e.
// conflict-free per-lane rows -- 15 x 1 KiB + 6 x 256 B pieces, no index arithmetic, no VGPR round trip), every lane pulls its 66
// inputs into registers, and the posed joints go back into the SAME 16.9 KB as the [64][66] image of the output, which leaves as 16.5
// coalesced 16-byte stores per lane.  k_fk_joints spends about a third of its ~3000 instructions per wave on scattering float4 loads
// into padded rows and gathering them back (integer division by the row width per element); this kernel issues ~1750.
template <typename Kin> __global__ void __launch_bounds__(64) k_fk_joints_dma(FkArgs a) {
    constexpr int J = Kin::J, NOUT = 22;
    extern __shared__ float lds[];                                  // 64 * 66 floats
    const int lane = threadIdx.x;
    const int64_t item0 = (int64_t)blockIdx.x * 64;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;
    const unsigned char* body = reinterpret_cast<const unsigned char*>(a.seg[1] + item0 * 63);
    const unsigned char* root = reinterpret_cast<const unsigned char*>(a.seg[0] + item0 * 3);
#pragma unroll
    for (int i = 0; i < 15; ++i) glds_asm_b128(body + i * 1024 + lane * 16, lds0 + i * 1024);
#pragma unroll
    for (int i = 0; i < 3; ++i) glds_asm_b32(body + 15360 + i * 256 + lane * 4, lds0 + 15360 + i * 256);
    if (a.seg[0]) {                                                 // (absent root segment = identity rotation)
#pragma unroll
        for (int i = 0; i < 3; ++i) glds_asm_b32(root + i * 256 + lane * 4, lds0 + 16128 + i * 256);
    }
    const int64_t b = item0 + lane;
    const float* jr = a.j_rest_batched ? a.j_rest + b * (int64_t)J * 3 : a.j_rest;
    float tr[3] = {0.f, 0.f, 0.f};
    if (a.transl) { tr[0] = a.transl[b * 3]; tr[1] = a.transl[b * 3 + 1]; tr[2] = a.transl[b * 3 + 2]; }
    __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
    asm volatile("" ::: "memory");
    float pr[NOUT * 3];
#pragma unroll
    for (int k = 0; k < 3; ++k) pr[k] = a.seg[0] ? lds[4032 + lane * 3 + k] : 0.f;
#pragma unroll
    for (int k = 0; k < 63; ++k) pr[3 + k] = lds[lane * 63 + k];
    __syncthreads();                                                // every input is in registers before the first output lands on it
    Xf G[J];
    fk_chain<Kin, NOUT>(std::make_integer_sequence<int, J>{}, G, pr, jr, lds + lane * (NOUT * 3), tr, a, b, NOUT);
    __syncthreads();
    const f32x4* l4 = reinterpret_cast<const f32x4*>(lds);
    f32x4* o4 = reinterpret_cast<f32x4*>(a.joints + item0 * (NOUT * 3));
#pragma unroll
    for (int i = 0; i < 16; ++i) o4[lane + 64 * i] = l4[lane + 64 * i];
    if (lane < 32) o4[1024 + lane] = l4[1024 + lane];
}

// ---- small batches: one wave per pose, one lane per joint -------------------------------------------------------------------
// k_fk_joints gives a lane a whole pose: its ~55 Rodrigues + matrix products are a serial chain, 80-100 us however few poses
// there are (a 60-frame motion-denoising step spent 190 of its 490 us in the forward and backward chains of 60 poses).  Below
// `fk_small_max()` poses the chain is walked by tree depth instead: lane I owns joint I (J <= 64), every lane evaluates its
// Rodrigues formula at once, and level d of the tree (<= 10 for SMPL-X) reads its parents' transforms from LDS.  Per-joint
// arithmetic is the same expressions in the same order as fk_step => bit-identical outputs (tests/test_gpu_fk.py).
template <typename Kin> constexpr int fk_max_depth() {
    int m = 0;
    for (int i = 0; i < Kin::J; ++i) {
        int d = 0;
        for (int q = i; q > 0; q = Kin::P[q]) ++d;
        m = d > m ? d : m;
    }
    return m;
}
template <typename Kin> constexpr int fk_max_children() {
    int m = 0;
    for (int i = 0; i < Kin::J; ++i) {
        int n = 0;
        for (int c = 0; c < Kin::J; ++c)
            if (Kin::P[c] == i) ++n;
        m = n > m ? n : m;
    }
    return m;
}
template <typename Kin> struct KinTable {          // the parents table where device code can index it with a lane id
    static constexpr int MAXC = 6;
    int p[Kin::J];
    // per joint: depth in the tree, number of children, the children in DESCENDING index order (the one-lane-per-joint kernels used to find
    // these by walking the parents table per lane: a chain of dependent loads up to the tree's depth + a 54-iteration search at every launch)
    int depth[Kin::J], nchild[Kin::J], child[Kin::J][MAXC];
    constexpr KinTable() : p(), depth(), nchild(), child() {
        for (int i = 0; i < Kin::J; ++i) p[i] = Kin::P[i] < 0 ? 0 : Kin::P[i];
        for (int i = 0; i < Kin::J; ++i) {
            int d = 0;
            for (int q = i; q > 0; q = p[q]) ++d;
            depth[i] = d;
            int n = 0;
            for (int c = Kin::J - 1; c > i; --c)
                if (p[c] == i) { if (n < MAXC) child[i][n] = c; ++n; }
            nchild[i] = n;
        }
    }
};
template <typename Kin> __device__ constexpr KinTable<Kin> kKinTable{};

// A/B switches of the body-model kernels: read from the environment ONCE (first use), not per call; tests and tuners that change
// them inside one process call dposer_body_tuning_reload() afterwards.
struct BodyTuning {
    int64_t fk_small_max = 8192;          // DPOSER_FK_SMALL_MAX: up to this many poses FK runs one wave per pose / one lane per joint
    int64_t joint_stream_min = 320;       // DPOSER_LBS_JOINT_STREAM_MIN: from this batch the one-pass streaming skinning backward is used (round 5: with k_skin_bwd_mfma it wins from 512 poses, ties at 256)
    bool blend_fp32 = false;              // DPOSER_LBS_BLEND=fp32: exact-fp32 pose-blend chain
    int skin_mode = 3;                    // DPOSER_SKIN_WAVE=0: one vertex per thread and iteration (k_skin); 2: four in flight, one pose per block (k_skin_x4); 3: runs of poses (k_skin_run)
    bool skin_bwd_fused = true;           // DPOSER_SKIN_BWD_FUSED=0: k_skin_bwd + k_skin_bwd_joints instead of the one-pass kernel (A/B)
    bool lbs_fwd_big = true;              // DPOSER_LBS_FWD_BIG=0: 128x128 tiles for the pose-blend GEMM of the forward at every batch size (A/B)
    bool lbs_bwd_big = true;              // DPOSER_LBS_BWD_BIG=0: 128x128 tiles for the blend-gradient GEMMs at every batch size (A/B)
    bool fk_dma = true;                   // DPOSER_FK_DMA=0: joints-only body query through k_fk_joints instead of k_fk_joints_dma (A/B)
    int lbs_bwd_panel_order = 1;          // DPOSER_LBS_BWD_PANEL_ORDER=0: generic block -> tile order for the blend-gradient GEMMs (A/B)
    bool lbs_k_prefix = true;             // DPOSER_LBS_K_PREFIX=0: blend GEMMs over all padded pose-feature columns, posed or not (A/B)
    int skin_bwd_mfma = 1;                // DPOSER_SKIN_BWD_MFMA=0: the LDS-walking one-pass kernel (k_skin_bwd_fused) instead of the one whose joint reduction runs
                                          // on the matrix pipe (k_skin_bwd_mfma); 2 / 4 / 5: poses per workgroup of the latter (default 4; the temporal form 5 from 5120 frames)
    bool lbs_bwd_terms_parallel = true;   // DPOSER_LBS_BWD_TERMS_PARALLEL=0: the three product terms of the bf16 x 3 blend-gradient GEMM one after the other on the
                                          // caller's stream (A/B) instead of side by side on three streams
    bool lbs_bwd_rowcat = true;           // DPOSER_LBS_BWD_ROWCAT=0: the two blend-gradient product terms that read the high plane of d_offsets as two launches
                                          // (A/B) instead of ONE launch against the row-concatenated [posedirs high ; posedirs low] (round 6: the 258 MB plane is read once)
    int lbs_bwd_ksplit = 0;               // DPOSER_LBS_BWD_KSPLIT=n: force the split count of the 256x256 blend-gradient GEMMs (A/B; 0 = chosen by lbs_bwd_big_ksplit)
    int64_t fk_lds_pad = 0;               // DPOSER_FK_LDS_PAD=bytes: extra (unused) dynamic LDS per workgroup of k_fk_joints_dma -- an occupancy probe (fewer resident
                                          // waves per CU, same kernel): tools/fk_occupancy_sweep.sh, profiles/r06_fk_occupancy.md
    bool lbs_fwd_chunk_serial = false;    // DPOSER_LBS_FWD_CHUNK_SERIAL=1: the chunks of DPOSER_LBS_FWD_CHUNK one after the other on the caller's stream (blend GEMM of chunk i,
                                          // its skinning, chunk i + 1 ...): no overlap, but a chunk's offsets may still sit in the memory-side cache when they are read
    int64_t lbs_fwd_chunk = 0;            // DPOSER_LBS_FWD_CHUNK=n (multiple of 256): the full forward runs blend GEMM and skinning in chunks of n poses, the
                                          // skinning of chunk i on a side stream beside the GEMM of chunk i + 1 (0: one launch each over the whole batch)
    void load() {
        const char* e = getenv("DPOSER_FK_SMALL_MAX");
        fk_small_max = e ? atoll(e) : (int64_t)8192;
        e = getenv("DPOSER_LBS_JOINT_STREAM_MIN");
        joint_stream_min = e ? atoll(e) : (int64_t)320;
        e = getenv("DPOSER_LBS_BLEND");
        blend_fp32 = e && e[0] == 'f';
        e = getenv("DPOSER_SKIN_WAVE");
        skin_mode = e ? atoi(e) : 3;
        e = getenv("DPOSER_SKIN_BWD_FUSED");
        skin_bwd_fused = !(e && e[0] == '0');
        e = getenv("DPOSER_LBS_BWD_BIG");
        lbs_bwd_big = !(e && e[0] == '0');
        e = getenv("DPOSER_LBS_FWD_BIG");
        lbs_fwd_big = !(e && e[0] == '0');
        e = getenv("DPOSER_FK_DMA");
        fk_dma = !(e && e[0] == '0');
        e = getenv("DPOSER_LBS_K_PREFIX");
        lbs_k_prefix = !(e && e[0] == '0');
        e = getenv("DPOSER_LBS_BWD_PANEL_ORDER");
        lbs_bwd_panel_order = (e && e[0] == '0') ? 0 : 1;
        e = getenv("DPOSER_SKIN_BWD_MFMA");
        skin_bwd_mfma = e ? atoi(e) : 1;
        e = getenv("DPOSER_LBS_BWD_TERMS_PARALLEL");
        lbs_bwd_terms_parallel = !(e && e[0] == '0');
        e = getenv("DPOSER_LBS_BWD_ROWCAT");
        lbs_bwd_rowcat = !(e && e[0] == '0');
        e = getenv("DPOSER_LBS_BWD_KSPLIT");
        lbs_bwd_ksplit = e ? atoi(e) : 0;
        e = getenv("DPOSER_FK_LDS_PAD");
        fk_lds_pad = e ? atoll(e) : (int64_t)0;
        e = getenv("DPOSER_LBS_FWD_CHUNK_SERIAL");
        lbs_fwd_chunk_serial = e && e[0] == '1';
        e = getenv("DPOSER_LBS_FWD_CHUNK");
        lbs_fwd_chunk = e ? atoll(e) / 256 * 256 : (int64_t)0;
    }
};
static BodyTuning& body_tuning() {
    static BodyTuning t = [] { BodyTuning x; x.load(); return x; }();
    return t;
}
extern "C" void dposer_body_tuning_reload(void) { body_tuning().load(); }
static int64_t fk_small_max() { return body_tuning().fk_small_max; }

template <typename Kin> __global__ void __launch_bounds__(64) k_fk_small(FkArgs a) {
    constexpr int J = Kin::J;
    constexpr int MAXD = fk_max_depth<Kin>();
    __shared__ float sG[J][12];
    const int64_t b = blockIdx.x;
    const int I = threadIdx.x;
    const int n_out = a.n_out;
    const bool on = I < n_out;
    int P = 0, depth = 0;
    if (on) {
        P = kKinTable<Kin>.p[I];
        depth = kKinTable<Kin>.depth[I];
    }
    // this joint's axis-angle: segment pointers through unrolled selects (a runtime index into the kernel-argument arrays would
    // spill the argument struct to scratch)
    int li = I;
    const float* pose = a.seg[0];
    int seg_nj = a.seg_joints[0];
#pragma unroll
    for (int k = 1; k < FK_MAX_SEG; ++k)
        if (k < a.nseg && I >= a.seg_first[k]) { li = I - a.seg_first[k]; pose = a.seg[k]; seg_nj = a.seg_joints[k]; }
    float rx = 0.f, ry = 0.f, rz = 0.f;
    if (on && pose) { const float* q = pose + (b * seg_nj + li) * 3; rx = q[0]; ry = q[1]; rz = q[2]; }
    const float* jr = a.j_rest_batched ? a.j_rest + b * (int64_t)J * 3 : a.j_rest;
    float tr[3] = {0.f, 0.f, 0.f};
    if (a.transl) { tr[0] = a.transl[b * 3]; tr[1] = a.transl[b * 3 + 1]; tr[2] = a.transl[b * 3 + 2]; }
    const Mat3 R = rodrigues(rx, ry, rz);
    float rel[3] = {0.f, 0.f, 0.f};
    if (on) {
#pragma unroll
        for (int k = 0; k < 3; ++k) rel[k] = jr[3 * I + k] - (I > 0 ? jr[3 * P + k] : 0.f);
    }
    Xf G;
#pragma unroll
    for (int k = 0; k < 9; ++k) G.r[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) G.t[k] = 0.f;
    for (int L = 0; L <= MAXD; ++L) {
        if (on && depth == L) {
            if (I == 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k) G.r[k] = R.m[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) G.t[k] = rel[k];
            } else {
                Xf Pm;
#pragma unroll
                for (int k = 0; k < 9; ++k) Pm.r[k] = sG[P][k];
#pragma unroll
                for (int k = 0; k < 3; ++k) Pm.t[k] = sG[P][9 + k];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        G.r[3 * r + c] = fmaf(Pm.r[3 * r], R.m[c], fmaf(Pm.r[3 * r + 1], R.m[3 + c], Pm.r[3 * r + 2] * R.m[6 + c]));
                    G.t[r] = fmaf(Pm.r[3 * r], rel[0], fmaf(Pm.r[3 * r + 1], rel[1], fmaf(Pm.r[3 * r + 2], rel[2], Pm.t[r])));
                }
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) sG[I][k] = G.r[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) sG[I][9 + k] = G.t[k];
        }
        __syncthreads();
    }
    if (!on) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) a.joints[b * a.joints_ld + 3 * I + k] = G.t[k] + tr[k];
    if (a.pf && I > 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) a.pf[FT<float>::index(b, (I - 1) * 9 + k, a.pf_K)] = R.m[k] - ((k % 4 == 0) ? 1.0f : 0.0f);
    }
    if (a.rel) {
        const float jx = jr[3 * I], jy = jr[3 * I + 1], jz = jr[3 * I + 2];
        f32x4* q = reinterpret_cast<f32x4*>(a.rel + (b * n_out + I) * 12);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            f32x4 v;
            v[0] = G.r[3 * r]; v[1] = G.r[3 * r + 1]; v[2] = G.r[3 * r + 2];
            v[3] = G.t[r] - (G.r[3 * r] * jx + G.r[3 * r + 1] * jy + G.r[3 * r + 2] * jz);
            q[r] = v;
        }
    }
}

struct dposer_body_s {
    dposer_body_desc d;
    int kind;   // 0 SMPL, 1 SMPL-H, 2 SMPL-X
    int parents[64];
    // k_skin_bwd_joints' table: the caller's CSR-by-joint skinning lists re-cut into vertex chunks by the explicit setup call
    // dposer_lbs_prepare_joint_lists (freed by dposer_body_destroy).  (Round 2 built it inside dposer_lbs_backward, keyed by the
    // lists' device ADDRESSES: a caller that re-uploaded other lists to a recycled address got the stale table.)
    bool jl_ready = false;
    int jl_chunks = 0;
    int32_t* jl_vstart = nullptr;    // device [chunks + 1] first vertex of each chunk
    int32_t* jl_ptr = nullptr;       // device [chunks][J + 1] entry ranges, relative to the chunk's first entry
    int32_t* jl_first = nullptr;     // device [chunks + 1] first entry of each chunk
    float2* jl_entry = nullptr;      // device [nnz] (weight, local vertex index as int bits)
    // k_skin_bwd_fused's tables (same setup call): every chunk's joint lists cut into segments of <= 32 entries
    bool jl_fused_ok = false;        // chunks are the regular 256-vertex grid, <= 1024 entries and <= 128 segments per chunk, K = 4
    int32_t* jl_seg = nullptr;       // device [chunks][128] x int2 (begin, end) relative to the chunk's first entry
    int32_t* jl_nseg = nullptr;      // device [chunks]
    int32_t* jl_jseg = nullptr;      // device [chunks][J] (first segment | count << 16) of joint j in chunk c
    // k_skin_bwd_mfma's table (same setup call): the skinning weights of every 256-vertex chunk as a dense [64 joints][256 vertices] matrix,
    // bf16 hi / lo planes, in the lane order of the MFMA A operand (64 KB per chunk; J <= 64, regular chunks)
    bool jl_mfma_ok = false;
    void* jl_wfrag = nullptr;
    // chunked forward: the skinning of chunk i runs on this stream beside the blend GEMM of chunk i + 1 (created on first use)
    hipStream_t side = nullptr;
    hipEvent_t ev_chunk[2] = {nullptr, nullptr}, ev_join = nullptr;
    // backward: the three product terms of the bf16 x 3 blend-gradient GEMM run side by side on the caller's stream and these two
    hipStream_t bwd_side[2] = {nullptr, nullptr};
    hipEvent_t ev_bwd_fork = nullptr, ev_bwd_join[2] = {nullptr, nullptr};
    // the side streams and events above belong to ONE device: the one that was current when they were created.  A handle that moves
    // with its module (.to(other_device)) gets them re-created there (side_streams_for_current_device)
    int side_dev = -1;
};
// (re)create nothing, but drop the handle's side streams / events when they belong to another device than the current one
static int side_streams_for_current_device(dposer_body_t h) {
    int dev = -1;
    DP_CHECK_HIP(hipGetDevice(&dev));
    if (h->side_dev == dev) return DPOSER_OK;
    if (h->side) (void)hipStreamDestroy(h->side);
    for (hipStream_t* q : {&h->bwd_side[0], &h->bwd_side[1]}) { if (*q) (void)hipStreamDestroy(*q); *q = nullptr; }
    for (hipEvent_t* e : {&h->ev_chunk[0], &h->ev_chunk[1], &h->ev_join, &h->ev_bwd_fork, &h->ev_bwd_join[0], &h->ev_bwd_join[1]}) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; }
    h->side = nullptr;
    h->side_dev = dev;
    return DPOSER_OK;
}

template <typename Kin> static bool same_tree(const int32_t* p, int n) {
    if (n != Kin::J) return false;
    for (int i = 0; i < n; ++i)
        if (p[i] != Kin::P[i] && !(i == 0 && p[i] < 0)) return false;
    return true;
}

extern "C" int dposer_body_create(const dposer_body_desc* desc, const int32_t* parents_host, dposer_body_t* out) {
    DP_CHECK_ARG(desc && parents_host && out, "null argument");
    int kind = -1;
    if (same_tree<KinSMPL>(parents_host, desc->num_joints)) kind = 0;
    else if (same_tree<KinSMPLH>(parents_host, desc->num_joints)) kind = 1;
    else if (same_tree<KinSMPLX>(parents_host, desc->num_joints)) kind = 2;
    if (kind < 0)
        return dposer_set_error(DPOSER_ERR_UNSUPPORTED,
                                "dposer_body_create: kinematic tree is not SMPL (24), SMPL-H (52) or SMPL-X (55); "
                                "the FK kernel unrolls the chain over a compile-time parents table");
    auto* h = new dposer_body_s();
    h->d = *desc;
    h->kind = kind;
    for (int i = 0; i < desc->num_joints && i < 64; ++i) h->parents[i] = parents_host[i] < 0 ? 0 : parents_host[i];
    *out = h;
    return DPOSER_OK;
}
extern "C" void dposer_body_destroy(dposer_body_t h) {
    if (!h) return;
    (void)hipFree(h->jl_vstart); (void)hipFree(h->jl_ptr); (void)hipFree(h->jl_first); (void)hipFree(h->jl_entry);
    (void)hipFree(h->jl_seg); (void)hipFree(h->jl_nseg); (void)hipFree(h->jl_jseg); (void)hipFree(h->jl_wfrag);
    if (h->side) (void)hipStreamDestroy(h->side);
    for (hipStream_t q : {h->bwd_side[0], h->bwd_side[1]})
        if (q) (void)hipStreamDestroy(q);
    for (hipEvent_t e : {h->ev_chunk[0], h->ev_chunk[1], h->ev_join, h->ev_bwd_fork, h->ev_bwd_join[0], h->ev_bwd_join[1]})
        if (e) (void)hipEventDestroy(e);
    delete h;
}

template <typename Kin> static hipError_t launch_fk(const FkArgs& a, hipStream_t st) {
#ifdef FK_NT
    constexpr int NT = FK_NT;
#else
    constexpr int NT = 64;
#endif
    if (a.B <= fk_small_max()) {                                    // one wave per pose, one lane per joint
        hipLaunchKernelGGL(k_fk_small<Kin>, dim3((unsigned)a.B), dim3(64), 0, st, a);
        return hipGetLastError();
    }
    const int lds_floats = NT * ((a.n_out * 3) | 1);
    // (a persistent variant that prefetches the next pose tile into registers while the chain runs was measured 13-20 % SLOWER --
    //  6.1 vs 7.1 G poses/s at 2^20 poses, 6.6 vs 8.3 at 2^22: 256 VGPRs with spills; tools/experimental/fk_stream.md)
    // (also measured and rejected: staging / evaluating / writing the joints in two or three column groups so that the LDS row
    //  shrinks and 16 instead of 9 waves fit a CU -- bit-identical, but the 132-byte row segments cost 2x: 3.9 vs 7.1 G poses/s)
    if (a.n_out == 22 && Kin::J >= 22 && !a.pf && !a.rel) {     // joints-only body query (the hot case): lean specialisation
        const bool dma_ok = NT == 64 && body_tuning().fk_dma && a.nseg >= 2 && a.seg[1] && a.seg_first[0] == 0 && a.seg_joints[0] == 1 &&
                            a.seg_first[1] == 1 && a.seg_joints[1] == 21 && a.joints_ld == 66 && a.B >= 64 &&
                            (((uintptr_t)a.seg[0] | (uintptr_t)a.seg[1] | (uintptr_t)a.joints) & 15) == 0;
        if (dma_ok) {        // full blocks of 64 poses through the DMA kernel, the remaining < 64 poses through the general one
            const int64_t nfull = a.B / 64, rem = a.B - nfull * 64;
            hipLaunchKernelGGL(k_fk_joints_dma<Kin>, dim3((unsigned)nfull), dim3(64), 64 * 66 * sizeof(float) + (size_t)body_tuning().fk_lds_pad, st, a);
            if (rem == 0) return hipGetLastError();
            FkArgs t = a;
            const int64_t o = nfull * 64;
            t.seg[0] = a.seg[0] ? a.seg[0] + o * 3 : nullptr; t.seg[1] = a.seg[1] + o * 63;
            for (int sg = 2; sg < a.nseg && sg < FK_MAX_SEG; ++sg) t.seg[sg] = a.seg[sg] ? a.seg[sg] + o * a.seg_joints[sg] * 3 : nullptr;
            if (a.j_rest_batched) t.j_rest = a.j_rest + o * Kin::J * 3;
            if (a.transl) t.transl = a.transl + o * 3;
            t.joints = a.joints + o * a.joints_ld;
            t.B = rem;
            hipLaunchKernelGGL((k_fk_joints<Kin, NT, 22>), dim3(1), dim3(NT), lds_floats * sizeof(float), st, t);
            return hipGetLastError();
        }
        hipLaunchKernelGGL((k_fk_joints<Kin, NT, 22>), dim3((unsigned)ceil_div(a.B, NT)), dim3(NT), lds_floats * sizeof(float), st, a);
    }
    else
        hipLaunchKernelGGL((k_fk_joints<Kin, NT>), dim3((unsigned)ceil_div(a.B, NT)), dim3(NT), lds_floats * sizeof(float), st, a);
    return hipGetLastError();
}

extern "C" int dposer_fk_joints(dposer_body_t h, const float* const* pose_segments_host, const int32_t* segment_joints_host,
                                int32_t num_segments, const float* j_rest, int32_t j_rest_batched, const float* transl, float* joints,
                                float* rel_transforms, int32_t n_out, int64_t batch, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(h && pose_segments_host && segment_joints_host && j_rest && joints, "null argument");
    DP_CHECK_ARG(num_segments >= 1 && num_segments <= FK_MAX_SEG, "1..8 pose segments");
    DP_CHECK_ARG(batch > 0, "batch must be positive");
    FkArgs a;
    std::memset(&a, 0, sizeof(a));
    int first = 0;
    for (int i = 0; i < num_segments; ++i) {
        a.seg[i] = pose_segments_host[i];
        a.seg_first[i] = first;
        a.seg_joints[i] = segment_joints_host[i];
        first += segment_joints_host[i];
    }
    DP_CHECK_ARG(first == h->d.num_joints, "pose segments must cover all joints of the kinematic tree");
    DP_CHECK_ARG(n_out >= 1 && n_out <= h->d.num_joints, "n_out must be in 1..num_joints");
    a.nseg = num_segments; a.j_rest = j_rest; a.j_rest_batched = j_rest_batched; a.transl = transl; a.joints = joints;
    a.rel = rel_transforms; a.n_out = n_out; a.B = batch; a.joints_ld = (int64_t)n_out * 3;
    hipStream_t st = (hipStream_t)stream;
    if (h->kind == 0) FK_HIP_LAUNCH(launch_fk<KinSMPL>(a, st));
    else if (h->kind == 1) FK_HIP_LAUNCH(launch_fk<KinSMPLH>(a, st));
    else FK_HIP_LAUNCH(launch_fk<KinSMPLX>(a, st));
    return DPOSER_OK;
}


// ------------------------------------------------------------------------------------------------
// Linear blend skinning  (smplx lbs.py lbs(); reference call site lib/body_model/body_model.py:75-103)
//   v_posed = v_shaped + pose_feature @ posedirs            -> fp32 MFMA GEMM (exact fp32 fma chain)
//   T_v     = sum_j W[v][j] A_j ;  v = T_v [v_posed ; 1]     -> ELL-sparse skinning, A staged in LDS
//   joints  = [J posed joints | vertex-selected extras | barycentric landmarks] (+ transl)
// ------------------------------------------------------------------------------------------------
// FMA contraction for the code that FORMS SKINNED VERTICES (round 6).  This file is compiled with -ffp-contract=off; the transform blend is 48
// multiply-add pairs per (pose, vertex), and the motion-denoising backward that skins three frames per pose (k_skin_bwd_mfma<TMP>) is
// VALU-issue-bound: contracted it is 15 % fewer VALU instructions and the batched loop 5 % faster (profiles/r06_skin_contract_ab.md).  Every kernel
// that forms vertices (k_skin, k_skin_x4, k_skin_run, k_skin_temporal, the TMP pre-pass) carries the pragma over the SAME two expressions (transform
// blend, transform x position), so the bit-identity relations between them hold (tests); the temporal term's gradient is formed from those vertices
// with contraction OFF again (DP_SKIN_FP_OFF: k_md_vert_grad in tasks.hip writes `gx -= bx * inv` where the fused forms write `ax * inv - hold` --
// contracted, the two would round differently).  The plain backward k_skin_bwd_mfma<TMP = false> does NOT: contracted it is 7 % slower (same file, A/B'd
// in rounds 5 and 6) and its transform blend is bound to no other kernel's bits.
#define DP_SKIN_FP _Pragma("clang fp contract(fast)")
#define DP_SKIN_FP_OFF _Pragma("clang fp contract(off)")
struct SkinArgs {
    const float* offsets;      // [B][ld_off] pose-blend offsets (3V valid)
    int64_t ld_off;
    const float* v_shaped;     // [V][3] or [B][V][3]
    int v_shaped_batched;
    const float* A;            // [B][J][12]
    const int32_t* skin_idx;   // [V][K]
    const float* skin_w;       // [V][K]
    int K, J, V;
    const float* transl;       // [B][3] or null
    float* verts;              // [B][V][3]
};
// T = sum_k w_k A[idx_k] for one vertex.  K = 4 (every SMPL-family asset: at most four bones per vertex) fetches the ELL row as ONE
// 16-byte load per table: as four dword loads per table, each wave instruction touched 64 lanes x 16-byte stride = eight 128-B lines at
// a quarter density, and the eight of them were more than half of the kernel's vector-memory work.
__device__ __forceinline__ void skin_transform(const SkinArgs& a, const float* sA, int v, float (&T)[12]) {
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = 0.f;
    if (a.K == 4) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.skin_w + (int64_t)v * 4);
        const int4 j4 = *reinterpret_cast<const int4*>(a.skin_idx + (int64_t)v * 4);
        const int jj[4] = {j4.x, j4.y, j4.z, j4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4* Aj = reinterpret_cast<const f32x4*>(sA) + jj[k] * 3;        // 16-byte reads: see k_skin_x4
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const f32x4 row = Aj[r];
#pragma unroll
                for (int i = 0; i < 4; ++i) T[4 * r + i] += w4[k] * row[i];
            }
        }
    } else {
        for (int k = 0; k < a.K; ++k) {
            const float w = a.skin_w[(int64_t)v * a.K + k];
            const float* Aj = sA + a.skin_idx[(int64_t)v * a.K + k] * 12;
#pragma unroll
            for (int i = 0; i < 12; ++i) T[i] += w * Aj[i];
        }
    }
}
// (staging the block's coordinates through LDS so that they enter and leave as consecutive dwords -- what pays in k_skin_bwd -- was
//  measured SLOWER here: 461 -> 569 us at 4096 poses, four extra barriers per 256 vertices)
__global__ void __launch_bounds__(256) k_skin(SkinArgs a) {
    DP_SKIN_FP
    extern __shared__ __attribute__((aligned(16))) float sA[];   // [J][12]
    const int64_t b = blockIdx.y;
    for (int i = threadIdx.x; i < a.J * 12; i += 256) sA[i] = a.A[b * a.J * 12 + i];
    __syncthreads();
    float tr[3] = {0.f, 0.f, 0.f};
    if (a.transl) { tr[0] = a.transl[b * 3]; tr[1] = a.transl[b * 3 + 1]; tr[2] = a.transl[b * 3 + 2]; }
    for (int v = blockIdx.x * 256 + threadIdx.x; v < a.V; v += gridDim.x * 256) {
        const float* vs = a.v_shaped + (a.v_shaped_batched ? b * a.V * 3 : 0) + (int64_t)v * 3;
        const float* off = a.offsets + b * a.ld_off + (int64_t)v * 3;
        const float px = vs[0] + off[0], py = vs[1] + off[1], pz = vs[2] + off[2];
        float T[12];
        skin_transform(a, sA, v, T);
        float* o = a.verts + (b * a.V + v) * 3;
        o[0] = T[0] * px + T[1] * py + T[2] * pz + T[3] + tr[0];
        o[1] = T[4] * px + T[5] * py + T[6] * pz + T[7] + tr[1];
        o[2] = T[8] * px + T[9] * py + T[10] * pz + T[11] + tr[2];
    }
}

// k_skin with FOUR vertices per thread in flight (K = 4): every load of the four (rest shape, offsets, ELL row as one 16-byte
// load per table) is issued before the first use.  Round-3 A/Bs on this kernel (tools/lbs_skin_ab.py, 4096 / 16384 poses, LBS
// forward 0.885 ms with k_skin): this form 0.874 ms; 16-byte ELL loads alone 0.905; a wave-transposed variant (64 vertices = 192
// consecutive floats per wave, loaded / stored as fully coalesced dwords and turned into one vertex per lane through a
// wave-private LDS buffer) 0.964-1.0 ms.  PMC shows the waves parked in s_waitcnt 74 % of their cycles at 2.2 TB/s, and neither more
// loads in flight nor fewer cache-line touches move it: the kernel's 1.03 GB meet the pose-blend GEMM's freshly written 515 MB of
// offsets on their way out of L2 / MALL -- keeping the offsets out of HBM (a fused skinning epilogue) is what would help, and the
// per-(pose, vertex) transform T = sum_k w_k A[pose][j_k] makes that epilogue need 256 poses x 55 x 12 floats (675 KB) per tile.
__global__ void __launch_bounds__(256) k_skin_x4(SkinArgs a) {
    DP_SKIN_FP
    extern __shared__ __attribute__((aligned(16))) float sA[];   // [J][12]
    const int64_t b = blockIdx.y;
    for (int i = threadIdx.x; i < a.J * 12; i += 256) sA[i] = a.A[b * a.J * 12 + i];
    float tr[3] = {0.f, 0.f, 0.f};
    if (a.transl) { tr[0] = a.transl[b * 3]; tr[1] = a.transl[b * 3 + 1]; tr[2] = a.transl[b * 3 + 2]; }
    const float* vs_row = a.v_shaped + (a.v_shaped_batched ? b * a.V * 3 : 0);
    const float* off_row = a.offsets + b * a.ld_off;
    float* out_row = a.verts + b * a.V * 3;
    const int vbase = blockIdx.x * 1024 + threadIdx.x;
    float p[4][3];
    f32x4 w4[4];
    int4 j4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int v = vbase + u * 256;
        const int vc = v < a.V ? v : a.V - 1;                     // clamp: the tail threads load a valid vertex and do not store
#pragma unroll
        for (int c = 0; c < 3; ++c) p[u][c] = vs_row[(int64_t)vc * 3 + c] + off_row[(int64_t)vc * 3 + c];
        w4[u] = *reinterpret_cast<const f32x4*>(a.skin_w + (int64_t)vc * 4);
        j4[u] = *reinterpret_cast<const int4*>(a.skin_idx + (int64_t)vc * 4);
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int v = vbase + u * 256;
        const int jj[4] = {j4[u].x, j4[u].y, j4[u].z, j4[u].w};
        float T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // three 16-byte LDS reads per joint (the 48-byte transform records are 16-byte aligned).  As scalar float reads hipcc emitted
            // 24 ds_read2_b32 per vertex -- 4 LDS cycles each, banked modulo 32 dwords, so that joints 8 apart collide: the PMC showed the
            // LDS pipe busy for the whole kernel, 75 % of it bank-conflict cycles (1.9e8 of 2.6e8 at 4096 poses)
            const f32x4* Aj = reinterpret_cast<const f32x4*>(sA) + jj[k] * 3;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const f32x4 row = Aj[r];
#pragma unroll
                for (int i = 0; i < 4; ++i) T[4 * r + i] += w4[u][k] * row[i];
            }
        }
        if (v < a.V) {
            float* o = out_row + (int64_t)v * 3;
            o[0] = T[0] * p[u][0] + T[1] * p[u][1] + T[2] * p[u][2] + T[3] + tr[0];
            o[1] = T[4] * p[u][0] + T[5] * p[u][1] + T[6] * p[u][2] + T[7] + tr[1];
            o[2] = T[8] * p[u][0] + T[9] * p[u][1] + T[10] * p[u][2] + T[11] + tr[2];
        }
    }
}

// k_skin_x4 over a RUN of poses (round 4): a block keeps its 1024 vertices' ELL rows (and the shared rest shape) in registers and walks
// `run` consecutive poses -- per pose it reads only the 12 bytes of offsets per vertex instead of 56 (offsets + rest shape + two ELL
// tables: the tables alone were 1.37 GB of L1 / L2 traffic per 4096 poses, more than the 1.03 GB the kernel moves through HBM), the next
// pose's offsets and transforms are in flight while the current pose is skinned (transforms double-buffered in LDS, one barrier per
// pose).  Same expressions in the same order per vertex: bit-identical vertices.
__global__ void __launch_bounds__(256) k_skin_run(SkinArgs a, int run, int64_t B) {
    DP_SKIN_FP
    extern __shared__ __attribute__((aligned(16))) float sA2[];   // [2][J][12]
    const int J12 = a.J * 12;
    const int64_t b0 = (int64_t)blockIdx.y * run;
    const int64_t b1 = b0 + run < B ? b0 + run : B;
    const int vbase = blockIdx.x * 1024 + threadIdx.x;
    f32x4 w4[4];
    int4 j4[4];
    int vc[4];
    float vs[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int v = vbase + u * 256;
        vc[u] = v < a.V ? v : a.V - 1;                            // clamp: the tail threads load a valid vertex and do not store
        w4[u] = *reinterpret_cast<const f32x4*>(a.skin_w + (int64_t)vc[u] * 4);
        j4[u] = *reinterpret_cast<const int4*>(a.skin_idx + (int64_t)vc[u] * 4);
        if (!a.v_shaped_batched)
#pragma unroll
            for (int c = 0; c < 3; ++c) vs[u][c] = a.v_shaped[(int64_t)vc[u] * 3 + c];
    }
    // pose b0: transforms into buffer 0, offsets (and the batched rest shape) into registers
    float An[3];                                                   // this thread's <= 3 floats of the NEXT pose's transforms
    float on[4][3], vn[4][3];
    auto fetch = [&](int64_t b) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 3; ++i) { const int k = threadIdx.x + i * 256; An[i] = k < J12 ? a.A[b * J12 + k] : 0.f; }
        const float* off_row = a.offsets + b * a.ld_off;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) on[u][c] = off_row[(int64_t)vc[u] * 3 + c];
        if (a.v_shaped_batched) {
            const float* vs_row = a.v_shaped + b * a.V * 3;
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int c = 0; c < 3; ++c) vn[u][c] = vs_row[(int64_t)vc[u] * 3 + c];
        }
    };
    fetch(b0);
    for (int64_t b = b0; b < b1; ++b) {
        float* sA = sA2 + ((b - b0) & 1) * J12;
#pragma unroll
        for (int i = 0; i < 3; ++i) { const int k = threadIdx.x + i * 256; if (k < J12) sA[k] = An[i]; }
        float p[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) p[u][c] = (a.v_shaped_batched ? vn[u][c] : vs[u][c]) + on[u][c];
        float tr[3] = {0.f, 0.f, 0.f};
        if (a.transl) { tr[0] = a.transl[b * 3]; tr[1] = a.transl[b * 3 + 1]; tr[2] = a.transl[b * 3 + 2]; }
        __syncthreads();                                           // buffer (b & 1) is complete; every wave is done reading it two poses ago
        if (b + 1 < b1) fetch(b + 1);                              // in flight while this pose is skinned
        float* out_row = a.verts + b * a.V * 3;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int v = vbase + u * 256;
            const int jj[4] = {j4[u].x, j4[u].y, j4[u].z, j4[u].w};
            float T[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) T[i] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4* Aj = reinterpret_cast<const f32x4*>(sA) + jj[k] * 3;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const f32x4 row = Aj[r];
#pragma unroll
                    for (int i = 0; i < 4; ++i) T[4 * r + i] += w4[u][k] * row[i];
                }
            }
            if (v < a.V) {
                float* o = out_row + (int64_t)v * 3;
                o[0] = T[0] * p[u][0] + T[1] * p[u][1] + T[2] * p[u][2] + T[3] + tr[0];
                o[1] = T[4] * p[u][0] + T[5] * p[u][1] + T[6] * p[u][2] + T[7] + tr[1];
                o[2] = T[8] * p[u][0] + T[9] * p[u][1] + T[10] * p[u][2] + T[11] + tr[2];
            }
        }
    }
}

struct ExtraArgs {
    const float* verts;          // [B][V][3] (transl already applied)
    const int32_t* extra_ids;    // [n_extra] vertex ids   (smplx VertexJointSelector)
    const int32_t* lmk_tri;      // [n_lmk][3] vertex ids of the landmark faces (faces[lmk_faces_idx])
    const float* lmk_bary;       // [n_lmk][3]
    float* joints;               // [B][ld] rows; entries [J, J + n_extra + n_lmk) written here
    int64_t ld;
    int J, n_extra, n_lmk, V;
    int64_t B;
};
__global__ void __launch_bounds__(256) k_extra_joints(ExtraArgs a) {
    const int per = a.n_extra + a.n_lmk;
    const int64_t total = a.B * per;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / per;
        const int e = (int)(i % per);
        const float* vb = a.verts + b * a.V * 3;
        float x, y, z;
        if (e < a.n_extra) {
            const float* p = vb + (int64_t)a.extra_ids[e] * 3;
            x = p[0]; y = p[1]; z = p[2];
        } else {
            const int l = e - a.n_extra;
            x = y = z = 0.f;
#pragma unroll
            for (int f = 0; f < 3; ++f) {      // einsum('blfi,blf->bli') in smplx vertices2landmarks
                const float* p = vb + (int64_t)a.lmk_tri[l * 3 + f] * 3;
                const float w = a.lmk_bary[l * 3 + f];
                x += p[0] * w; y += p[1] * w; z += p[2] * w;
            }
        }
        float* o = a.joints + b * a.ld + (int64_t)(a.J + e) * 3;
        o[0] = x; o[1] = y; o[2] = z;
    }
}

// ---- pose-blend arithmetic -----------------------------------------------------------------------------------------------------
// offsets = pose_feature @ posedirs is 2 * B * 486 * 3V FLOPs (125 GFLOP at 4096 SMPL-X poses): on the exact-fp32 MFMA
// (v_mfma_f32_32x32x2_f32, 157 TFLOP/s peak) it is 2/3 of the LBS forward.  Default: both operands as TWO bf16 terms each
// (x = hi + lo, hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits) and the three products hi*hi + hi*lo + lo*hi on the bf16 MFMA
// with fp32 accumulation -- one GEMM with K = 3 * 512 (forward: [pf_hi | pf_hi | pf_lo] x [pd_hi ; pd_lo ; pd_hi] through the
// kernel's K segments), three split-K launches into one slab set (backward).  The dropped lo*lo term and the split residuals
// are <= 2^-16 of each product: the offsets (centimetres) move by < 1e-6 m, vertices agree with the fp64 oracle to ~1e-6
// (tests/test_gpu_fk.py; the bar is 1e-5).  DPOSER_LBS_BLEND=fp32 selects the exact-fp32 chain (both packings are kept).
static bool lbs_blend_fp32() { return body_tuning().blend_fp32; }
// FT32 [Bpad][Kin] (its first K columns) -> FT bf16 [Bpad][3K] = [hi | hi | lo]
__global__ void __launch_bounds__(256) k_split_pf(const float* __restrict__ pf, __bf16* __restrict__ out, int64_t Bpad, int Kin, int K) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one 8-element chunk of the bf16 layout
    const int chunks = K / 8;
    if (i >= Bpad * chunks) return;
    const int64_t b = i / chunks;
    const int k0 = (int)(i % chunks) * 8;
    __bf16 hi[8], lo[8];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(pf + FT<float>::index(b, k0 + 4 * q, Kin));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const __bf16 h = (__bf16)v[r];
            hi[4 * q + r] = h;
            lo[4 * q + r] = (__bf16)(v[r] - (float)h);
        }
    }
    const u32x4 H = *reinterpret_cast<u32x4*>(hi), L = *reinterpret_cast<u32x4*>(lo);
    *reinterpret_cast<u32x4*>(out + FT<__bf16>::index(b, k0, 3 * K)) = H;
    *reinterpret_cast<u32x4*>(out + FT<__bf16>::index(b, K + k0, 3 * K)) = H;
    *reinterpret_cast<u32x4*>(out + FT<__bf16>::index(b, 2 * K + k0, 3 * K)) = L;
}

// Pose-feature columns that can be non-zero: joint I >= 1 owns columns [(I-1) 9, I 9) (pose_feature = R_I - 1); a NULL pose segment is
// the identity rotation (smplx's default zeros), its columns are exactly zero.  The fitting loops and the benchmarks pose the BODY only
// (joints 1..21 of 55: hands, jaw and eyes stay NULL), so the pose-blend GEMM reduces over 192 of its 512 padded columns and the
// blend-gradient GEMMs produce 256 of theirs -- adding exact zeros changes no bit of the result, skipping them saves 62 % / 50 % of the
// two largest GEMMs of the body model.  (`want` = the segment's gradient pointers for the backward; null: the pose pointers.)
static int lbs_posed_cols(const float* const* segs, const int32_t* seg_joints, int nseg, int J, const float* const* want = nullptr) {
    int first = 0, last = 0;
    for (int i = 0; i < nseg; ++i) {
        const bool on = want ? (want[i] != nullptr && segs[i] != nullptr) : segs[i] != nullptr;
        if (on && seg_joints[i] > 0) last = first + seg_joints[i] - 1;
        first += seg_joints[i];
    }
    if (last > J - 1) last = J - 1;
    return last * 9;
}
static int64_t lbs_pad_batch(int64_t B) { return round_up(B, 128); }
static int lbs_ppad(int J) { return (int)round_up((J - 1) * 9, 32); }
static int64_t lbs_cpad(int V) { return round_up((int64_t)V * 3, 128); }

extern "C" int64_t dposer_lbs_posedirs_packed_bytes(dposer_body_t h) {
    if (!h) return -1;
    return lbs_cpad(h->d.num_vertices) * lbs_ppad(h->d.num_joints) * 8;      // FT32 | bf16 high parts | bf16 low parts
}
// posedirs [(J-1)*9][V*3] fp32 row-major (the layout smplx keeps after its reshape/transposition)
extern "C" int dposer_lbs_pack_posedirs(dposer_body_t h, const float* posedirs, void* packed, void* stream) {
    DP_CHECK_ARG(h && posedirs && packed, "null argument");
    PackJobs js;
    js.n = 1;
    PackJob& j = js.job[0];
    const int P = (h->d.num_joints - 1) * 9;
    j.dst_off = 0; j.src_off = 0; j.ktot = lbs_ppad(h->d.num_joints); j.koff = 0;
    j.rows_pad = (int)lbs_cpad(h->d.num_vertices); j.kpad = j.ktot; j.rows_valid = h->d.num_vertices * 3; j.cols_valid = P;
    j.ld = h->d.num_vertices * 3; j.trans = 1; j.f32 = 1; j.split = 0;
    const int64_t n = (int64_t)j.rows_pad * j.ktot;
    js.n = 3;
    js.job[1] = j; js.job[1].dst_off = n * 4; js.job[1].f32 = 0; js.job[1].split = 1;
    js.job[2] = j; js.job[2].dst_off = n * 6; js.job[2].f32 = 0; js.job[2].split = 2;
    FK_HIP_LAUNCH(launch_pack(js, posedirs, packed, (hipStream_t)stream));
    return DPOSER_OK;
}
extern "C" int64_t dposer_lbs_workspace_bytes(dposer_body_t h, int64_t batch) {
    if (!h || batch <= 0) return -1;
    const int64_t Bpad = lbs_pad_batch(batch);
    int64_t p = 0;
    p += round_up(Bpad * lbs_ppad(h->d.num_joints) * 4, 256);          // pose feature FT32
    p += round_up(batch * h->d.num_joints * 12 * 4, 256);              // A
    p += round_up(batch * lbs_cpad(h->d.num_vertices) * 4, 256);       // offsets
    p += round_up(Bpad * 3 * lbs_ppad(h->d.num_joints) * 2, 256);      // pose feature as bf16 [hi | hi | lo]
    return p;
}

// FK + pose-blend GEMM of the LBS forward (steps 1 and 2); A / offsets: where the skinning stage finds its inputs in `ws`
// The pose-blend GEMM of the padded pose rows [r0, r1) (multiples of the 256 / 128-row tile): lbs_forward_front runs it over the whole batch,
// the chunked forward (dposer_lbs_forward) chunk by chunk with the skinning of the previous chunk running beside it.
struct LbsBlend {
    float* pf;
    __bf16* pf_split;
    float* offsets;
    const void* posedirs_packed;
    int Keff, Ppad, V;
    int64_t Cpad, Bpad, batch;
};
static int lbs_blend_rows(const LbsBlend& b, int64_t r0, int64_t r1, hipStream_t st) {
    GemmArgs g;
    std::memset(&g, 0, sizeof(g));
    WgradParams wp;
    const int64_t valid = (b.batch < r1 ? b.batch : r1) - r0;
    if (valid <= 0) return DPOSER_OK;
    wp.slab = b.offsets + r0 * b.Cpad; wp.slab_stride = 0; wp.ld = (int)b.Cpad; wp.N_valid = (int)valid; wp.K_valid = b.V * 3;
    g.ksplit = 1;
    const int64_t rows = r1 - r0;
    if (lbs_blend_fp32()) {
        g.W = (const char*)b.pf + (r0 / 32) * (int64_t)(b.Ppad / 8) * 1024; g.w_stride_blocks = b.Ppad / 8; g.n_cblk = (int)(rows / 128); g.n_sblk = (int)(b.Cpad / 128);
        g.src[0] = b.posedirs_packed; g.seg_kblocks[0] = b.Keff / 8; g.seg_stride_blocks[0] = b.Ppad / 8; g.nseg = 1; g.ktot_blocks = b.Keff / 8;
        FK_HIP_LAUNCH(gemm_wgrad(PREC_FP32, SHAPE_MID, g, wp, st));
    } else {
        const char* hi = (const char*)b.posedirs_packed + b.Cpad * b.Ppad * 4;
        const char* lo = hi + b.Cpad * b.Ppad * 2;
        const int kb = b.Ppad / 16, kbe = b.Keff / 16;
        const int shape = (rows % 256 == 0 && b.Cpad % 256 == 0 && b.Bpad >= 1024 && body_tuning().lbs_fwd_big) ? SHAPE_BIG : SHAPE_MID;
        const int tile = shape == SHAPE_BIG ? 256 : 128;
        g.W = (const char*)b.pf_split + (r0 / 32) * (int64_t)(3 * kbe) * 1024; g.w_stride_blocks = 3 * kbe; g.n_cblk = (int)(rows / tile); g.n_sblk = (int)(b.Cpad / tile);
        g.src[0] = hi; g.src[1] = lo; g.src[2] = hi;                        // [pf_hi | pf_hi | pf_lo] x [hi ; lo ; hi]
        g.seg_kblocks[0] = g.seg_kblocks[1] = g.seg_kblocks[2] = kbe; g.nseg = 3; g.ktot_blocks = 3 * kbe;
        g.seg_stride_blocks[0] = g.seg_stride_blocks[1] = g.seg_stride_blocks[2] = kb;      // (rows of the packed posedirs keep their full width)
        FK_HIP_LAUNCH(gemm_wgrad(PREC_BF16, shape, g, wp, st));
    }
    return DPOSER_OK;
}

static int lbs_forward_front(dposer_body_t h, void* ws, const void* posedirs_packed, const float* const* pose_segments_host,
                             const int32_t* segment_joints_host, int32_t num_segments, const float* j_rest, int32_t j_rest_batched,
                             const float* transl, float* joints, int64_t batch, void* stream, float** A_out, float** offsets_out,
                             LbsBlend* blend = nullptr) {
    DP_CHECK_ARG(h && ws && posedirs_packed && pose_segments_host && segment_joints_host && j_rest && joints, "null argument");
    DP_CHECK_ARG(batch > 0, "bad size");
    DP_CHECK_ARG(((uintptr_t)ws & 255) == 0 && ((uintptr_t)posedirs_packed & 255) == 0, "workspace / packed posedirs must be 256-byte aligned");
    DP_CHECK_ARG(num_segments >= 1 && num_segments <= FK_MAX_SEG, "1..8 pose segments");
    hipStream_t st = (hipStream_t)stream;
    const int J = h->d.num_joints, V = h->d.num_vertices;
    const int64_t Bpad = lbs_pad_batch(batch);
    const int Ppad = lbs_ppad(J);
    const int64_t Cpad = lbs_cpad(V);
    char* p = (char*)ws;
    float* pf = (float*)p; p += round_up(Bpad * Ppad * 4, 256);
    float* A = (float*)p; p += round_up(batch * J * 12 * 4, 256);
    float* offsets = (float*)p; p += round_up(batch * Cpad * 4, 256);
    __bf16* pf_split = (__bf16*)p;
    const int n_total = J + h->d.num_extra + h->d.num_landmarks;

    // 1. FK: posed joints -> joints[:, :J], skinning transforms A, pose feature (FT32 operand of the blend GEMM)
    DP_CHECK_HIP(hipMemsetAsync(pf, 0, Bpad * Ppad * 4, st));
    FkArgs a;
    std::memset(&a, 0, sizeof(a));
    int first = 0;
    for (int i = 0; i < num_segments; ++i) {
        a.seg[i] = pose_segments_host[i];
        a.seg_first[i] = first;
        a.seg_joints[i] = segment_joints_host[i];
        first += segment_joints_host[i];
    }
    DP_CHECK_ARG(first == J, "pose segments must cover all joints of the kinematic tree");
    a.nseg = num_segments; a.j_rest = j_rest; a.j_rest_batched = j_rest_batched; a.transl = transl; a.joints = joints;
    a.joints_ld = (int64_t)n_total * 3; a.rel = A; a.pf = pf; a.pf_K = Ppad; a.n_out = J; a.B = batch;
    if (h->kind == 0) FK_HIP_LAUNCH(launch_fk<KinSMPL>(a, st));
    else if (h->kind == 1) FK_HIP_LAUNCH(launch_fk<KinSMPLH>(a, st));
    else FK_HIP_LAUNCH(launch_fk<KinSMPLX>(a, st));

    // 2. pose-blend offsets[b][3V] = pose_feature @ posedirs: rows = poses, lanes = vertex coords
    int Keff = body_tuning().lbs_k_prefix ? (int)round_up(lbs_posed_cols(pose_segments_host, segment_joints_host, num_segments, J), 32) : Ppad;
    Keff = Keff < 32 ? 32 : (Keff > Ppad ? Ppad : Keff);       // K prefix: the columns of the joints that are posed, in whole 32-column stages, at least one
    if (!lbs_blend_fp32()) {
        hipLaunchKernelGGL(k_split_pf, dim3((unsigned)ceil_div(Bpad * (Keff / 8), 256)), dim3(256), 0, st, (const float*)pf, pf_split, Bpad, Ppad, Keff);
        FK_HIP_LAUNCH(hipGetLastError());
    }
    if (blend) {
        blend->pf = pf; blend->pf_split = pf_split; blend->offsets = offsets; blend->posedirs_packed = posedirs_packed;
        blend->Keff = Keff; blend->Ppad = Ppad; blend->Cpad = Cpad; blend->Bpad = Bpad; blend->batch = batch; blend->V = V;
    } else {
        LbsBlend b;
        b.pf = pf; b.pf_split = pf_split; b.offsets = offsets; b.posedirs_packed = posedirs_packed;
        b.Keff = Keff; b.Ppad = Ppad; b.Cpad = Cpad; b.Bpad = Bpad; b.batch = batch; b.V = V;
        DP_TRY(lbs_blend_rows(b, 0, Bpad, st));
    }
    *A_out = A;
    *offsets_out = offsets;
    return DPOSER_OK;
}

extern "C" int dposer_lbs_forward(dposer_body_t h, void* ws, const void* posedirs_packed, const float* const* pose_segments_host,
                                  const int32_t* segment_joints_host, int32_t num_segments, const float* j_rest, int32_t j_rest_batched,
                                  const float* v_shaped, int32_t v_shaped_batched, const int32_t* skin_idx, const float* skin_w,
                                  int32_t skin_k, const float* transl, const int32_t* extra_vertex_ids, const int32_t* lmk_tri,
                                  const float* lmk_bary, float* verts, float* joints, int64_t batch, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(v_shaped && skin_idx && skin_w && verts, "null argument");
    DP_CHECK_ARG(skin_k >= 1, "bad size");
    float *A = nullptr, *offsets = nullptr;
    hipStream_t st = (hipStream_t)stream;
    const int J = h->d.num_joints, V = h->d.num_vertices;
    const int64_t Cpad = lbs_cpad(V);
    const int n_total = J + h->d.num_extra + h->d.num_landmarks;
    const int64_t chunk = body_tuning().lbs_fwd_chunk;
    const bool chunked = chunk >= 256 && batch > chunk;
    LbsBlend blend;
    DP_TRY(lbs_forward_front(h, ws, posedirs_packed, pose_segments_host, segment_joints_host, num_segments, j_rest, j_rest_batched, transl, joints,
                             batch, stream, &A, &offsets, chunked ? &blend : nullptr));
    // 3. skinning of the poses [b0, b0 + nb)
    auto skin = [&](int64_t b0, int64_t nb, hipStream_t ss) -> int {
        SkinArgs s;
        s.offsets = offsets + b0 * Cpad; s.ld_off = Cpad; s.v_shaped = v_shaped_batched ? v_shaped + b0 * V * 3 : v_shaped; s.v_shaped_batched = v_shaped_batched;
        s.A = A + b0 * J * 12; s.skin_idx = skin_idx; s.skin_w = skin_w; s.K = skin_k; s.J = J; s.V = V; s.transl = transl ? transl + b0 * 3 : nullptr;
        s.verts = verts + b0 * V * 3;
        dim3 grid((unsigned)ceil_div(V, 256 * 4), (unsigned)nb);
        const int mode = body_tuning().skin_mode;
        if (mode >= 3 && skin_k == 4 && J * 12 <= 768) {
            // poses per block: enough blocks for ~4 per resident slot (256 CUs x 8), at most 16 poses
            int run = 1;
            while (run < 16 && (int64_t)grid.x * ceil_div(nb, run * 2) >= 8192) run *= 2;
            hipLaunchKernelGGL(k_skin_run, dim3(grid.x, (unsigned)ceil_div(nb, run)), dim3(256), 2 * J * 12 * sizeof(float), ss, s, run, (int64_t)nb);
        } else if (mode != 0 && skin_k == 4) hipLaunchKernelGGL(k_skin_x4, grid, dim3(256), J * 12 * sizeof(float), ss, s);
        else hipLaunchKernelGGL(k_skin, grid, dim3(256), J * 12 * sizeof(float), ss, s);
        FK_HIP_LAUNCH(hipGetLastError());
        return DPOSER_OK;
    };
    if (!chunked) {
        DP_TRY(skin(0, batch, st));
    } else {
        // blend GEMM chunk by chunk on the caller's stream; the skinning of a finished chunk runs on the side stream beside the next
        // chunk's GEMM (matrix pipe beside a streaming kernel), and reads its pose-blend offsets while they are still cache-resident
        DP_TRY(side_streams_for_current_device(h));
        if (!h->side) {
            DP_CHECK_HIP(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
            for (hipEvent_t* e : {&h->ev_chunk[0], &h->ev_chunk[1], &h->ev_join}) DP_CHECK_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        }
        // (a failing launch must not leave the side stream's kernels unjoined: the caller may free or reuse the workspace as soon as
        //  this call has returned -- the join below runs on the error path too)
        const bool serial = body_tuning().lbs_fwd_chunk_serial;
        auto chunks = [&]() -> int {
            int k = 0;
            for (int64_t b0 = 0; b0 < batch; b0 += chunk, ++k) {
                const int64_t r1 = b0 + chunk < blend.Bpad ? b0 + chunk : blend.Bpad;
                DP_TRY(lbs_blend_rows(blend, b0, b0 + chunk >= batch ? blend.Bpad : r1, st));
                if (serial) { DP_TRY(skin(b0, (b0 + chunk < batch ? b0 + chunk : batch) - b0, st)); continue; }
                DP_CHECK_HIP(hipEventRecord(h->ev_chunk[k & 1], st));
                DP_CHECK_HIP(hipStreamWaitEvent(h->side, h->ev_chunk[k & 1], 0));
                DP_TRY(skin(b0, (b0 + chunk < batch ? b0 + chunk : batch) - b0, h->side));
            }
            return DPOSER_OK;
        };
        const int rc_chunks = chunks();
        DP_CHECK_HIP(hipEventRecord(h->ev_join, h->side));
        DP_CHECK_HIP(hipStreamWaitEvent(st, h->ev_join, 0));
        if (rc_chunks != DPOSER_OK) return rc_chunks;
    }
    // 4. extra joints + landmarks
    // (all three tables NULL: the caller reads tree joints only -- rows [J, J + num_extra + num_landmarks) of `joints` are left unwritten)
    if (h->d.num_extra + h->d.num_landmarks > 0 && (extra_vertex_ids || lmk_tri || lmk_bary)) {
        DP_CHECK_ARG((h->d.num_extra == 0 || extra_vertex_ids) && (h->d.num_landmarks == 0 || (lmk_tri && lmk_bary)), "missing landmark tables");
        ExtraArgs e;
        e.verts = verts; e.extra_ids = extra_vertex_ids; e.lmk_tri = lmk_tri; e.lmk_bary = lmk_bary; e.joints = joints;
        e.ld = (int64_t)n_total * 3; e.J = J; e.n_extra = h->d.num_extra; e.n_lmk = h->d.num_landmarks; e.V = V; e.B = batch;
        const int64_t total = batch * (e.n_extra + e.n_lmk);
        hipLaunchKernelGGL(k_extra_joints, dim3((unsigned)(ceil_div(total, 256) > 4096 ? 4096 : ceil_div(total, 256))), dim3(256), 0, st, e);
        FK_HIP_LAUNCH(hipGetLastError());
    }
    return DPOSER_OK;
}


// ------------------------------------------------------------------------------------------------
// Skinning fused with the temporal term of the motion-denoising loss (run/motion_denoising.py:253-255):
//   temp = mean over (F-1, V) of ||v[t] - v[t+1]||,  d temp / d v[t] = c (u_t - u_{t-1}),  u_t = (v[t] - v[t+1]) / ||v[t] - v[t+1]||.
// The fitting loop needs the vertices for nothing else, so they never reach HBM: a block owns NV x 256 vertices of one run of
// frames of one sequence and walks the frames, the previous frame's vertices in registers -- per frame it reads the pose-blend
// offsets once and writes d verts once (k_skin_x4 + k_md_vert_grad: offsets in, vertices out, vertices in three times through L2,
// d verts out).  Vertices = k_skin_x4's expression, gradient and the per-(frame, 256-vertex block) distance sums = k_md_vert_grad's,
// operation for operation (the file is compiled without FMA contraction): bit-identical to that pair of kernels.
// A run [f0, f1) recomputes the vertices of frames f0 - 1 and f1 (its halo): `nseg` runs per sequence trade 2 / (F / nseg) extra
// skinning work for nseg x the workgroups.
// ------------------------------------------------------------------------------------------------
struct SkinTemporalArgs {
    SkinArgs s;                // (verts unused)
    float* dverts;             // [B][V][3] out
    float* part;               // [B][ceil(V / 256)] out: sum over the block's vertices of ||v[t] - v[t+1]|| (0 for the last frame of a sequence)
    int F, nseg;
    float c;
};
template <int NV, bool VSB> __global__ void __launch_bounds__(256, 4) k_skin_temporal(SkinTemporalArgs a) {
    DP_SKIN_FP
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const SkinArgs& s = a.s;
    const int JA = s.J * 12;
    float* sred = smem + 2 * JA;                                  // [frames of the run][NV][4 waves]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int seq = blockIdx.y / a.nseg, sg = blockIdx.y % a.nseg;
    const int f0 = (int)((int64_t)sg * a.F / a.nseg), f1 = (int)((int64_t)(sg + 1) * a.F / a.nseg);
    const int ts = f0 > 0 ? f0 - 1 : 0, te = f1 < a.F ? f1 : a.F - 1;         // frames whose vertices this block forms
    const int64_t b0 = (int64_t)seq * a.F;
    const int vb = (s.V + 255) >> 8;
    const int vbase = blockIdx.x * (NV * 256) + threadIdx.x;
    int vc[NV];
    f32x4 w4[NV];
    int4 j4[NV];
    float vs[NV][3];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int v = vbase + u * 256;
        vc[u] = v < s.V ? v : s.V - 1;                              // clamp: the tail threads work on a valid vertex and do not store
        w4[u] = *reinterpret_cast<const f32x4*>(s.skin_w + (int64_t)vc[u] * 4);
        j4[u] = *reinterpret_cast<const int4*>(s.skin_idx + (int64_t)vc[u] * 4);
        if (!VSB) {
#pragma unroll
            for (int c = 0; c < 3; ++c) vs[u][c] = s.v_shaped[(int64_t)vc[u] * 3 + c];
        }
    }
    float an[3];                                                   // the next frame's transforms on their way into LDS
    auto load_A = [&](int64_t b) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = threadIdx.x + i * 256;
            an[i] = e < JA ? s.A[b * JA + e] : 0.f;
        }
    };
    auto store_A = [&](float* dst) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int e = threadIdx.x + i * 256;
            if (e < JA) dst[e] = an[i];
        }
    };
    float on[NV][3], vn[NV][3];                                    // the next frame's offsets (and rest shape, if per frame)
    auto load_p = [&](int64_t b) {
        const float* off_row = s.offsets + b * s.ld_off;
#pragma unroll
        for (int u = 0; u < NV; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                on[u][c] = off_row[(int64_t)vc[u] * 3 + c];
                if (VSB) vn[u][c] = s.v_shaped[(b * s.V + vc[u]) * 3 + c];
            }
    };
    load_A(b0 + ts);
    load_p(b0 + ts);
    store_A(smem);
    float pv[NV][3], hold[NV][3];
#pragma unroll
    for (int u = 0; u < NV; ++u)
#pragma unroll
        for (int c = 0; c < 3; ++c) { pv[u][c] = 0.f; hold[u][c] = 0.f; }
    int cur = 0;
#pragma unroll 1
    for (int t = ts; t <= te; ++t) {
        __syncthreads();                                           // sA[cur] complete; every wave is done with sA[cur ^ 1]
        const float* sA = smem + cur * JA;
        float p[NV][3];
#pragma unroll
        for (int u = 0; u < NV; ++u)
#pragma unroll
            for (int c = 0; c < 3; ++c) p[u][c] = (VSB ? vn[u][c] : vs[u][c]) + on[u][c];
        float tr[3] = {0.f, 0.f, 0.f};
        if (s.transl) { tr[0] = s.transl[(b0 + t) * 3]; tr[1] = s.transl[(b0 + t) * 3 + 1]; tr[2] = s.transl[(b0 + t) * 3 + 2]; }
        if (t < te) { load_A(b0 + t + 1); load_p(b0 + t + 1); }    // in flight behind this frame's arithmetic
        const bool out_prev = t > ts && t - 1 >= f0;               // frame t - 1 is one of this block's output frames
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int v = vbase + u * 256;
            const int jj[4] = {j4[u].x, j4[u].y, j4[u].z, j4[u].w};
            float T[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) T[i] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4* Aj = reinterpret_cast<const f32x4*>(sA) + jj[k] * 3;    // 16-byte reads: see k_skin_x4
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const f32x4 row = Aj[r];
#pragma unroll
                    for (int i = 0; i < 4; ++i) T[4 * r + i] += w4[u][k] * row[i];
                }
            }
            float q[3];
            q[0] = T[0] * p[u][0] + T[1] * p[u][1] + T[2] * p[u][2] + T[3] + tr[0];
            q[1] = T[4] * p[u][0] + T[5] * p[u][1] + T[6] * p[u][2] + T[7] + tr[1];
            q[2] = T[8] * p[u][0] + T[9] * p[u][1] + T[10] * p[u][2] + T[11] + tr[2];
            if (t > ts) {
                DP_SKIN_FP_OFF
                // the pair (t - 1, t): forward difference of frame t - 1, backward difference of frame t (k_md_vert_grad's two branches)
                const float ax = pv[u][0] - q[0], ay = pv[u][1] - q[1], az = pv[u][2] - q[2];
                const float ss = ax * ax + ay * ay + az * az;
                const float d = sqrtf(ss);
                const float inv = __builtin_amdgcn_rcpf(d);
                const float gx = ax * inv - hold[u][0], gy = ay * inv - hold[u][1], gz = az * inv - hold[u][2];
                const float invb = __builtin_amdgcn_rsqf(ss);
                hold[u][0] = ax * invb; hold[u][1] = ay * invb; hold[u][2] = az * invb;
                if (out_prev) {
                    if (v < s.V) {
                        float* o = a.dverts + ((b0 + t - 1) * s.V + v) * 3;
                        o[0] = a.c * gx; o[1] = a.c * gy; o[2] = a.c * gz;
                    }
                    float dsum = v < s.V ? d : 0.f;
#pragma unroll
                    for (int sh = 32; sh >= 1; sh >>= 1) dsum += __shfl_xor(dsum, sh);
                    if (lane == 0) sred[((t - 1 - f0) * NV + u) * 4 + wave] = dsum;
                }
            }
            pv[u][0] = q[0]; pv[u][1] = q[1]; pv[u][2] = q[2];
            __builtin_amdgcn_sched_barrier(0);                     // one vertex at a time: hoisting the 48 LDS reads of all NV vertices spills
        }
        if (t < te) store_A(smem + (cur ^ 1) * JA);
        cur ^= 1;
    }
    if (f1 == a.F) {                                               // last frame of the sequence: backward difference only, no distance
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int v = vbase + u * 256;
            if (v < s.V) {
                float* o = a.dverts + ((b0 + a.F - 1) * s.V + v) * 3;
                o[0] = a.c * (0.f - hold[u][0]); o[1] = a.c * (0.f - hold[u][1]); o[2] = a.c * (0.f - hold[u][2]);
            }
            if (lane == 0) sred[((a.F - 1 - f0) * NV + u) * 4 + wave] = 0.f;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (f1 - f0) * NV; i += 256) {
        const int vblk = blockIdx.x * NV + (i % NV);
        const float* r = sred + i * 4;
        if (vblk < vb) a.part[(b0 + f0 + i / NV) * vb + vblk] = r[0] + r[1] + r[2] + r[3];
    }
}

// runs per sequence: enough workgroups for two rounds of the chip, runs of at least 10 frames (halo <= 20 %)
static int skin_temporal_nseg(int64_t n_seq, int F, int vchunks) {
    const char* e = getenv("DPOSER_SKIN_TEMPORAL_NSEG");          // (A/B and tests; read per call)
    const int forced = e ? atoi(e) : 0;
    if (forced > 0) return forced < F / 2 ? forced : (F / 2 > 0 ? F / 2 : 1);
    int nseg = 1;
    while (n_seq * vchunks * nseg < 2048 && F / (nseg + 1) >= 10) ++nseg;
    return nseg;
}

extern "C" int dposer_lbs_forward_temporal_grad(dposer_body_t h, void* ws, const void* posedirs_packed, const float* const* pose_segments_host,
                                                const int32_t* segment_joints_host, int32_t num_segments, const float* j_rest,
                                                int32_t j_rest_batched, const float* v_shaped, int32_t v_shaped_batched, const int32_t* skin_idx,
                                                const float* skin_w, int32_t skin_k, const float* transl, int64_t frames_per_sequence, float scale,
                                                float* d_verts, float* dist_part, float* joints, int64_t batch, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(v_shaped && skin_idx && skin_w && d_verts && dist_part, "null argument");
    DP_CHECK_ARG(skin_k == 4, "four skinning weights per vertex (the ELL width of every SMPL-family asset)");
    DP_CHECK_ARG(frames_per_sequence >= 2 && batch % frames_per_sequence == 0, "batch must be whole sequences of >= 2 frames");
    DP_CHECK_ARG(frames_per_sequence <= 4096 && batch / frames_per_sequence <= 16384, "sequence too long / too many sequences");
    float *A = nullptr, *offsets = nullptr;
    DP_TRY(lbs_forward_front(h, ws, posedirs_packed, pose_segments_host, segment_joints_host, num_segments, j_rest, j_rest_batched, transl, joints,
                             batch, stream, &A, &offsets));
    hipStream_t st = (hipStream_t)stream;
    const int J = h->d.num_joints, V = h->d.num_vertices, F = (int)frames_per_sequence;
    const int64_t n_seq = batch / F;
    SkinTemporalArgs a;
    a.s.offsets = offsets; a.s.ld_off = lbs_cpad(V); a.s.v_shaped = v_shaped; a.s.v_shaped_batched = v_shaped_batched; a.s.A = A;
    a.s.skin_idx = skin_idx; a.s.skin_w = skin_w; a.s.K = skin_k; a.s.J = J; a.s.V = V; a.s.transl = transl; a.s.verts = nullptr;
    a.dverts = d_verts; a.part = dist_part; a.F = F; a.c = scale;
    constexpr int NV = 2;
    const int vchunks = (int)ceil_div(V, 256 * NV);
    a.nseg = skin_temporal_nseg(n_seq, F, vchunks);
    const int maxrun = (int)ceil_div(F, a.nseg) + 1;
    const size_t lds = (size_t)(2 * J * 12 + maxrun * NV * 4) * sizeof(float);
    DP_CHECK_ARG(lds <= 64 * 1024, "sequence too long for one run per workgroup");
    const dim3 grid((unsigned)vchunks, (unsigned)(n_seq * a.nseg));
    if (v_shaped_batched) hipLaunchKernelGGL((k_skin_temporal<NV, true>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_skin_temporal<NV, false>), grid, dim3(256), lds, st, a);
    FK_HIP_LAUNCH(hipGetLastError());
    return DPOSER_OK;
}


// ------------------------------------------------------------------------------------------------
// Backward of linear blend skinning + forward kinematics (d verts, d joints -> d pose, d rest joints, d v_shaped)
// needed by the fitting loops that differentiate through the body model (run/motion_denoising.py:217-218,255-267).
// ------------------------------------------------------------------------------------------------
// Gradients of the vertex-selected extra joints and of the barycentric landmarks (smplx VertexJointSelector / vertices2landmarks:
// rows J ... of the joint output are gathers / 3-term combinations of vertices) folded into the vertex gradient WITHOUT touching the
// caller's d_verts: k_fold_rows writes the corrected rows of the <= n_extra + 3 n_landmarks vertices they touch into a compact
// [B][U][3] array (d_verts row + sum over the vertex's entries, in table order: deterministic), and every kernel below that reads a
// vertex gradient goes through VertGrad::row(), which picks the corrected row where the vertex has a slot.  (Round 3-4: the Python
// wrapper did this with torch.matmul -- a hipBLASLt launch -- plus index / index_put kernels, and wrote into autograd's incoming
// gradient in place.)
struct FoldArgs {
    const int32_t* vslot;      // [V]: slot u of vertex v in the compact array, or -1; null: no fold
    const int32_t* uniq;       // [U] vertex of slot u
    const int32_t* ptr;        // [U + 1] entries of slot u
    const int32_t* row;        // [n] row of d_joints the entry gathers from (absolute: >= J)
    const float* w;            // [n] its weight (1 for an extra joint, a barycentric coordinate for a landmark)
    int U;
};
struct VertGrad {
    const float* dverts;       // [B][V][3] (the caller's, read-only)
    const int32_t* vslot;      // or null
    const float* fixed;        // [B][U][3] corrected rows
    int U;
    __device__ __forceinline__ const float* row(int64_t b, int v, int V) const {
        if (vslot) {
            const int u = vslot[v];
            if (u >= 0) return fixed + (b * U + u) * 3;
        }
        return dverts + (b * V + v) * 3;
    }
};
__global__ void __launch_bounds__(64) k_fold_rows(FoldArgs f, const float* __restrict__ dverts, const float* __restrict__ djoints, int64_t ld_j,
                                                  float* __restrict__ fixed, int V, int64_t B) {
    const int u = blockIdx.x * 64 + threadIdx.x;
    const int64_t b = blockIdx.y;
    if (u >= f.U || b >= B) return;
    const float* dv = dverts + (b * V + f.uniq[u]) * 3;
    float acc[3] = {dv[0], dv[1], dv[2]};
    for (int e = f.ptr[u]; e < f.ptr[u + 1]; ++e) {
        const float* dj = djoints + b * ld_j + (int64_t)f.row[e] * 3;
        const float w = f.w[e];
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] += w * dj[k];
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) fixed[(b * f.U + u) * 3 + k] = acc[k];
}

struct SkinBwdArgs {
    VertGrad vg;
    const float* dverts;       // [B][V][3]
    const float* offsets;      // [B][ld_off] (forward workspace)
    int64_t ld_off;
    const float* v_shaped;
    int v_shaped_batched;
    const float* A;            // [B][J][12]
    const int32_t* skin_idx;
    const float* skin_w;
    int K, J, V;
    float* vp;                 // [B][V][3] out: posed-blend vertices (v_shaped + offsets)
    float* dvp;                // [B][V][3] out: d loss / d v_posed, or null
    float* doff_ft;            // FT32 [Bpad][Cpad] out (same values, GEMM operand layout) ...
    __bf16* doff_hi;           // ... or, for the bf16 x 3 blend GEMMs, the two bf16 terms as FT bf16 [Bpad][Cpad] each (doff_ft null)
    __bf16* doff_lo;
    int Cpad;
};
__global__ void __launch_bounds__(256) k_skin_bwd(SkinBwdArgs a) {
    extern __shared__ float sA[];       // [J][12] transforms of this pose, then 2 x [768] staging of the block's d_off / vp values
    float* stage = sA + a.J * 12;
    float* stage_vp = stage + 768;
    const int64_t b = blockIdx.y;
    for (int i = threadIdx.x; i < a.J * 12; i += 256) sA[i] = a.A[b * a.J * 12 + i];
    __syncthreads();
    for (int v0 = blockIdx.x * 256; v0 < a.V; v0 += gridDim.x * 256) {
        const int v = v0 + threadIdx.x;
        float g[3] = {0.f, 0.f, 0.f}, p[3] = {0.f, 0.f, 0.f};
        if (v < a.V) {
            const float* vs = a.v_shaped + (a.v_shaped_batched ? b * a.V * 3 : 0) + (int64_t)v * 3;
            const float* off = a.offsets + b * a.ld_off + (int64_t)v * 3;
            const float* dv = a.vg.row(b, v, a.V);
            float T[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) T[i] = 0.f;
            for (int k = 0; k < a.K; ++k) {
                const float w = a.skin_w[(int64_t)v * a.K + k];
                const float* Aj = sA + a.skin_idx[(int64_t)v * a.K + k] * 12;
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 3; ++c) T[3 * r + c] += w * Aj[4 * r + c];
            }
            const float dx = dv[0], dy = dv[1], dz = dv[2];
#pragma unroll
            for (int c = 0; c < 3; ++c) g[c] = T[c] * dx + T[3 + c] * dy + T[6 + c] * dz;     // T_R^T dv
#pragma unroll
            for (int c = 0; c < 3; ++c) p[c] = vs[c] + off[c];
        }
        // All three outputs leave through LDS: the block's 768 consecutive coordinates are staged, then stored as
        // consecutive dwords (vp, dvp: [B][V][3], rows not 16-byte aligned) and as whole 16-byte quads (the FT32 GEMM
        // operand).  Per-thread 3 x 4-byte stores at a 12-byte stride cost 1.8x the HBM write traffic (PMC WRITE_SIZE).
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 3; ++c) { stage[threadIdx.x * 3 + c] = g[c]; stage_vp[threadIdx.x * 3 + c] = p[c]; }
        __syncthreads();
        const int64_t row = (b * a.V + v0) * 3;
        const int nval = (a.V - v0 < 256 ? a.V - v0 : 256) * 3;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int k = i * 256 + threadIdx.x;
            if (k < nval) {
                a.vp[row + k] = stage_vp[k];
                if (a.dvp) a.dvp[row + k] = stage[k];      // gradient w.r.t. v_shaped: only when the caller wants it (515 MB at 4096 poses)
            }
        }
        if (a.doff_ft) {
            if (threadIdx.x < 192) {
                const int kq = v0 * 3 + threadIdx.x * 4;                                     // first coordinate of this quad
                if (kq < a.Cpad) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(stage + threadIdx.x * 4);
                    *reinterpret_cast<f32x4*>(a.doff_ft + FT<float>::index(b, kq, a.Cpad)) = q;
                }
            }
        } else if (threadIdx.x < 96) {
            const int k8 = v0 * 3 + threadIdx.x * 8;                                         // first coordinate of this 8-element chunk
            if (k8 < a.Cpad) {
                __bf16 hi[8], lo[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float x = stage[threadIdx.x * 8 + e];
                    hi[e] = (__bf16)x;
                    lo[e] = (__bf16)(x - (float)hi[e]);
                }
                *reinterpret_cast<u32x4*>(a.doff_hi + FT<__bf16>::index(b, k8, a.Cpad)) = *reinterpret_cast<u32x4*>(hi);
                *reinterpret_cast<u32x4*>(a.doff_lo + FT<__bf16>::index(b, k8, a.Cpad)) = *reinterpret_cast<u32x4*>(lo);
            }
        }
    }
}

// dA[b][j] = sum over the vertices skinned to joint j of w_vj * (dv_v (x) [vp_v ; 1])
// One block per pose streams the pose's dverts / vp rows ONCE, a chunk of <= 256 vertices at a time through LDS (coalesced),
// together with the chunk's slice of the skinning lists (re-cut by chunk on the host, sorted by joint inside a chunk: one
// contiguous run of (weight, local vertex) pairs).  Joint j is served by 4 lanes (lane q accumulates column q of [vp ; 1] for
// the three rows); partial sums stay in registers over the whole pose, in a fixed order => deterministic.
// (Round 1-2 version: one block per (pose, joint) gathering 2 x 12 B per list entry from L2 -- 4.1 GB of gathers per 4096 poses,
//  1.15 ms even with an XCD-aware block order.  A first streaming version that walked the caller's CSR with a per-lane cursor was
//  latency-bound on its two dependent global loads per entry: 1.53 ms.)
constexpr int JL_MAXV = 256, JL_MAXE = 2048;
struct JointBwdArgs {
    VertGrad vg;
    const float* dverts;
    const float* vp;
    const int32_t* vstart;     // [chunks + 1]
    const int32_t* cptr;       // [chunks][J + 1]
    const int32_t* cfirst;     // [chunks + 1]
    const float2* entry;       // [nnz]
    float* dA;                 // [B][J][12]
    int J, V, chunks;
    int64_t B;
};
__global__ void __launch_bounds__(256) k_skin_bwd_joints(JointBwdArgs a) {
    __shared__ f32x4 sdv[JL_MAXV];          // (dv.x, dv.y, dv.z, -)
    __shared__ f32x4 svp[JL_MAXV];          // (vp.x, vp.y, vp.z, 1)
    __shared__ float2 sent[JL_MAXE];
    __shared__ int sptr[68];
    const int64_t b = blockIdx.x;
    const int j = threadIdx.x >> 2, q = threadIdx.x & 3;
    const bool on = j < a.J;
    float acc[3] = {0.f, 0.f, 0.f};
    const float* dvb = a.dverts + b * a.V * 3;
    const float* vpb = a.vp + b * a.V * 3;
    for (int c = 0; c < a.chunks; ++c) {
        const int v0 = a.vstart[c], nv = a.vstart[c + 1] - v0;
        const int e0 = a.cfirst[c], ne = a.cfirst[c + 1] - e0;
        for (int k = threadIdx.x; k < nv * 3; k += 256) {
            const int l = k / 3, r = k - 3 * l;
            reinterpret_cast<float*>(&sdv[l])[r] = a.vg.vslot ? a.vg.row(b, v0 + l, a.V)[r] : dvb[(int64_t)v0 * 3 + k];
            reinterpret_cast<float*>(&svp[l])[r] = vpb[(int64_t)v0 * 3 + k];
        }
        if ((int)threadIdx.x < nv) reinterpret_cast<float*>(&svp[threadIdx.x])[3] = 1.0f;
        for (int k = threadIdx.x; k < ne; k += 256) sent[k] = a.entry[e0 + k];
        if ((int)threadIdx.x <= a.J) sptr[threadIdx.x] = a.cptr[c * (a.J + 1) + threadIdx.x];
        __syncthreads();
        if (on) {
            const int i1 = sptr[j + 1];
            int i = sptr[j];
            // four entries at a time: their three dependent LDS reads each (entry -> dv, vp) overlap; the sums stay in list order
            for (; i + 4 <= i1; i += 4) {
                float2 en[4];
                f32x4 d[4];
                float h[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) en[u] = sent[i + u];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int l = __float_as_int(en[u].y);
                    d[u] = sdv[l];
                    h[u] = reinterpret_cast<const float*>(&svp[l])[q];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 3; ++r) acc[r] += en[u].x * d[u][r] * h[u];
            }
            for (; i < i1; ++i) {
                const float2 en = sent[i];
                const int l = __float_as_int(en.y);
                const f32x4 d = sdv[l];
                const float h = reinterpret_cast<const float*>(&svp[l])[q];
#pragma unroll
                for (int r = 0; r < 3; ++r) acc[r] += en.x * d[r] * h;
            }
        }
        __syncthreads();
    }
    if (on) {
#pragma unroll
        for (int r = 0; r < 3; ++r) a.dA[(b * a.J + j) * 12 + 4 * r + q] = acc[r];
    }
}
// Small batches: one block per (pose, joint) gathering its list entries -- enough parallelism when there are few poses (the
// streaming kernel above gives a pose ONE block that walks 41 chunks: ~300 us however few poses there are; this one takes 20 us
// at 60 poses and loses above ~1500, where its 2 x 12 B gathers per entry from L2 become the bound: 1152 vs 732 us at 4096).
struct JointGatherArgs {
    VertGrad vg;
    const float* dverts;
    const float* vp;
    const int32_t* jptr;       // [J+1] CSR by joint
    const int32_t* jvidx;      // [nnz]
    const float* jw;           // [nnz]
    float* dA;                 // [B][J][12]
    int J, V;
    int64_t B;
};
__global__ void __launch_bounds__(128) k_skin_bwd_joints_gather(JointGatherArgs a) {
    __shared__ float red[128][13];
    // XCD-aware order (hardware XCD = linear block id % 8): all J joints of one pose run back to back on ONE XCD, so the
    // pose's dverts / vp rows come from HBM once and from that XCD's L2 afterwards
    const int64_t q = blockIdx.x >> 3;
    const int j = (int)(q % a.J);
    const int64_t b = (q / a.J) * 8 + (blockIdx.x & 7);
    if (b >= a.B) return;
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;
    for (int e = a.jptr[j] + threadIdx.x; e < a.jptr[j + 1]; e += 128) {
        const int v = a.jvidx[e];
        const float w = a.jw[e];
        const float* dv = a.vg.row(b, v, a.V);
        const float* p = a.vp + (b * a.V + v) * 3;
        const float h[4] = {p[0], p[1], p[2], 1.0f};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[4 * r + c] += w * dv[r] * h[c];
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) red[threadIdx.x][i] = acc[i];
    __syncthreads();
    for (int s = 64; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int i = 0; i < 12; ++i) red[threadIdx.x][i] += red[threadIdx.x + s][i];
        __syncthreads();
    }
    if (threadIdx.x < 12) a.dA[(b * a.J + j) * 12 + threadIdx.x] = red[0][threadIdx.x];
}
// Skinning backward in ONE streaming pass per pose (round 3): k_skin_bwd + k_skin_bwd_joints read d_verts twice and pass v_posed
// through HBM (516 MB written and read back at 4096 SMPL-X poses; PMC: 2.8 + 1.05 GB for the pair).  One block per pose walks the
// regular 256-vertex chunks; per chunk
//   A  thread = vertex: T = sum_k w_k A[j_k], g = T_R^T dv, p = v_shaped + offsets  ->  LDS (dv, p, g) + the chunk's entry / segment tables
//   B  g leaves as the two bf16 terms of the blend GEMM's operand (FT) [and as d v_posed if asked for]; the 64 lane quads take the
//      chunk's segments (12 ... 32 list entries each, one per quad: see dposer_lbs_prepare_joint_lists) and leave partial sums in LDS
//   C  the 4 lanes that own joint j add its segments' partials, in table order, to their running dA[j] (registers, whole pose)
// Two barriers per chunk; the next chunk's global loads are issued before B.  Summation order: entries inside a segment in list
// order, segments in order, chunks in order: deterministic (and different from k_skin_bwd_joints' by rounding only).
constexpr int FUSED_MAXSEG = 128, FUSED_MAXE = 1024;
struct SkinBwdFusedArgs {
    VertGrad vg;
    const float* dverts;       // [B][V][3]
    const float* offsets;      // [B][ld_off]
    int64_t ld_off;
    const float* v_shaped;
    int v_shaped_batched;
    const float* A;            // [B][J][12]
    const int32_t* skin_idx;   // [V][4]
    const float* skin_w;       // [V][4]
    int J, V;
    float* dvp;                // [B][V][3] or null
    __bf16* doff_hi;           // FT bf16 [Bpad][Cpad]
    __bf16* doff_lo;
    int Cpad;
    const int32_t* cfirst;     // [chunks + 1]
    const float2* entry;       // [nnz]
    const int2* seg;           // [chunks][FUSED_MAXSEG]
    const int32_t* nseg;       // [chunks]
    const int32_t* jseg;       // [chunks][J]
    float* dA;                 // [B][J][12]
    int chunks;
    int64_t B;
};
__global__ void __launch_bounds__(256) k_skin_bwd_fused(SkinBwdFusedArgs a) {
    __shared__ __attribute__((aligned(16))) float sA[64 * 12];
    __shared__ f32x4 sP[4 * 257];           // [q][vertex] planes of 257 records (conflict-free 16-byte writes; the quad's four reads land 4 banks apart): per vertex and q the products dv_r * [p ; 1]_q (r = 0..2, one 16-byte record): formed ONCE per vertex by
                                            // its own thread (a vertex sits in four joint lists; per list entry a lane does one 16-byte read, 3 FMAs)
    __shared__ float stage[768];            // g = d loss / d v_posed of the chunk, coordinate-major
    __shared__ float2 sent[FUSED_MAXE];
    __shared__ int2 sseg[FUSED_MAXSEG];
    __shared__ float part[FUSED_MAXSEG][4][3];
    // block -> pose: the FT operand keeps 4 consecutive poses in one 128-byte line (32 B each), and block i runs on XCD i % 8 with its
    // own L2 -- blocks 32 g + 8 i + x (i < 4) take the poses 32 g + 4 x + i, so that the four writers of a line share an L2 and are
    // dispatched back to back (the PMC counted 1043 MB written for 516 MB of operand with the identity map)
    const int64_t blk = blockIdx.x;
    const int64_t b = (blk < (a.B & ~(int64_t)31)) ? (blk & ~(int64_t)31) + 4 * (blk & 7) + ((blk >> 3) & 3) : blk;
    const int tid = threadIdx.x;
    for (int i = tid; i < a.J * 12; i += 256) sA[i] = a.A[b * a.J * 12 + i];
    const float* vs_row = a.v_shaped + (a.v_shaped_batched ? b * a.V * 3 : 0);
    const float* off_row = a.offsets + b * a.ld_off;
    const float* dv_row = a.dverts + b * a.V * 3;
    const int jq = tid >> 2, q = tid & 3;                  // joint owner / segment worker: quad and lane inside it
    float tot[3] = {0.f, 0.f, 0.f};
    // the chunk's per-vertex loads (clamped: tail threads load a valid vertex and contribute zeros)
    float dv[3], pp[3];
    f32x4 w4;
    int4 j4;
    auto load_vertex = [&](int c) __attribute__((always_inline)) {
        const int v = c * 256 + tid;
        const int vc = v < a.V ? v : a.V - 1;
        const float* dvr = a.vg.vslot ? a.vg.row(b, vc, a.V) : dv_row + (int64_t)vc * 3;     // (corrected row where a landmark / extra joint touches the vertex)
#pragma unroll
        for (int k = 0; k < 3; ++k) { dv[k] = dvr[k]; pp[k] = vs_row[(int64_t)vc * 3 + k] + off_row[(int64_t)vc * 3 + k]; }
        w4 = *reinterpret_cast<const f32x4*>(a.skin_w + (int64_t)vc * 4);
        j4 = *reinterpret_cast<const int4*>(a.skin_idx + (int64_t)vc * 4);
    };
    // ... and its tables (entries, segment bounds, this quad's joint): registers, one chunk ahead like the vertex data; the per-chunk
    // counts sit in LDS (a dependent scalar load -> global load -> LDS chain per chunk was most of the first version's time)
    __shared__ int scf[64], sns[64];
    for (int i = tid; i <= a.chunks && i < 64; i += 256) scf[i] = a.cfirst[i];
    for (int i = tid; i < a.chunks && i < 64; i += 256) sns[i] = a.nseg[i];
    float2 en_r[4];
    int2 sg_r;
    int js_r;
    auto load_tables = [&](int c) __attribute__((always_inline)) {
        const int e0 = scf[c], ne = scf[c + 1] - e0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = tid + 256 * i;
            en_r[i] = a.entry[e0 + (k < ne ? k : 0)];
        }
        sg_r = a.seg[(int64_t)c * FUSED_MAXSEG + (tid < FUSED_MAXSEG ? tid : 0)];
        js_r = a.jseg[(int64_t)c * a.J + (jq < a.J ? jq : 0)];
    };
    __syncthreads();
    load_vertex(0);
    load_tables(0);
    for (int c = 0; c < a.chunks; ++c) {
        const int v0 = c * 256, v = v0 + tid;
        const int ne = scf[c + 1] - scf[c], ns = sns[c];
        // ---- A
        {
            const bool live = v < a.V;
            const int jj[4] = {j4.x, j4.y, j4.z, j4.w};
            float T[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) T[i] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4* Aj = reinterpret_cast<const f32x4*>(sA) + jj[k] * 3;    // 16-byte reads: see k_skin_x4
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const f32x4 row = Aj[r];
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) T[3 * r + cc] += w4[k] * row[cc];
                }
            }
            const float dx = live ? dv[0] : 0.f, dy = live ? dv[1] : 0.f, dz = live ? dv[2] : 0.f;
            const float hv[4] = {pp[0], pp[1], pp[2], 1.0f};
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                f32x4 o;
                o[0] = dx * hv[qq]; o[1] = dy * hv[qq]; o[2] = dz * hv[qq]; o[3] = 0.f;
                sP[qq * 257 + tid] = o;
            }
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) stage[tid * 3 + cc] = T[cc] * dx + T[3 + cc] * dy + T[6 + cc] * dz;      // T_R^T dv
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (tid + 256 * i < ne) sent[tid + 256 * i] = en_r[i];
        if (tid < ns) sseg[tid] = sg_r;
        const int js = (jq < a.J) ? js_r : 0;                               // this quad's joint: (first segment | count << 16)
        if (c + 1 < a.chunks) { load_vertex(c + 1); load_tables(c + 1); }   // in flight across B and C
        __syncthreads();
        // ---- B: outputs of the vertex half
        {
            const int nval = (a.V - v0 < 256 ? a.V - v0 : 256) * 3;
            if (a.dvp) {
                const int64_t row = (b * a.V + v0) * 3;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int k = i * 256 + tid;
                    if (k < nval) a.dvp[row + k] = stage[k];
                }
            }
            if (tid < 96) {
                const int k8 = v0 * 3 + tid * 8;                                     // first coordinate of this 8-element chunk
                if (k8 < a.Cpad) {
                    __bf16 hi[8], lo[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float x = (tid * 8 + e < nval) ? stage[tid * 8 + e] : 0.f;
                        hi[e] = (__bf16)x;
                        lo[e] = (__bf16)(x - (float)hi[e]);
                    }
                    *reinterpret_cast<u32x4*>(a.doff_hi + FT<__bf16>::index(b, k8, a.Cpad)) = *reinterpret_cast<u32x4*>(hi);
                    *reinterpret_cast<u32x4*>(a.doff_lo + FT<__bf16>::index(b, k8, a.Cpad)) = *reinterpret_cast<u32x4*>(lo);
                }
            }
        }
        // ---- B: segments
        for (int sgi = jq; sgi < ns; sgi += 64) {
            const int2 se = sseg[sgi];
            float acc[3] = {0.f, 0.f, 0.f};
            // four entries per iteration, ONE entry read per lane: lane q of the quad fetches entry i + q, DPP quad broadcasts hand every
            // entry to all four lanes (the LDS pipe is this kernel's bound; a per-lane read of the same entry by all four lanes costs an LDS
            // instruction each).  Entries past the segment's end carry weight 0 and vertex 0.  Sums in list order.
            for (int i = se.x; i < se.y; i += 4) {
                const bool in = i + q < se.y;
                const float2 mine = sent[in ? i + q : se.x];
                const int wbits = in ? __float_as_int(mine.x) : 0, lbits = in ? __float_as_int(mine.y) : 0;
                f32x4 pr[4];
                float wu[4];
#define DP_QUAD_BCAST(U)                                                                              \
    wu[U] = __int_as_float(__builtin_amdgcn_mov_dpp(wbits, U * 0x55, 0xf, 0xf, false));                 \
    pr[U] = sP[q * 257 + __builtin_amdgcn_mov_dpp(lbits, U * 0x55, 0xf, 0xf, false)]      /* quad_perm: [U, U, U, U] */
                DP_QUAD_BCAST(0); DP_QUAD_BCAST(1); DP_QUAD_BCAST(2); DP_QUAD_BCAST(3);
#undef DP_QUAD_BCAST
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 3; ++r) acc[r] += wu[u] * pr[u][r];
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) part[sgi][q][r] = acc[r];
        }
        __syncthreads();
        // ---- C
        {
            const int f = js & 0xffff, n = js >> 16;
            for (int i = 0; i < n; ++i)
#pragma unroll
                for (int r = 0; r < 3; ++r) tot[r] += part[f + i][q][r];
        }
    }
    if (jq < a.J) {
#pragma unroll
        for (int r = 0; r < 3; ++r) a.dA[(b * a.J + jq) * 12 + 4 * r + q] = tot[r];
    }
}

// ------------------------------------------------------------------------------------------------
// Skinning backward with the joint reduction on the MATRIX PIPE (round 5).  k_skin_bwd_fused is LDS-bound: every one of a chunk's
// ~1024 (vertex, joint) list entries costs four 16-byte LDS reads (65 KB per pose and chunk, a third of them bank conflicts).  The
// reduction  dA[j][4 r + q] = sum_v W[v][j] * (dv_r [p ; 1]_q)  is a GEMM with a sparse factor -- D [64 joints][12] = W^T [64][256] P [256][12] per
// pose and chunk -- and small enough for the dense form: 8 k-steps x 4 joint tiles of v_mfma_f32_16x16x32_bf16, three products per term
// (bf16 hi / lo planes of both factors, fp32 accumulate: the arithmetic of the blend GEMMs).  A workgroup owns G consecutive poses and
// walks the chunks; wave w reduces the chunk's vertices [64 w, 64 w + 64) for all 64 joints: its slice of W^T -- 16 fragments, prepared in
// lane order by dposer_lbs_prepare_joint_lists -- is loaded from L2 once per chunk and serves the G poses from registers; P is formed
// once per vertex by the vertex's thread (as before), leaves as 24 two-byte LDS writes and is read back ONCE (four 16-byte reads per
// lane).  The running dA of every pose stays in accumulator registers over the whole walk; the four waves' partial sums meet in LDS at
// the end.  One barrier per pose and chunk (P and the d v_posed stage are double-buffered).  Deterministic; differs from
// k_skin_bwd_fused by the rounding of the split products (~1e-6 relative).
// ------------------------------------------------------------------------------------------------
constexpr int SBM_PSTRIDE = 72;                  // bf16 per row of a wave's P plane: 64 vertices + 8 (144 B: the 16 rows of a fragment read land 4 banks apart)
constexpr int SBM_PLANE = 16 * SBM_PSTRIDE;      // 12 product rows + 4 zero rows
struct SkinBwdMfmaArgs {
    VertGrad vg;
    const float* dverts;       // [B][V][3]
    const float* offsets;      // [B][ld_off]
    int64_t ld_off;
    const float* v_shaped;
    int v_shaped_batched;
    const float* A;            // [B][J][12]
    const int32_t* skin_idx;   // [V][4]
    const float* skin_w;       // [V][4]
    int J, V;
    float* dvp;                // [B][V][3] or null
    __bf16* doff_hi;           // FT bf16 [Bpad][Cpad]
    __bf16* doff_lo;
    int Cpad;
    const bf16x8* wfrag;       // [chunks][4 waves][4 joint tiles][2 k-steps][hi | lo][64 lanes]
    float* dA;                 // [B][J][12]
    int chunks;
    int64_t B;
    // TMP (motion-denoising loop, run/motion_denoising.py:253-255): the vertex gradient is NOT read -- it is the temporal term's,
    //   d/dv[t] of w/((F-1)V) sum_{t,v} ||v[t] - v[t+1]|| = c (u_t - u_{t-1}),  u_t = (v[t] - v[t+1]) / ||v[t] - v[t+1]||,
    // formed here from the skinned vertices of the pose itself and of its two neighbouring frames (k_skin_x4's expression for the
    // vertices, k_md_vert_grad's for the gradient and the distance sums, operation for operation: the bits of that pair of kernels).
    int F;                     // frames per sequence (the poses of a batch of sequences are consecutive)
    float c;                   // weight / ((F - 1) V)
    float* part4;              // [B][chunks][4]: per wave, the sum over its 64 vertices of ||v[t] - v[t+1]|| (0 for a sequence's last frame)
};
template <int G, bool VSB, bool TMP = false> __global__ void __launch_bounds__(256, 2) k_skin_bwd_mfma(SkinBwdMfmaArgs a) {
    constexpr int NQ = TMP ? G + 2 : G;             // pose slots: TMP adds the frame in front of the workgroup's poses (slot 0) and the one behind (slot G + 1)
    constexpr int Q0 = TMP ? 1 : 0;                 // slot of the workgroup's first own pose
    __shared__ __attribute__((aligned(16))) float sA[NQ][64 * 12];
    __shared__ __attribute__((aligned(16))) __bf16 sP[4][2][SBM_PLANE];     // [wave][hi | lo][product row][the wave's 64 vertices]
    __shared__ __attribute__((aligned(16))) float stage[4][G][192];         // [wave][pose]: g = d loss / d v_posed of the wave's vertices, coordinate-major
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t b0 = (int64_t)blockIdx.x * G;
    // pose of slot q (clamped to the batch: a clamped slot is never used -- own poses beyond the batch are masked, the neighbour
    // slots of a sequence's first / last frame too)
    auto slot_pose = [&](int q) __attribute__((always_inline)) -> int64_t {
        int64_t b = b0 + q - Q0;
        b = b < 0 ? 0 : b;
        return b < a.B ? b : (TMP ? a.B - 1 : b0);
    };
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int64_t b = slot_pose(q);
        for (int i = tid; i < a.J * 12; i += 256) sA[q][i] = a.A[b * a.J * 12 + i];
    }
    for (int i = tid; i < 4 * 2 * SBM_PLANE / 8; i += 256) reinterpret_cast<u32x4*>(&sP[0][0][0])[i] = u32x4{0u, 0u, 0u, 0u};      // (rows 12 ... 15 stay zero)
    f32x4 acc[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[g][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // per-vertex data of (pose, chunk): dv and the rest + pose-blend position of this thread's vertex, loaded a whole chunk ahead (the
    // set of pose g is refilled for chunk c + 1 as soon as (g, c) has consumed it: G iterations of lead -- with one iteration of lead
    // every pose and chunk waited for a full HBM round trip, 70 % of the wave cycles)
    float dvn[TMP ? 1 : G][3], ofn[NQ][3], vsn[VSB ? NQ : 1][3];
    // slot of this thread's vertex in the corrected-row array (extra joints / landmarks folded in: VertGrad), for the chunk whose pose data
    // is being prefetched: the lookup is pose-independent and runs TWO chunks ahead -- as part of every pose's load (vg.row) it put a
    // dependent load in front of every dv load and a full vmcnt(0) drain into every iteration
    int slot_pf = -1, slot_nn = -1;              // of chunk c + 1 (used by the prefetches issued during chunk c) / of chunk c + 2
    float vs_pf[3] = {0.f, 0.f, 0.f}, vs_nn[3] = {0.f, 0.f, 0.f};      // rest-shape position of the vertex, same schedule (unless it is per pose)
    // (every load below is unconditional and its address a select: a branch around a load makes hipcc wait for it at the join -- the
    //  first version's `vslot ? row() : ...` and `v_shaped_batched ? ... : ...` drained the whole prefetch queue in every iteration)
    const int32_t* vslot = a.vg.vslot ? a.vg.vslot : a.skin_idx;          // (no fold: any readable table, the value is ignored)
    const bool has_fold = a.vg.vslot != nullptr;
    const float* fixed = has_fold ? a.vg.fixed : a.dverts;
    auto load_slot = [&](int c) __attribute__((always_inline)) {
        const int v = c * 256 + tid;
        const int vc = v < a.V ? v : a.V - 1;
        const int sl = vslot[vc];
        slot_nn = has_fold ? sl : -1;
        if constexpr (!VSB) {
#pragma unroll
            for (int k = 0; k < 3; ++k) vs_nn[k] = a.v_shaped[(int64_t)vc * 3 + k];
        }
    };
    auto load_pose = [&](int c, int g) __attribute__((always_inline)) {       // g: pose slot
        const int v = c * 256 + tid;
        const int vc = v < a.V ? v : a.V - 1;
        const int64_t b = slot_pose(g);
        const float* off = a.offsets + b * a.ld_off + (int64_t)vc * 3;
        if constexpr (!TMP) {
            const float* plain = a.dverts + (b * a.V + vc) * 3;
            const float* corr = fixed + (b * a.vg.U + (slot_pf >= 0 ? slot_pf : 0)) * 3;
            const float* dvr = slot_pf >= 0 ? corr : plain;
#pragma unroll
            for (int k = 0; k < 3; ++k) dvn[g][k] = dvr[k];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) ofn[g][k] = off[k];
        if constexpr (VSB) {
#pragma unroll
            for (int k = 0; k < 3; ++k) vsn[g][k] = a.v_shaped[(b * a.V + vc) * 3 + k];
        }
    };
    f32x4 w4n;
    int4 j4n;
    auto load_chunk = [&](int c) __attribute__((always_inline)) {
        const int v = c * 256 + tid;
        const int vc = v < a.V ? v : a.V - 1;
        w4n = *reinterpret_cast<const f32x4*>(a.skin_w + (int64_t)vc * 4);
        j4n = *reinterpret_cast<const int4*>(a.skin_idx + (int64_t)vc * 4);
    };
    load_chunk(0);
    load_slot(0);
    slot_pf = slot_nn;
#pragma unroll
    for (int k = 0; k < 3; ++k) vs_pf[k] = vs_nn[k];
#pragma unroll
    for (int q = 0; q < NQ; ++q) load_pose(0, q);
    load_slot(a.chunks > 1 ? 1 : 0);
    __syncthreads();
    __bf16* myP = &sP[wave][0][0];
    // A wave reduces the vertices its OWN threads own (64 wave ... 64 wave + 63 of the chunk): P goes through the wave's private LDS
    // rows only to be transposed (thread = vertex -> lane = 8 consecutive vertices of one product row), LDS operations of one wave
    // execute in order, and so the walk over poses and chunks needs NO barrier (the first version's barrier per pose and chunk left
    // the eight resident waves of a CU parked 58 % of their cycles).
    for (int c = 0; c < a.chunks; ++c) {
        const int v0 = c * 256, v = v0 + tid;
        const bool live = v < a.V;
        const f32x4 w4 = w4n;
        const int jj[4] = {j4n.x, j4n.y, j4n.z, j4n.w};
        // this wave's slice of the chunk's dense skinning weights: joints x its 64 vertices, hi and lo planes
        bf16x8 wf[4][2][2];
        auto load_wf = [&]() __attribute__((always_inline)) {
            const bf16x8* wp = a.wfrag + ((int64_t)(c * 4 + wave) * 16) * 64 + lane;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int p = 0; p < 2; ++p) wf[mt][kk][p] = wp[((mt * 2 + kk) * 2 + p) * 64];
        };
        if constexpr (!TMP) load_wf();      // (TMP: behind the pre-pass below -- 64 registers that its window of transforms needs)
        const int cn = c + 1 < a.chunks ? c + 1 : c;                         // (the last chunk prefetches itself again: no branch around a load)
        load_chunk(cn);
        const float vs_cur[3] = {vs_pf[0], vs_pf[1], vs_pf[2]};              // the current chunk's rest positions
        slot_pf = slot_nn;                                                   // chunk c + 1's (loaded during chunk c - 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) vs_pf[k] = vs_nn[k];
        load_slot(c + 2 < a.chunks ? c + 2 : a.chunks - 1);
        // TMP pre-pass: the temporal term's gradient of the thread's vertex for the G own poses.  The frames are skinned one after the
        // other as k_skin_x4 skins them (slot 0 = the frame in front of the workgroup's poses, G + 1 = the one behind) with a window of
        // three vertices and two transforms; per own pose the gradient (k_md_vert_grad's expression), its distance sum and T_R^T dv leave
        // the window.  The weight fragments of the joint reduction are loaded BEHIND it (with them the window spilled 49 registers).
        float dvk[TMP ? G : 1][3];
        if constexpr (TMP) {
            DP_SKIN_FP
            float Tq[9], vprev[3], vcur[3], hold[3];
            auto skin_slot = [&](int q, float* T9, float* vout) __attribute__((always_inline)) {
                DP_SKIN_FP
                float T[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) T[i] = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x4* Aj = reinterpret_cast<const f32x4*>(&sA[q][0]) + jj[k] * 3;
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const f32x4 row = Aj[r];
#pragma unroll
                        for (int i = 0; i < 4; ++i) T[4 * r + i] += w4[k] * row[i];
                    }
                }
                float pq[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) pq[k] = (VSB ? vsn[VSB ? q : 0][k] : vs_cur[k]) + ofn[q][k];      // rest + pose-blend offset, k_skin_x4's operand order
                const float tr0 = 0.f;         // (k_skin_x4 adds the translation -- none in this loop -- behind the transform: the add is kept, -0 + 0 = +0)
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    vout[r] = T[4 * r] * pq[0] + T[4 * r + 1] * pq[1] + T[4 * r + 2] * pq[2] + T[4 * r + 3] + tr0;
#pragma unroll
                    for (int cc = 0; cc < 3; ++cc) T9[3 * r + cc] = T[4 * r + cc];
                }
            };
            {
                float T0[9];
                skin_slot(0, T0, vprev);
                load_pose(cn, 0);
                skin_slot(1, Tq, vcur);
                {
                    DP_SKIN_FP_OFF
                    // pair (frame in front, first own frame): its backward-difference role only (k_md_vert_grad: v_rsq_f32 of the squared distance)
                    const float ax = vprev[0] - vcur[0], ay = vprev[1] - vcur[1], az = vprev[2] - vcur[2];
                    const float invb = __builtin_amdgcn_rsqf(ax * ax + ay * ay + az * az);
                    hold[0] = ax * invb; hold[1] = ay * invb; hold[2] = az * invb;
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const bool bv = b0 + g < a.B;
                const int tf = (int)((b0 + g) % a.F);
                const bool has_prev = tf > 0, has_next = tf + 1 < a.F;
                float Tn[9], vnext[3];
                skin_slot(g + 2, Tn, vnext);
                if (g + 1 == G) load_pose(cn, g + 2);
                float dx, dy, dz, d;
                {
                    DP_SKIN_FP_OFF
                    const float ax = vcur[0] - vnext[0], ay = vcur[1] - vnext[1], az = vcur[2] - vnext[2];
                    const float ss = ax * ax + ay * ay + az * az;
                    d = sqrtf(ss);
                    const float inv = __builtin_amdgcn_rcpf(d);
                    float gx = 0.f, gy = 0.f, gz = 0.f;
                    if (has_next) { gx = ax * inv; gy = ay * inv; gz = az * inv; }
                    if (has_prev) { gx -= hold[0]; gy -= hold[1]; gz -= hold[2]; }
                    const float invb = __builtin_amdgcn_rsqf(ss);
                    hold[0] = ax * invb; hold[1] = ay * invb; hold[2] = az * invb;
                    dx = (live && bv) ? a.c * gx : 0.f; dy = (live && bv) ? a.c * gy : 0.f; dz = (live && bv) ? a.c * gz : 0.f;
                }
                dvk[g][0] = dx; dvk[g][1] = dy; dvk[g][2] = dz;
                float dsum = (live && bv && has_next) ? d : 0.f;
#pragma unroll
                for (int sh = 32; sh >= 1; sh >>= 1) dsum += __shfl_xor(dsum, sh);
                if (lane == 0 && bv) a.part4[((b0 + g) * a.chunks + c) * 4 + wave] = dsum;
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) stage[wave][g][lane * 3 + cc] = Tq[cc] * dx + Tq[3 + cc] * dy + Tq[6 + cc] * dz;      // T_R^T dv
#pragma unroll
                for (int i = 0; i < 9; ++i) Tq[i] = Tn[i];
#pragma unroll
                for (int k = 0; k < 3; ++k) vcur[k] = vnext[k];
            }
            __builtin_amdgcn_sched_barrier(0);
            load_wf();
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const bool bv = b0 + g < a.B;
            float dx, dy, dz;
            float hv[4];
            if constexpr (TMP) {
                dx = dvk[g][0]; dy = dvk[g][1]; dz = dvk[g][2];
#pragma unroll
                for (int k = 0; k < 3; ++k) hv[k] = ofn[g + 1][k] + (VSB ? vsn[VSB ? g + 1 : 0][k] : vs_cur[k]);
                hv[3] = 1.0f;
                load_pose(cn, g + 1);
            } else {
                dx = (live && bv) ? dvn[g][0] : 0.f; dy = (live && bv) ? dvn[g][1] : 0.f; dz = (live && bv) ? dvn[g][2] : 0.f;
                hv[0] = ofn[g][0] + (VSB ? vsn[VSB ? g : 0][0] : vs_cur[0]); hv[1] = ofn[g][1] + (VSB ? vsn[VSB ? g : 0][1] : vs_cur[1]);
                hv[2] = ofn[g][2] + (VSB ? vsn[VSB ? g : 0][2] : vs_cur[2]); hv[3] = 1.0f;
                load_pose(cn, g);
                float T[9];
                {
#ifdef DPOSER_CT2      // (A/B: contraction in the plain backward's transform blend only)
                    DP_SKIN_FP
#endif
#pragma unroll
                    for (int i = 0; i < 9; ++i) T[i] = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const f32x4* Aj = reinterpret_cast<const f32x4*>(&sA[g][0]) + jj[k] * 3;
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            const f32x4 row = Aj[r];
#pragma unroll
                            for (int cc = 0; cc < 3; ++cc) T[3 * r + cc] += w4[k] * row[cc];
                        }
                    }
                }
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) stage[wave][g][lane * 3 + cc] = T[cc] * dx + T[3 + cc] * dy + T[6 + cc] * dz;      // T_R^T dv
            }
            // ---- A: thread = vertex
            {
                const float d3[3] = {dx, dy, dz};
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        const float pv = d3[r] * hv[qq];
                        const __bf16 ph = (__bf16)pv;
                        myP[(4 * r + qq) * SBM_PSTRIDE + lane] = ph;
                        myP[SBM_PLANE + (4 * r + qq) * SBM_PSTRIDE + lane] = (__bf16)(pv - (float)ph);
                    }
            }
            // ---- B: joint reduction of the wave's 64 vertices on the matrix pipe
            {
                bf16x8 pf[2][2];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int p = 0; p < 2; ++p)
                        pf[kk][p] = *reinterpret_cast<const bf16x8*>(myP + p * SBM_PLANE + (lane & 15) * SBM_PSTRIDE + 32 * kk + 8 * (lane >> 4));
                // (the four joint tiles are independent accumulators: innermost, so that no MFMA waits for its predecessor)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[g][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[mt][kk][0], pf[kk][0], acc[g][mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[g][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[mt][kk][1], pf[kk][0], acc[g][mt], 0, 0, 0);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) acc[g][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[mt][kk][0], pf[kk][1], acc[g][mt], 0, 0, 0);
                }
            }
        }
        // ---- outputs of the wave's vertices, all G poses at once: the G poses of a workgroup are consecutive rows of the FT operand,
        // i.e. 16 G contiguous bytes per group of 8 coordinates (pose by pose the same bytes left as G separate 16-byte stores: the
        // counters read 1037 MB written for 516 MB of operand)
        {
            const int vw = v0 + 64 * wave;                                  // first vertex of the wave
            const int nval = (a.V - vw < 64 ? (a.V - vw < 0 ? 0 : a.V - vw) : 64) * 3;
            for (int i = lane; i < 24 * G; i += 64) {
                const int g = i % G, grp = i / G;
                const int64_t b = b0 + g;
                const int k8 = vw * 3 + grp * 8;
                if (b < a.B && k8 < a.Cpad) {
                    const f32x4 x0 = *reinterpret_cast<const f32x4*>(&stage[wave][g][grp * 8]), x1 = *reinterpret_cast<const f32x4*>(&stage[wave][g][grp * 8 + 4]);
                    __bf16 hi[8], lo[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float x = (grp * 8 + e < nval) ? (e < 4 ? x0[e] : x1[e - 4]) : 0.f;
                        hi[e] = (__bf16)x;
                        lo[e] = (__bf16)(x - (float)hi[e]);
                    }
                    *reinterpret_cast<u32x4*>(a.doff_hi + FT<__bf16>::index(b, k8, a.Cpad)) = *reinterpret_cast<u32x4*>(hi);
                    *reinterpret_cast<u32x4*>(a.doff_lo + FT<__bf16>::index(b, k8, a.Cpad)) = *reinterpret_cast<u32x4*>(lo);
                }
            }
            if (a.dvp) {
                for (int i = lane; i < nval * G; i += 64) {
                    const int g = i / nval, k = i % nval;
                    if (b0 + g < a.B) a.dvp[((b0 + g) * a.V + vw) * 3 + k] = stage[wave][g][k];
                }
            }
        }
    }
    // the four waves' partial sums (vertex quarters of every chunk), added in wave order
    __syncthreads();
    float* red = reinterpret_cast<float*>(&sP[0][0][0]);                   // 16 KB of the planes' 33 KB
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f32x4*>(red + ((wave * 4 + mt) * 64 + lane) * 4) = acc[g][mt];
        __syncthreads();
        {
            const int mt = tid >> 6;
            f32x4 s = *reinterpret_cast<const f32x4*>(red + ((0 * 4 + mt) * 64 + lane) * 4);
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(red + ((w * 4 + mt) * 64 + lane) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) s[i] += t[i];
            }
            const int64_t b = b0 + g;
            const int n = lane & 15;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int j = 16 * mt + (lane >> 4) * 4 + i;                // C / D layout of the 16x16 MFMAs: column = lane & 15, row = 4 (lane >> 4) + register
                if (b < a.B && j < a.J && n < 12) a.dA[(b * a.J + j) * 12 + n] = s[i];
            }
        }
        __syncthreads();
    }
}
static int64_t lbs_joint_stream_min() { return body_tuning().joint_stream_min; }
// dposer_lbs_prepare_joint_lists: (host) cut the CSR-by-joint lists (vertex ids ascending inside a joint) into chunks of
// <= JL_MAXV vertices and <= JL_MAXE entries, sorted by joint inside a chunk -- the table the streaming joint-gradient kernel
// walks.  A SETUP call: it synchronises the stream, copies the lists to the host and (re)allocates the handle's tables.
// dposer_lbs_backward itself never synchronises or allocates: without prepared lists it runs the gather kernel.
extern "C" int dposer_lbs_prepare_joint_lists(dposer_body_t h, const int32_t* jptr_dev, const int32_t* jvidx_dev, const float* jw_dev, void* stream) {
    DP_CHECK_ARG(h && jptr_dev && jvidx_dev && jw_dev, "null argument");
    hipStream_t st = (hipStream_t)stream;
    const int J = h->d.num_joints, V = h->d.num_vertices;
    h->jl_ready = false;
    DP_CHECK_HIP(hipStreamSynchronize(st));
    std::vector<int32_t> jptr(J + 1);
    DP_CHECK_HIP(hipMemcpy(jptr.data(), jptr_dev, (J + 1) * sizeof(int32_t), hipMemcpyDeviceToHost));
    const int nnz = jptr[J];
    DP_CHECK_ARG(nnz >= 0 && jptr[0] == 0, "joint_ptr must be a CSR offset array");
    std::vector<int32_t> jv(nnz);
    std::vector<float> jw(nnz);
    if (nnz) {
        DP_CHECK_HIP(hipMemcpy(jv.data(), jvidx_dev, nnz * sizeof(int32_t), hipMemcpyDeviceToHost));
        DP_CHECK_HIP(hipMemcpy(jw.data(), jw_dev, nnz * sizeof(float), hipMemcpyDeviceToHost));
    }
    std::vector<int> per_vertex(V + 1, 0);
    for (int e = 0; e < nnz; ++e) {
        DP_CHECK_ARG(jv[e] >= 0 && jv[e] < V, "joint_vidx out of range");
        per_vertex[jv[e]]++;
    }
    std::vector<int32_t> vstart{0};
    for (int v = 0, nv = 0, ne = 0; v < V; ++v) {
        DP_CHECK_ARG(per_vertex[v] <= JL_MAXE, "a vertex is skinned to too many joints");
        if (nv + 1 > JL_MAXV || ne + per_vertex[v] > JL_MAXE) { vstart.push_back(v); nv = 0; ne = 0; }
        ++nv; ne += per_vertex[v];
    }
    vstart.push_back(V);
    const int chunks = (int)vstart.size() - 1;
    std::vector<int> chunk_of(V);
    for (int c = 0; c < chunks; ++c)
        for (int v = vstart[c]; v < vstart[c + 1]; ++v) chunk_of[v] = c;
    // count entries per (chunk, joint), then fill in (chunk, joint, ascending vertex) order
    std::vector<int32_t> cnt((size_t)chunks * (J + 1), 0), cfirst(chunks + 1, 0);
    for (int j = 0; j < J; ++j)
        for (int e = jptr[j]; e < jptr[j + 1]; ++e) cnt[(size_t)chunk_of[jv[e]] * (J + 1) + j + 1]++;
    for (int c = 0; c < chunks; ++c) {
        int32_t* p = &cnt[(size_t)c * (J + 1)];
        for (int j = 0; j < J; ++j) p[j + 1] += p[j];
        cfirst[c + 1] = cfirst[c] + p[J];
    }
    std::vector<float2> entry(nnz > 0 ? nnz : 1);
    std::vector<int32_t> fill(cnt);
    for (int j = 0; j < J; ++j)
        for (int e = jptr[j]; e < jptr[j + 1]; ++e) {
            const int c = chunk_of[jv[e]];
            const int pos = cfirst[c] + fill[(size_t)c * (J + 1) + j]++;
            int32_t local = jv[e] - vstart[c];
            float lf;
            std::memcpy(&lf, &local, 4);
            entry[pos] = make_float2(jw[e], lf);
        }
    (void)hipFree(h->jl_vstart); (void)hipFree(h->jl_ptr); (void)hipFree(h->jl_first); (void)hipFree(h->jl_entry);
    h->jl_vstart = nullptr; h->jl_ptr = nullptr; h->jl_first = nullptr; h->jl_entry = nullptr;
    DP_CHECK_HIP(hipMalloc(&h->jl_vstart, vstart.size() * sizeof(int32_t)));
    DP_CHECK_HIP(hipMalloc(&h->jl_ptr, cnt.size() * sizeof(int32_t)));
    DP_CHECK_HIP(hipMalloc(&h->jl_first, cfirst.size() * sizeof(int32_t)));
    DP_CHECK_HIP(hipMalloc(&h->jl_entry, entry.size() * sizeof(float2)));
    DP_CHECK_HIP(hipMemcpy(h->jl_vstart, vstart.data(), vstart.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    DP_CHECK_HIP(hipMemcpy(h->jl_ptr, cnt.data(), cnt.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    DP_CHECK_HIP(hipMemcpy(h->jl_first, cfirst.data(), cfirst.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    DP_CHECK_HIP(hipMemcpy(h->jl_entry, entry.data(), entry.size() * sizeof(float2), hipMemcpyHostToDevice));
    h->jl_chunks = chunks;
    // segment tables of the fused backward kernel: joint j's run in chunk c cut into pieces of <= 32 entries, so that the 64 lane
    // quads of a block share a chunk's ~1024 entries evenly (a chunk's vertices hang on a handful of joints: whole runs per quad,
    // as k_skin_bwd_joints walks them, leave one quad with hundreds of entries and the rest idle)
    (void)hipFree(h->jl_seg); (void)hipFree(h->jl_nseg); (void)hipFree(h->jl_jseg);
    h->jl_seg = nullptr; h->jl_nseg = nullptr; h->jl_jseg = nullptr;
    h->jl_fused_ok = false;
    {
        bool ok = true;
        std::vector<int32_t> seg((size_t)chunks * FUSED_MAXSEG * 2, 0), nseg(chunks, 0), jseg((size_t)chunks * J, 0);
        for (int c = 0; c < chunks && ok; ++c) {
            if (vstart[c] != c * 256 || cfirst[c + 1] - cfirst[c] > FUSED_MAXE) { ok = false; break; }
            const int32_t* pj = &cnt[(size_t)c * (J + 1)];
            // shortest segment length that gives every one of the block's 64 quads at most one segment
            int len = 32;
            for (int cand : {12, 16, 20, 24, 28, 32}) {
                int n = 0;
                for (int j = 0; j < J; ++j) n += (pj[j + 1] - pj[j] + cand - 1) / cand;
                if (n <= 64) { len = cand; break; }
            }
            int ns = 0;
            for (int j = 0; j < J; ++j) {
                const int first = ns;
                for (int b0 = pj[j]; b0 < pj[j + 1]; b0 += len) {
                    if (ns >= FUSED_MAXSEG) { ok = false; break; }
                    seg[((size_t)c * FUSED_MAXSEG + ns) * 2] = b0;
                    seg[((size_t)c * FUSED_MAXSEG + ns) * 2 + 1] = b0 + len < pj[j + 1] ? b0 + len : pj[j + 1];
                    ++ns;
                }
                jseg[(size_t)c * J + j] = first | ((ns - first) << 16);
            }
            nseg[c] = ns;
        }
        if (ok && J * 4 <= 256 && chunks < 64) {
            DP_CHECK_HIP(hipMalloc(&h->jl_seg, seg.size() * sizeof(int32_t)));
            DP_CHECK_HIP(hipMalloc(&h->jl_nseg, nseg.size() * sizeof(int32_t)));
            DP_CHECK_HIP(hipMalloc(&h->jl_jseg, jseg.size() * sizeof(int32_t)));
            DP_CHECK_HIP(hipMemcpy(h->jl_seg, seg.data(), seg.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            DP_CHECK_HIP(hipMemcpy(h->jl_nseg, nseg.data(), nseg.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            DP_CHECK_HIP(hipMemcpy(h->jl_jseg, jseg.data(), jseg.size() * sizeof(int32_t), hipMemcpyHostToDevice));
            h->jl_fused_ok = true;
        }
    }
    // dense weight chunks of k_skin_bwd_mfma: W^T[c][joint][local vertex] (entries of one (vertex, joint) pair summed), split hi / lo,
    // stored in the lane order of the MFMA A operand: [chunk][wave = vertex quarter][joint tile][k-step][plane][lane][8 vertices]
    (void)hipFree(h->jl_wfrag);
    h->jl_wfrag = nullptr;
    h->jl_mfma_ok = false;
    {
        bool regular = J <= 64 && chunks >= 1;
        for (int c = 0; c < chunks && regular; ++c) regular = vstart[c] == c * 256;
        if (regular) {
            std::vector<float> dense((size_t)chunks * 64 * 256, 0.f);
            for (int j = 0; j < J; ++j)
                for (int e = jptr[j]; e < jptr[j + 1]; ++e) dense[((size_t)(jv[e] >> 8) * 64 + j) * 256 + (jv[e] & 255)] += jw[e];
            std::vector<uint16_t> frag((size_t)chunks * 4 * 16 * 64 * 8);
            auto bf16_bits = [](float x) {                       // round to nearest even, as the device's float -> __bf16 conversion
                uint32_t u;
                std::memcpy(&u, &x, 4);
                u += 0x7fffu + ((u >> 16) & 1u);
                return (uint16_t)(u >> 16);
            };
            auto bf16_val = [](uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; std::memcpy(&f, &u, 4); return f; };
            for (int c = 0; c < chunks; ++c)
                for (int w = 0; w < 4; ++w)
                    for (int mt = 0; mt < 4; ++mt)
                        for (int kk = 0; kk < 2; ++kk)
                            for (int l = 0; l < 64; ++l)
                                for (int e = 0; e < 8; ++e) {
                                    const float x = dense[((size_t)c * 64 + 16 * mt + (l & 15)) * 256 + 64 * w + 32 * kk + 8 * (l >> 4) + e];
                                    const uint16_t hi = bf16_bits(x), lo = bf16_bits(x - bf16_val(hi));
                                    const size_t base = ((((size_t)(c * 4 + w) * 4 + mt) * 2 + kk) * 2) * 64;
                                    frag[((base + l) * 8) + e] = hi;
                                    frag[((base + 64 + l) * 8) + e] = lo;
                                }
            DP_CHECK_HIP(hipMalloc(&h->jl_wfrag, frag.size() * sizeof(uint16_t)));
            DP_CHECK_HIP(hipMemcpy(h->jl_wfrag, frag.data(), frag.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            h->jl_mfma_ok = true;
        }
    }
    h->jl_ready = true;
    return DPOSER_OK;
}

struct FkBwdArgs {
    const float* seg[FK_MAX_SEG];
    float* dseg[FK_MAX_SEG];       // gradient w.r.t. each pose segment (null: not needed)
    int seg_first[FK_MAX_SEG], seg_joints[FK_MAX_SEG], nseg;
    const float* j_rest;
    int j_rest_batched;
    const float* A;                // [B][J][12] forward skinning transforms
    const float* dA;               // [B][J][12]
    const float* djoints;          // [B][ld_dj] gradient w.r.t. posed joints (first J*3 of each row)
    int64_t ld_dj;
    const float* dpf;              // [nsplit][Bpad_pf][ldpf] slabs of d loss / d pose_feature, or null
    int64_t dpf_slab, ldpf;
    int pf_cols;                   // columns of dpf that were produced (joints beyond them get no pose gradient: their segment's output is null)
    int nsplit;
    float* djrest;                 // [B][J][3] out or null
    float* dG;                     // scratch [B][J][12]
    const int* parents;            // device [J]
    int J;
    int64_t B;
};
// Reverse walk of the kinematic chain, one lane per pose, joint index (and parent) compile-time constants so that the
// running gradients acc[joint][12] (d loss / d global transform, accumulated from the children) and the rest-joint
// gradients stay in REGISTERS: only the few joints between a leaf and its first processed ancestor are live at any time.
// (A first version kept them in a global scratch and paid a store -> load round trip per joint: 2 ms per call at any batch.)
template <typename Kin, int I>
__device__ __forceinline__ void fk_bwd_step(const FkBwdArgs& a, int64_t b, const float* jr, const float* A, float (&acc)[Kin::J][12],
                                            float (&djr)[Kin::J][3]) {
    constexpr int J = Kin::J;
    constexpr int P = Kin::P[I] < 0 ? 0 : Kin::P[I];
    // segment of joint I and its pointers through unrolled selects (indexing the kernel-argument arrays with a runtime value
    // would spill the whole argument struct to scratch)
    int li = I;
    const float* pose = a.seg[0];
    float* dq = a.dseg[0];
    int seg_nj = a.seg_joints[0];
#pragma unroll
    for (int k = 1; k < FK_MAX_SEG; ++k)
        if (k < a.nseg && I >= a.seg_first[k]) { li = I - a.seg_first[k]; pose = a.seg[k]; dq = a.dseg[k]; seg_nj = a.seg_joints[k]; }
    float rx = 0.f, ry = 0.f, rz = 0.f;
    if (pose) { const float* q = pose + (b * seg_nj + li) * 3; rx = q[0]; ry = q[1]; rz = q[2]; }
    // dG_I = (children's contributions) + (own terms):  A_I = [R | t - R J_I],  joints_I = t
    const float* dAi = a.dA + (b * J + I) * 12;
    const float J3[3] = {jr[3 * I], jr[3 * I + 1], jr[3 * I + 2]};
    float dRG[9], dt[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float dat = dAi[4 * r + 3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            dRG[3 * r + c] = acc[I][4 * r + c] + (dAi[4 * r + c] - dat * J3[c]);
            djr[I][c] -= A[I * 12 + 4 * r + c] * dat;                                   // d J_I += -R^T dA_t
        }
        dt[r] = acc[I][4 * r + 3] + (dat + a.djoints[b * a.ld_dj + 3 * I + r]);
    }
    float dR[9];                                   // gradient w.r.t. the LOCAL rotation R_I
    if constexpr (I == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) dR[k] = dRG[k];
#pragma unroll
        for (int c = 0; c < 3; ++c) djr[0][c] += dt[c];                                 // t_0 = J_0
    } else {
        const Mat3 R = rodrigues(rx, ry, rz);
        // parent's global transform from the forward pass: R^G_p = A_p[:, :3],  rel_I = J_I - J_p
        float RP[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) RP[3 * r + c] = A[P * 12 + 4 * r + c];
        const float rel[3] = {jr[3 * I] - jr[3 * P], jr[3 * I + 1] - jr[3 * P + 1], jr[3 * I + 2] - jr[3 * P + 2]};
        // dR_I = RP^T dRG ;  d rel = RP^T dt ;  dRG_p += dRG R_I^T + dt (x) rel ;  dt_p += dt
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) dR[3 * r + c] = RP[r] * dRG[c] + RP[3 + r] * dRG[3 + c] + RP[6 + r] * dRG[6 + c];
            const float drel = RP[r] * dt[0] + RP[3 + r] * dt[1] + RP[6 + r] * dt[2];
            djr[I][r] += drel;
            djr[P][r] -= drel;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                acc[P][4 * r + c] += dRG[3 * r] * R.m[3 * c] + dRG[3 * r + 1] * R.m[3 * c + 1] + dRG[3 * r + 2] * R.m[3 * c + 2] + dt[r] * rel[c];
            acc[P][4 * r + 3] += dt[r];
        }
        if (a.dpf && I * 9 <= a.pf_cols) {         // pose feature = R_I - I for I >= 1 (slabs already summed: nsplit == 1)
#pragma unroll
            for (int k = 0; k < 9; ++k) dR[k] += a.dpf[b * a.ldpf + (I - 1) * 9 + k];
        }
    }
    if (a.djrest) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.djrest[(b * J + I) * 3 + c] = djr[I][c];        // every child (index > I) has been processed
    }
    if (!dq) return;
    rodrigues_bwd(rx, ry, rz, dR, dq + (b * seg_nj + li) * 3);
}
template <typename Kin, int I>
__device__ __forceinline__ void fk_bwd_chain(const FkBwdArgs& a, int64_t b, const float* jr, const float* A, float (&acc)[Kin::J][12],
                                             float (&djr)[Kin::J][3]) {
    fk_bwd_step<Kin, I>(a, b, jr, A, acc, djr);
    // keep the next joint's global loads from being hoisted above this joint's arithmetic (that would make every joint's
    // inputs live at once and spill the accumulators)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" ::: "memory");
    if constexpr (I > 0) fk_bwd_chain<Kin, I - 1>(a, b, jr, A, acc, djr);
}
template <typename Kin> __global__ void __launch_bounds__(64, 1) k_fk_bwd(FkBwdArgs a) {
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= a.B) return;
    constexpr int J = Kin::J;
    const float* jr = a.j_rest_batched ? a.j_rest + b * J * 3 : a.j_rest;
    const float* A = a.A + b * J * 12;
    float acc[J][12], djr[J][3];
#pragma unroll
    for (int i = 0; i < J; ++i) {
#pragma unroll
        for (int k = 0; k < 12; ++k) acc[i][k] = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) djr[i][k] = 0.f;
    }
    fk_bwd_chain<Kin, J - 1>(a, b, jr, A, acc, djr);
}

// Small batches: one wave per pose, one lane per joint, the reverse walk by tree depth (deepest level first).  A joint hands
// its contribution to the parent's global-transform gradient (12 floats) and to the parent's rest-joint gradient (3) through
// LDS; the parent adds its children's contributions in DESCENDING joint order -- the order the unrolled chain of k_fk_bwd
// produces -- so both kernels return the same bits.
template <typename Kin> __global__ void __launch_bounds__(64) k_fk_bwd_small(FkBwdArgs a) {
    constexpr int J = Kin::J;
    constexpr int MAXD = fk_max_depth<Kin>();
    constexpr int MAXC = KinTable<Kin>::MAXC;
    static_assert(fk_max_children<Kin>() <= MAXC, "children per joint");
    __shared__ float sC[J][12];
    __shared__ float sD[J][3];
    const int64_t b = blockIdx.x;
    const int I = threadIdx.x;
    const bool on = I < J;
    int P = 0, depth = 0, child[MAXC], nchild = 0;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) child[k] = 0;
    if (on) {
        P = kKinTable<Kin>.p[I];
        depth = kKinTable<Kin>.depth[I];
        nchild = kKinTable<Kin>.nchild[I];
#pragma unroll
        for (int k = 0; k < MAXC; ++k) child[k] = kKinTable<Kin>.child[I][k];
    }
    int li = I;
    const float* pose = a.seg[0];
    float* dq = a.dseg[0];
    int seg_nj = a.seg_joints[0];
#pragma unroll
    for (int k = 1; k < FK_MAX_SEG; ++k)
        if (k < a.nseg && I >= a.seg_first[k]) { li = I - a.seg_first[k]; pose = a.seg[k]; dq = a.dseg[k]; seg_nj = a.seg_joints[k]; }
    float rx = 0.f, ry = 0.f, rz = 0.f;
    if (on && pose) { const float* q = pose + (b * seg_nj + li) * 3; rx = q[0]; ry = q[1]; rz = q[2]; }
    const float* jr = a.j_rest_batched ? a.j_rest + b * J * 3 : a.j_rest;
    const float* A = a.A + b * J * 12;
    const int Ic = on ? I : 0;
    const float* dAi = a.dA + (b * J + Ic) * 12;
    const float J3[3] = {jr[3 * Ic], jr[3 * Ic + 1], jr[3 * Ic + 2]};
    const Mat3 R = rodrigues(rx, ry, rz);
    float RP[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) RP[3 * r + c] = A[P * 12 + 4 * r + c];
    const float rel[3] = {jr[3 * Ic] - jr[3 * P], jr[3 * Ic + 1] - jr[3 * P + 1], jr[3 * Ic + 2] - jr[3 * P + 2]};
    float dR[9], djr[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 9; ++k) dR[k] = 0.f;
    for (int L = MAXD; L >= 0; --L) {
        if (on && depth == L) {
            float acc[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) acc[k] = 0.f;
#pragma unroll
            for (int k = 0; k < MAXC; ++k)
                if (k < nchild) {
                    const int c = child[k];
#pragma unroll
                    for (int i = 0; i < 12; ++i) acc[i] += sC[c][i];
#pragma unroll
                    for (int r = 0; r < 3; ++r) djr[r] -= sD[c][r];
                }
            float dRG[9], dt[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float dat = dAi[4 * r + 3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    dRG[3 * r + c] = acc[4 * r + c] + (dAi[4 * r + c] - dat * J3[c]);
                    djr[c] -= A[I * 12 + 4 * r + c] * dat;
                }
                dt[r] = acc[4 * r + 3] + (dat + a.djoints[b * a.ld_dj + 3 * I + r]);
            }
            if (I == 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k) dR[k] = dRG[k];
#pragma unroll
                for (int c = 0; c < 3; ++c) djr[c] += dt[c];
            } else {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) dR[3 * r + c] = RP[r] * dRG[c] + RP[3 + r] * dRG[3 + c] + RP[6 + r] * dRG[6 + c];
                    const float drel = RP[r] * dt[0] + RP[3 + r] * dt[1] + RP[6 + r] * dt[2];
                    djr[r] += drel;
                    sD[I][r] = drel;
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        sC[I][4 * r + c] = dRG[3 * r] * R.m[3 * c] + dRG[3 * r + 1] * R.m[3 * c + 1] + dRG[3 * r + 2] * R.m[3 * c + 2] + dt[r] * rel[c];
                    sC[I][4 * r + 3] = dt[r];
                }
            }
        }
        __syncthreads();
    }
    if (!on) return;
    if (I > 0 && a.dpf && I * 9 <= a.pf_cols) {
#pragma unroll
        for (int k = 0; k < 9; ++k) dR[k] += a.dpf[b * a.ldpf + (I - 1) * 9 + k];
    }
    if (a.djrest) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.djrest[(b * J + I) * 3 + c] = djr[c];
    }
    if (!dq) return;
    rodrigues_bwd(rx, ry, rz, dR, dq + (b * seg_nj + li) * 3);
}

// d pose_feature: sum of the split-K slabs of the d_off @ posedirs^T GEMM, [nsplit][Bpad][ld] -> slab 0 (fixed order)
__global__ void __launch_bounds__(256) k_sum_slabs(float* slabs, int64_t slab_elems, int nsplit) {
    const int64_t n4 = slab_elems >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 v = reinterpret_cast<const f32x4*>(slabs)[i];
        // (eight slabs' loads in flight, added in slab order: at 60 poses this is 32 blocks walking 72 slabs -- one load per round trip took 19 us)
        int k = 1;
        for (; k + 8 <= nsplit; k += 8) {
            f32x4 w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = reinterpret_cast<const f32x4*>(slabs + (int64_t)(k + u) * slab_elems)[i];
#pragma unroll
            for (int u = 0; u < 8; ++u) { v[0] += w[u][0]; v[1] += w[u][1]; v[2] += w[u][2]; v[3] += w[u][3]; }
        }
        for (; k < nsplit; ++k) {
            const f32x4 w = reinterpret_cast<const f32x4*>(slabs + (int64_t)k * slab_elems)[i];
            v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
        }
        reinterpret_cast<f32x4*>(slabs)[i] = v;
    }
}

// The slabs of the row-concatenated blend gradient: A [ka][rows][2 pe] (columns [0, pe) = d_off_hi pd_hi^T, [pe, 2 pe) = d_off_hi pd_lo^T) and
// B [kb][rows][pe] (d_off_lo pd_hi^T) -> out [rows][pe], added in k_sum_slabs' order of the three-launch form (term by term, split by split):
// the same bits.  `out` may be B's first slab (every thread reads its own elements before it writes them).
__global__ void __launch_bounds__(256) k_sum_slabs_rowcat(const float* A, int ka, const float* B, int kb, float* out, int64_t rows, int pe) {
    const int64_t n4 = rows * pe / 4;
    const int pe4 = pe / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / pe4;
        const int c4 = (int)(i % pe4);
        const f32x4* a0 = reinterpret_cast<const f32x4*>(A + r * 2 * pe) + c4;
        f32x4 v = a0[0];
        for (int k = 1; k < ka; ++k) {
            const f32x4 w = a0[k * (rows * 2 * pe / 4)];
            v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
        }
        for (int k = 0; k < ka; ++k) {
            const f32x4 w = a0[k * (rows * 2 * pe / 4) + pe4];
            v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
        }
        for (int k = 0; k < kb; ++k) {
            const f32x4 w = reinterpret_cast<const f32x4*>(B + k * rows * pe)[i];
            v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
        }
        reinterpret_cast<f32x4*>(out)[i] = v;
    }
}
extern "C" int64_t dposer_lbs_posedirs_bwd_packed_bytes(dposer_body_t h) {
    if (!h) return -1;
    const int64_t prow = round_up((h->d.num_joints - 1) * 9, 128), Cpad = lbs_cpad(h->d.num_vertices);
    // [rows = pose feature][k = vertex coord]: FT32 | bf16 high | bf16 low | (prow > 256) the first 256 rows once more as ONE matrix
    // [high 0..255 ; low 0..255]: the row-concatenated operand of the combined blend-gradient launch when only the body is posed
    return prow * Cpad * 8 + (prow > 256 ? 2 * 256 * Cpad * 2 : 0);
}
extern "C" int dposer_lbs_pack_posedirs_bwd(dposer_body_t h, const float* posedirs, void* packed, void* stream) {
    DP_CHECK_ARG(h && posedirs && packed, "null argument");
    PackJobs js;
    js.n = 1;
    PackJob& j = js.job[0];
    const int P = (h->d.num_joints - 1) * 9;
    j.dst_off = 0; j.src_off = 0; j.ktot = (int)lbs_cpad(h->d.num_vertices); j.koff = 0;
    j.rows_pad = (int)round_up(P, 128); j.kpad = j.ktot; j.rows_valid = P; j.cols_valid = h->d.num_vertices * 3;
    j.ld = h->d.num_vertices * 3; j.trans = 0; j.f32 = 1; j.split = 0;
    const int64_t n = (int64_t)j.rows_pad * j.ktot;
    js.n = 3;
    js.job[1] = j; js.job[1].dst_off = n * 4; js.job[1].f32 = 0; js.job[1].split = 1;
    js.job[2] = j; js.job[2].dst_off = n * 6; js.job[2].f32 = 0; js.job[2].split = 2;
    if (j.rows_pad > 256) {
        const int64_t n256 = (int64_t)256 * j.ktot;
        js.n = 5;
        js.job[3] = j; js.job[3].dst_off = n * 8; js.job[3].f32 = 0; js.job[3].split = 1; js.job[3].rows_pad = 256; js.job[3].rows_valid = P < 256 ? P : 256;
        js.job[4] = js.job[3]; js.job[4].dst_off = n * 8 + n256 * 2; js.job[4].split = 2;
    }
    FK_HIP_LAUNCH(launch_pack(js, posedirs, packed, (hipStream_t)stream));
    return DPOSER_OK;
}
static int lbs_bwd_ksplit(int64_t stages) {
    int ks = 1;
    for (int c = 2; c <= 24; ++c)
        if (stages % c == 0 && stages / c >= 8) ks = c;
    return ks;
}
// bf16 x 3 backward: three launches share the GPU; enough splits that one launch alone covers ~2 workgroups per CU
static int lbs_bwd_ksplit_bf16(int64_t kblocks, int64_t tiles, int max_split = 24) {
    const int64_t want = ceil_div(512, tiles);
    int best = 1;
    for (int c = 1; c <= max_split && c <= 24; ++c) {
        if (kblocks % (2 * c) != 0 || kblocks / c < 16) continue;      // whole 2-k-block stages per split
        best = c;
        if (c >= want) break;
    }
    return best;
}
// 256x256 tiles for the bf16 backward blend GEMMs from 2048 poses up (one 256-CU round of long-K workgroups: 3 x 110 us instead of
// 3 x 161 us with the 128x128 tiles at 4096 poses): the split count that fills the last round best, at most 8 (24 slabs); 0 = not applicable
static int lbs_bwd_big_ksplit(int64_t Bpad, int64_t prow, int64_t kblocks, int max_split = 8, int concurrent = 1) {
    if (!body_tuning().lbs_bwd_big || Bpad % 256 != 0 || prow % 256 != 0 || Bpad < 2048) return 0;
    const int64_t tiles = (Bpad / 256) * (prow / 256) * concurrent;      // (launches that share the chip count as one grid)
    int best = 0;
    double best_u = 0.0;
    for (int c = 1; c <= max_split; ++c) {
        if (kblocks % (2 * c) != 0 || kblocks / c < 32) continue;
        const int64_t wg = tiles * c;
        const double u = (double)wg / (double)(256 * ceil_div(wg, 256));
        if (u > best_u + 1e-9) { best_u = u; best = c; }
    }
    return best_u >= 0.75 ? best : 0;
}
static int64_t lbs_bwd_slabs(int64_t Bpad, int64_t Cpad, int64_t prow) {
    const int ks = lbs_bwd_ksplit_bf16(Cpad / 16, (Bpad / 128) * (prow / 128));
    return 3 * ks > 24 ? 3 * ks : 24;
}
extern "C" int64_t dposer_lbs_backward_workspace_bytes(dposer_body_t h, int64_t batch) {
    if (!h || batch <= 0) return -1;
    const int64_t Bpad = lbs_pad_batch(batch), Cpad = lbs_cpad(h->d.num_vertices);
    const int J = h->d.num_joints, V = h->d.num_vertices;
    const int64_t prow = round_up((J - 1) * 9, 128);
    int64_t p = 0;
    p += round_up(batch * V * 3 * 4, 256) * 2;         // vp, dvp
    p += round_up(Bpad * Cpad * 4, 256);               // doff FT32
    p += round_up(batch * J * 12 * 4, 256) * 2;        // dA, dG
    p += round_up(lbs_bwd_slabs(Bpad, Cpad, prow) * Bpad * prow * 4, 256);          // dpf slabs
    p += 256;                                          // parents
    return p;
}

// ws_fwd: the workspace the matching dposer_lbs_forward call used (pose feature, A, offsets are read back).
extern "C" int dposer_lbs_backward(dposer_body_t h, const void* ws_fwd, void* ws_bwd, const void* posedirs_bwd_packed,
                                   const float* const* pose_segments_host, const int32_t* segment_joints_host, int32_t num_segments,
                                   const float* j_rest, int32_t j_rest_batched, const float* v_shaped, int32_t v_shaped_batched,
                                   const int32_t* skin_idx, const float* skin_w, int32_t skin_k, const int32_t* joint_ptr,
                                   const int32_t* joint_vidx, const float* joint_w, const float* d_verts, const float* d_joints,
                                   int64_t d_joints_ld, float* const* d_pose_segments_host, float* d_jrest, float* d_vposed,
                                   int64_t batch, void* stream) {
    return dposer_lbs_backward_fold(h, ws_fwd, ws_bwd, posedirs_bwd_packed, pose_segments_host, segment_joints_host, num_segments, j_rest, j_rest_batched,
                                    v_shaped, v_shaped_batched, skin_idx, skin_w, skin_k, joint_ptr, joint_vidx, joint_w, d_verts, d_joints, d_joints_ld,
                                    nullptr, d_pose_segments_host, d_jrest, d_vposed, batch, stream);
}
// the temporal term of the motion-denoising loss as the vertex gradient (k_skin_bwd_mfma<TMP>): frames per sequence, weight / ((F - 1) V),
// per-wave distance sums [batch][ceil(V / 256)][4]
struct LbsTemporal { int64_t F; float scale; float* part4; };
static bool lbs_temporal_in_backward_ok(dposer_body_t h, int32_t skin_k, int64_t batch) {
    return h && !lbs_blend_fp32() && skin_k == 4 && h->jl_ready && h->jl_fused_ok && h->jl_mfma_ok && batch >= lbs_joint_stream_min() &&
           body_tuning().skin_bwd_fused && body_tuning().skin_bwd_mfma != 0 && (int64_t)h->jl_chunks * 768 >= lbs_cpad(h->d.num_vertices) &&
           h->jl_chunks == (int)ceil_div(h->d.num_vertices, 256);
}
extern "C" int32_t dposer_lbs_temporal_in_backward_ok(dposer_body_t h, int32_t skin_k, int64_t batch) { return lbs_temporal_in_backward_ok(h, skin_k, batch) ? 1 : 0; }
static int lbs_backward_impl(dposer_body_t h, const void* ws_fwd, void* ws_bwd, const void* posedirs_bwd_packed,
                             const float* const* pose_segments_host, const int32_t* segment_joints_host, int32_t num_segments,
                             const float* j_rest, int32_t j_rest_batched, const float* v_shaped, int32_t v_shaped_batched,
                             const int32_t* skin_idx, const float* skin_w, int32_t skin_k, const int32_t* joint_ptr,
                             const int32_t* joint_vidx, const float* joint_w, const float* d_verts, const float* d_joints,
                             int64_t d_joints_ld, const dposer_lbs_joint_fold* fold, float* const* d_pose_segments_host, float* d_jrest,
                             float* d_vposed, int64_t batch, void* stream, const LbsTemporal* tmp);
extern "C" int dposer_lbs_backward_fold(dposer_body_t h, const void* ws_fwd, void* ws_bwd, const void* posedirs_bwd_packed,
                                        const float* const* pose_segments_host, const int32_t* segment_joints_host, int32_t num_segments,
                                        const float* j_rest, int32_t j_rest_batched, const float* v_shaped, int32_t v_shaped_batched,
                                        const int32_t* skin_idx, const float* skin_w, int32_t skin_k, const int32_t* joint_ptr,
                                        const int32_t* joint_vidx, const float* joint_w, const float* d_verts, const float* d_joints,
                                        int64_t d_joints_ld, const dposer_lbs_joint_fold* fold, float* const* d_pose_segments_host, float* d_jrest,
                                        float* d_vposed, int64_t batch, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(d_verts, "null argument");
    return lbs_backward_impl(h, ws_fwd, ws_bwd, posedirs_bwd_packed, pose_segments_host, segment_joints_host, num_segments, j_rest, j_rest_batched, v_shaped,
                             v_shaped_batched, skin_idx, skin_w, skin_k, joint_ptr, joint_vidx, joint_w, d_verts, d_joints, d_joints_ld, fold,
                             d_pose_segments_host, d_jrest, d_vposed, batch, stream, nullptr);
}
// LBS backward of the motion-denoising step (run/motion_denoising.py:253-267) WITHOUT a vertex gradient in HBM: d loss / d vertices is the
// temporal term's, formed inside the skinning-backward kernel from the forward workspace (skinning transforms + pose-blend offsets of the
// pose and of its two neighbouring frames); d_joints carries the data term.  dist_part4 [batch][ceil(V / 256)][4] receives the per-wave sums
// of ||v[t] - v[t+1]|| (0 behind a sequence's last frame): ((p0 + p1) + p2) + p3 of an entry is k_md_vert_grad's block sum.  Needs
// dposer_lbs_temporal_in_backward_ok().  Same vertices and vertex gradient as dposer_lbs_forward + k_md_vert_grad, bit for bit; the backward half (own pose's
// 3 x 3 transform, T^T dv) is FMA-contracted here and agrees with dposer_lbs_backward to fp32 rounding.
extern "C" int dposer_lbs_backward_temporal(dposer_body_t h, const void* ws_fwd, void* ws_bwd, const void* posedirs_bwd_packed,
                                            const float* const* pose_segments_host, const int32_t* segment_joints_host, int32_t num_segments,
                                            const float* j_rest, int32_t j_rest_batched, const float* v_shaped, int32_t v_shaped_batched,
                                            const int32_t* skin_idx, const float* skin_w, int32_t skin_k, const int32_t* joint_ptr,
                                            const int32_t* joint_vidx, const float* joint_w, int64_t frames_per_sequence, float scale,
                                            float* dist_part4, const float* d_joints, int64_t d_joints_ld, float* const* d_pose_segments_host,
                                            int64_t batch, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(dist_part4 && frames_per_sequence >= 2 && batch > 0 && batch % frames_per_sequence == 0, "batch must be whole sequences of >= 2 frames");
    if (!lbs_temporal_in_backward_ok(h, skin_k, batch))
        return dposer_set_error(DPOSER_ERR_UNSUPPORTED, "dposer_lbs_backward_temporal: needs the matrix-pipe skinning backward (4 influences per vertex, prepared joint lists, batch >= DPOSER_LBS_JOINT_STREAM_MIN)");
    LbsTemporal t;
    t.F = frames_per_sequence; t.scale = scale; t.part4 = dist_part4;
    return lbs_backward_impl(h, ws_fwd, ws_bwd, posedirs_bwd_packed, pose_segments_host, segment_joints_host, num_segments, j_rest, j_rest_batched, v_shaped,
                             v_shaped_batched, skin_idx, skin_w, skin_k, joint_ptr, joint_vidx, joint_w, nullptr, d_joints, d_joints_ld, nullptr,
                             d_pose_segments_host, nullptr, nullptr, batch, stream, &t);
}
// FK + pose-blend offsets of dposer_lbs_forward WITHOUT the skinning kernel: joints[:, :J] and the workspace (pose feature, skinning transforms,
// offsets) that dposer_lbs_backward_temporal reads.  The rows of `joints` behind the kinematic tree (extra vertices, landmarks) are not written.
extern "C" int dposer_lbs_forward_front(dposer_body_t h, void* ws, const void* posedirs_packed, const float* const* pose_segments_host,
                                        const int32_t* segment_joints_host, int32_t num_segments, const float* j_rest, int32_t j_rest_batched,
                                        const float* transl, float* joints, int64_t batch, void* stream) {
    DP_RANGE();
    float *A = nullptr, *offsets = nullptr;
    return lbs_forward_front(h, ws, posedirs_packed, pose_segments_host, segment_joints_host, num_segments, j_rest, j_rest_batched, transl, joints, batch,
                             stream, &A, &offsets);
}
static int lbs_backward_impl(dposer_body_t h, const void* ws_fwd, void* ws_bwd, const void* posedirs_bwd_packed,
                             const float* const* pose_segments_host, const int32_t* segment_joints_host, int32_t num_segments,
                             const float* j_rest, int32_t j_rest_batched, const float* v_shaped, int32_t v_shaped_batched,
                             const int32_t* skin_idx, const float* skin_w, int32_t skin_k, const int32_t* joint_ptr,
                             const int32_t* joint_vidx, const float* joint_w, const float* d_verts, const float* d_joints,
                             int64_t d_joints_ld, const dposer_lbs_joint_fold* fold, float* const* d_pose_segments_host, float* d_jrest,
                             float* d_vposed, int64_t batch, void* stream, const LbsTemporal* tmp) {
    DP_CHECK_ARG(!fold || (fold->n_slots >= 0 && fold->n_slots <= h->d.num_vertices && (fold->n_slots == 0 || (fold->vertex_slot && fold->slot_vertex &&
                                                                                     fold->slot_ptr && fold->entry_row && fold->entry_weight))),
                 "bad joint fold tables");
    DP_CHECK_ARG(h && ws_fwd && ws_bwd && posedirs_bwd_packed && pose_segments_host && segment_joints_host && j_rest && v_shaped && skin_idx &&
                     skin_w && joint_ptr && joint_vidx && joint_w && (d_verts || tmp) && d_joints && d_pose_segments_host,
                 "null argument");
    DP_CHECK_ARG(batch > 0 && num_segments >= 1 && num_segments <= FK_MAX_SEG, "bad size");
    hipStream_t st = (hipStream_t)stream;
    const int J = h->d.num_joints, V = h->d.num_vertices;
    const int64_t Bpad = lbs_pad_batch(batch), Cpad = lbs_cpad(V);
    const int Ppad = lbs_ppad(J);
    const int64_t prow = round_up((J - 1) * 9, 128);
    // forward workspace layout (dposer_lbs_forward)
    const char* pf_ = (const char*)ws_fwd;
    const float* A = (const float*)(pf_ + round_up(Bpad * Ppad * 4, 256));
    const float* offsets = (const float*)((const char*)A + round_up(batch * J * 12 * 4, 256));
    char* p = (char*)ws_bwd;
    float* vp = (float*)p; p += round_up(batch * V * 3 * 4, 256);
    float* dv_fixed = (float*)p;                                        // [B][U][3] corrected vertex-gradient rows of the joint fold (U <= V)
    p += round_up(batch * V * 3 * 4, 256);
    float* doff = (float*)p; p += round_up(Bpad * Cpad * 4, 256);
    float* dA = (float*)p; p += round_up(batch * J * 12 * 4, 256);
    float* dG = (float*)p; p += round_up(batch * J * 12 * 4, 256);
    float* dpf = (float*)p; p += round_up(lbs_bwd_slabs(Bpad, Cpad, prow) * Bpad * prow * 4, 256);
    // (the kernels walk compile-time parents tables: the per-call host-to-device copy of the parents array that used to sit
    //  here -- a pageable-memory copy in the middle of every optimisation step -- fed nothing)
    const bool blend32 = lbs_blend_fp32();
    __bf16* doff_hi = reinterpret_cast<__bf16*>(doff);                  // bf16 x 3: the two terms share the FT32 operand's bytes
    __bf16* doff_lo = doff_hi + Bpad * Cpad;
    VertGrad vg;
    vg.dverts = d_verts; vg.vslot = nullptr; vg.fixed = dv_fixed; vg.U = 0;
    if (fold && fold->n_slots > 0) {
        FoldArgs f;
        f.vslot = fold->vertex_slot; f.uniq = fold->slot_vertex; f.ptr = fold->slot_ptr; f.row = fold->entry_row; f.w = fold->entry_weight; f.U = fold->n_slots;
        hipLaunchKernelGGL(k_fold_rows, dim3((unsigned)ceil_div(f.U, 64), (unsigned)batch), dim3(64), 0, st, f, d_verts, d_joints, d_joints_ld, dv_fixed, V, batch);
        FK_HIP_LAUNCH(hipGetLastError());
        vg.vslot = fold->vertex_slot; vg.U = fold->n_slots;
    }
    const bool fused = !blend32 && skin_k == 4 && h->jl_ready && h->jl_fused_ok && batch >= lbs_joint_stream_min() && body_tuning().skin_bwd_fused;
    if (fused && (int64_t)h->jl_chunks * 768 >= Cpad) {
        // the fused kernel writes every coordinate column of every pose: only the 32-row groups that hold padding rows need zeros (a
        // contiguous tail of each FT array).  The full clear was 516 MB = 91 us per backward at 4096 poses.
        const int64_t r0 = batch & ~(int64_t)31;
        if (r0 < Bpad) {
            DP_CHECK_HIP(hipMemsetAsync(doff_hi + r0 * Cpad, 0, (Bpad - r0) * Cpad * 2, st));
            DP_CHECK_HIP(hipMemsetAsync(doff_lo + r0 * Cpad, 0, (Bpad - r0) * Cpad * 2, st));
        }
    } else {
        DP_CHECK_HIP(hipMemsetAsync(doff, 0, Bpad * Cpad * 4, st));
    }
    const int mfma_g = body_tuning().skin_bwd_mfma;
    if (fused && mfma_g != 0 && h->jl_mfma_ok) {
        SkinBwdMfmaArgs a;
        a.vg = vg;
        a.dverts = d_verts; a.offsets = offsets; a.ld_off = Cpad; a.v_shaped = v_shaped; a.v_shaped_batched = v_shaped_batched; a.A = A;
        a.skin_idx = skin_idx; a.skin_w = skin_w; a.J = J; a.V = V; a.dvp = d_vposed; a.doff_hi = doff_hi; a.doff_lo = doff_lo; a.Cpad = (int)Cpad;
        a.wfrag = reinterpret_cast<const bf16x8*>(h->jl_wfrag); a.dA = dA; a.chunks = h->jl_chunks; a.B = batch;
        a.F = tmp ? (int)tmp->F : 1; a.c = tmp ? tmp->scale : 0.f; a.part4 = tmp ? tmp->part4 : nullptr;
        if (tmp) {
            // five poses per workgroup from 5120 frames: 7 skinnings per 5 poses instead of 6 per 4, the chunk's weight fragments serve one pose
            // more (243 VGPRs, no spills; six spill) -- cfg 5 x 128 (7680 frames): 2.37 -> 2.28 ms per step; below, the smaller grid's last round
            // costs more than that (1920 frames: 0.78 -> 0.85 ms).  DPOSER_SKIN_BWD_MFMA=4 / 5 forces four / five (A/B; profiles/r06_skin_contract_ab3.md).
            // (a rest shape per frame keeps four: with five its instantiation spills 4 VGPRs)
            const bool five = !v_shaped_batched && (mfma_g == 5 || (mfma_g != 4 && batch >= 5120));
            if (v_shaped_batched) hipLaunchKernelGGL((k_skin_bwd_mfma<4, true, true>), dim3((unsigned)ceil_div(batch, 4)), dim3(256), 0, st, a);
            else if (!five) hipLaunchKernelGGL((k_skin_bwd_mfma<4, false, true>), dim3((unsigned)ceil_div(batch, 4)), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_skin_bwd_mfma<5, false, true>), dim3((unsigned)ceil_div(batch, 5)), dim3(256), 0, st, a);
        } else if (mfma_g == 2) {
            if (v_shaped_batched) hipLaunchKernelGGL((k_skin_bwd_mfma<2, true>), dim3((unsigned)ceil_div(batch, 2)), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_skin_bwd_mfma<2, false>), dim3((unsigned)ceil_div(batch, 2)), dim3(256), 0, st, a);
        } else {
            if (v_shaped_batched) hipLaunchKernelGGL((k_skin_bwd_mfma<4, true>), dim3((unsigned)ceil_div(batch, 4)), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((k_skin_bwd_mfma<4, false>), dim3((unsigned)ceil_div(batch, 4)), dim3(256), 0, st, a);
        }
        FK_HIP_LAUNCH(hipGetLastError());
    } else if (fused) {
        SkinBwdFusedArgs a;
        a.vg = vg;
        a.dverts = d_verts; a.offsets = offsets; a.ld_off = Cpad; a.v_shaped = v_shaped; a.v_shaped_batched = v_shaped_batched; a.A = A;
        a.skin_idx = skin_idx; a.skin_w = skin_w; a.J = J; a.V = V; a.dvp = d_vposed; a.doff_hi = doff_hi; a.doff_lo = doff_lo; a.Cpad = (int)Cpad;
        a.cfirst = h->jl_first; a.entry = h->jl_entry; a.seg = reinterpret_cast<const int2*>(h->jl_seg); a.nseg = h->jl_nseg; a.jseg = h->jl_jseg;
        a.dA = dA; a.chunks = h->jl_chunks; a.B = batch;
        hipLaunchKernelGGL(k_skin_bwd_fused, dim3((unsigned)batch), dim3(256), 0, st, a);
        FK_HIP_LAUNCH(hipGetLastError());
    } else {
        SkinBwdArgs a;
        a.vg = vg;
        a.dverts = d_verts; a.offsets = offsets; a.ld_off = Cpad; a.v_shaped = v_shaped; a.v_shaped_batched = v_shaped_batched; a.A = A;
        a.skin_idx = skin_idx; a.skin_w = skin_w; a.K = skin_k; a.J = J; a.V = V; a.vp = vp; a.dvp = d_vposed;
        a.doff_ft = blend32 ? doff : nullptr; a.doff_hi = doff_hi; a.doff_lo = doff_lo;
        a.Cpad = (int)Cpad;
        hipLaunchKernelGGL(k_skin_bwd, dim3((unsigned)ceil_div(V, 256 * 4), (unsigned)batch), dim3(256), (J * 12 + 1536) * sizeof(float), st, a);
        FK_HIP_LAUNCH(hipGetLastError());
    }
    if (!fused) {
        if (batch >= lbs_joint_stream_min() && h->jl_ready) {
            JointBwdArgs a;
            a.vg = vg;
            a.dverts = d_verts; a.vp = vp; a.vstart = h->jl_vstart; a.cptr = h->jl_ptr; a.cfirst = h->jl_first; a.entry = h->jl_entry; a.dA = dA;
            a.J = J; a.V = V; a.chunks = h->jl_chunks; a.B = batch;
            hipLaunchKernelGGL(k_skin_bwd_joints, dim3((unsigned)batch), dim3(256), 0, st, a);
        } else {
            JointGatherArgs a;
            a.vg = vg;
            a.dverts = d_verts; a.vp = vp; a.jptr = joint_ptr; a.jvidx = joint_vidx; a.jw = joint_w; a.dA = dA; a.J = J; a.V = V; a.B = batch;
            hipLaunchKernelGGL(k_skin_bwd_joints_gather, dim3((unsigned)(ceil_div(batch, 8) * 8 * J)), dim3(128), 0, st, a);
        }
        FK_HIP_LAUNCH(hipGetLastError());
    }
    // d pose_feature [B][486] = d_off [B][3V] @ posedirs^T : split over the vertex dimension.  Only the columns of joints whose pose
    // gradient is wanted are produced (lbs_posed_cols: 189 -> 256 of 512 for the body): compact slabs [Bpad][pe], fewer column tiles
    const int64_t slab_budget = lbs_bwd_slabs(Bpad, Cpad, prow) * Bpad * prow;      // floats reserved for the slabs
    const int want_cols = body_tuning().lbs_k_prefix ? lbs_posed_cols(pose_segments_host, segment_joints_host, num_segments, J, (const float* const*)d_pose_segments_host)
                                                     : (J - 1) * 9;
    const int64_t stages = Cpad / 32;
    int ks = lbs_bwd_ksplit(stages);            // number of slabs k_sum_slabs adds up below
    int64_t pe = prow;                          // slab row length = columns produced
    if (blend32) {
        pe = round_up(want_cols < 1 ? 1 : want_cols, 128);
        pe = pe > prow ? prow : pe;
        GemmArgs g;
        std::memset(&g, 0, sizeof(g));
        g.W = doff; g.w_stride_blocks = (int)(Cpad / 8); g.n_cblk = (int)(Bpad / 128); g.n_sblk = (int)(pe / 128); g.ksplit = ks;
        g.src[0] = posedirs_bwd_packed; g.seg_kblocks[0] = (int)(Cpad / 8); g.nseg = 1; g.ktot_blocks = (int)(Cpad / 8);
        WgradParams wp;
        wp.slab = dpf; wp.slab_stride = Bpad * pe; wp.ld = (int)pe; wp.N_valid = (int)batch; wp.K_valid = (int)((J - 1) * 9 < pe ? (J - 1) * 9 : pe);
        FK_HIP_LAUNCH(gemm_wgrad(PREC_FP32, SHAPE_MID, g, wp, st));
    } else {
        // d pf = doff_hi pd_hi^T + doff_hi pd_lo^T + doff_lo pd_hi^T: three split-K launches into consecutive slab sets
        const char* pd_hi = (const char*)posedirs_bwd_packed + prow * Cpad * 4;
        const char* pd_lo = pd_hi + prow * Cpad * 2;
        const int kb = (int)(Cpad / 16);
        // 256-wide tiles when the batch allows them: the column count rounds to 256 then (the narrower slabs pay for more splits)
        int64_t pe_big = round_up(want_cols < 1 ? 1 : want_cols, 256);
        pe_big = pe_big > prow ? prow : pe_big;
        const int max_split = (int)(slab_budget / (3 * Bpad * pe_big) < 16 ? slab_budget / (3 * Bpad * pe_big) : 16);
        // The three terms are independent launches of (tiles x splits) workgroups each.  One after the other they left half the chip idle
        // at 4096 poses (16 tiles x 8 splits = 128 workgroups per launch, 3 x 109 us); side by side on three streams they share it:
        // the split count is then chosen for the three launches TOGETHER (3 x 16 x 4 = 192 workgroups of twice the K range).
        const bool par = body_tuning().lbs_bwd_terms_parallel;
        int kbig = lbs_bwd_big_ksplit(Bpad, pe_big, kb, max_split < 1 ? 1 : max_split, par ? 3 : 1);
        {   // (A/B: a forced split count, if it is a valid one -- whole two-k-block stages per split, inside the slab space)
            const int f = body_tuning().lbs_bwd_ksplit;
            if (kbig && f >= 1 && f <= max_split && kb % (2 * f) == 0 && kb / f >= 32) kbig = f;
        }
        if (kbig) pe = pe_big;
        else {
            pe = round_up(want_cols < 1 ? 1 : want_cols, 128);
            pe = pe > prow ? prow : pe;
        }
        const int tile = kbig ? 256 : 128;
        // (the narrower slabs of a column prefix allow more splits than the full width did, but never more than the slab space holds:
        //  only whole-stage splits are valid, so the bound goes INTO the search)
        const int64_t fit = slab_budget / (3 * Bpad * pe);
        const int k1 = kbig ? kbig : lbs_bwd_ksplit_bf16(kb, (Bpad / 128) * (pe / 128), (int)(fit < 1 ? 1 : (fit > 24 ? 24 : fit)));
        // (small batches, 128x128 tiles, 3 x 48 workgroups at 60 poses: side by side on three streams measured SLOWER -- 0.33 vs 0.22 ms per step of a
        //  60-frame motion-denoising loop, profiles/r06_md_small_ab.md: the event hand-overs cost more than the launches; they stay on one stream)
        const bool fork = par && kbig;
        // round 6: the terms d_off_hi pd_hi^T and d_off_hi pd_lo^T as ONE launch against the row-concatenated [pd_hi ; pd_lo] (2 pe rows: the
        // natural layout when every pose-feature row is wanted, the packed 256-row prefix panel when only the body is posed) -- the two
        // column tiles of a row panel run side by side on one XCD (panel_order) and d_off_hi leaves HBM once instead of twice
        // (256x256 tiles only: on the 128x128 tiles of the small batches two launches instead of three measured no gain at 60 frames and a loss at 480,
        //  profiles/r06_md_small_ab.md)
        const bool rowcat = kbig && body_tuning().lbs_bwd_rowcat && (pe == prow || (pe == 256 && prow > 256));
        const int nlaunch = rowcat ? 2 : 3;
        if (fork) {
            DP_TRY(side_streams_for_current_device(h));
            if (!h->bwd_side[0]) {
                for (hipStream_t* q : {&h->bwd_side[0], &h->bwd_side[1]}) DP_CHECK_HIP(hipStreamCreateWithFlags(q, hipStreamNonBlocking));
                for (hipEvent_t* e : {&h->ev_bwd_fork, &h->ev_bwd_join[0], &h->ev_bwd_join[1]}) DP_CHECK_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
            }
            DP_CHECK_HIP(hipEventRecord(h->ev_bwd_fork, st));
            for (int i = 0; i < nlaunch - 1; ++i) DP_CHECK_HIP(hipStreamWaitEvent(h->bwd_side[i], h->ev_bwd_fork, 0));
        }
        // (between fork and join nothing returns: a failed launch of one term still joins the side streams back into `st` -- the caller
        //  may free or reuse ws_bwd as soon as this call has returned -- and is reported afterwards)
        hipError_t term_err = hipSuccess;
        float* slabB = dpf + (int64_t)2 * k1 * Bpad * pe;               // (rowcat) the d_off_lo term's slabs behind the combined launch's
        for (int term = 0; term < nlaunch && term_err == hipSuccess; ++term) {
            hipStream_t ts = (fork && term > 0) ? h->bwd_side[term - 1] : st;
            GemmArgs g;
            std::memset(&g, 0, sizeof(g));
            g.w_stride_blocks = kb; g.n_cblk = (int)(Bpad / tile); g.ksplit = k1; g.seg_kblocks[0] = kb; g.nseg = 1; g.ktot_blocks = kb;
            g.panel_order = body_tuning().lbs_bwd_panel_order;
            WgradParams wp;
            wp.N_valid = (int)batch;
            if (rowcat) {
                const bool both = term == 0;
                g.W = both ? (const void*)doff_hi : (const void*)doff_lo;
                g.src[0] = both ? (pe == prow ? pd_hi : pd_lo + prow * Cpad * 2) : pd_hi;
                g.n_sblk = (int)((both ? 2 * pe : pe) / tile);
                wp.slab = both ? dpf : slabB;
                wp.ld = (int)(both ? 2 * pe : pe); wp.slab_stride = Bpad * (int64_t)wp.ld;
                wp.K_valid = both ? (int)(2 * pe) : (int)((J - 1) * 9 < pe ? (J - 1) * 9 : pe);      // (padded posedirs rows are zeros: their columns are written as such)
            } else {
                g.W = term == 2 ? (const void*)doff_lo : (const void*)doff_hi;
                g.src[0] = term == 1 ? pd_lo : pd_hi;
                g.n_sblk = (int)(pe / tile);
                wp.slab = dpf + (int64_t)term * k1 * Bpad * pe; wp.slab_stride = Bpad * pe; wp.ld = (int)pe;
                wp.K_valid = (int)((J - 1) * 9 < pe ? (J - 1) * 9 : pe);
            }
            term_err = gemm_wgrad(PREC_BF16, kbig ? SHAPE_BIG : SHAPE_MID, g, wp, ts);
        }
        if (fork) {
            for (int i = 0; i < nlaunch - 1; ++i) {
                DP_CHECK_HIP(hipEventRecord(h->ev_bwd_join[i], h->bwd_side[i]));
                DP_CHECK_HIP(hipStreamWaitEvent(st, h->ev_bwd_join[i], 0));
            }
        }
        FK_HIP_LAUNCH(term_err);
        ks = 3 * k1;
        if (rowcat) {
            const int64_t n4 = Bpad * pe / 4;
            hipLaunchKernelGGL(k_sum_slabs_rowcat, dim3((unsigned)(n4 < 256 * 2048 ? ceil_div(n4, 256) : 2048)), dim3(256), 0, st, (const float*)dpf, k1, (const float*)slabB, k1,
                               slabB, Bpad, (int)pe);
            FK_HIP_LAUNCH(hipGetLastError());
            dpf = slabB;
            ks = 1;                                                      // (already summed)
        }
    }
    {
        FkBwdArgs a;
        std::memset(&a, 0, sizeof(a));
        int first = 0;
        for (int i = 0; i < num_segments; ++i) {
            a.seg[i] = pose_segments_host[i];
            a.dseg[i] = d_pose_segments_host[i];
            a.seg_first[i] = first;
            a.seg_joints[i] = segment_joints_host[i];
            first += segment_joints_host[i];
        }
        DP_CHECK_ARG(first == J, "pose segments must cover all joints of the kinematic tree");
        a.nseg = num_segments; a.j_rest = j_rest; a.j_rest_batched = j_rest_batched; a.A = A; a.dA = dA; a.djoints = d_joints; a.ld_dj = d_joints_ld;
        if (ks > 1) {
            const int64_t n4 = Bpad * pe / 4;
            hipLaunchKernelGGL(k_sum_slabs, dim3((unsigned)(n4 < 256 * 2048 ? ceil_div(n4, 256) : 2048)), dim3(256), 0, st, dpf, Bpad * pe, ks);
            FK_HIP_LAUNCH(hipGetLastError());
        }
        a.dpf = dpf; a.dpf_slab = Bpad * pe; a.ldpf = pe; a.pf_cols = (int)pe; a.nsplit = 1; a.djrest = d_jrest; a.dG = dG; a.parents = nullptr; a.J = J;
        a.B = batch;
        if (batch <= fk_small_max()) {
            const dim3 grid((unsigned)batch);
            if (h->kind == 0) hipLaunchKernelGGL(k_fk_bwd_small<KinSMPL>, grid, dim3(64), 0, st, a);
            else if (h->kind == 1) hipLaunchKernelGGL(k_fk_bwd_small<KinSMPLH>, grid, dim3(64), 0, st, a);
            else hipLaunchKernelGGL(k_fk_bwd_small<KinSMPLX>, grid, dim3(64), 0, st, a);
        } else {
            const dim3 grid((unsigned)ceil_div(batch, 64));
            if (h->kind == 0) hipLaunchKernelGGL(k_fk_bwd<KinSMPL>, grid, dim3(64), 0, st, a);
            else if (h->kind == 1) hipLaunchKernelGGL(k_fk_bwd<KinSMPLH>, grid, dim3(64), 0, st, a);
            else hipLaunchKernelGGL(k_fk_bwd<KinSMPLX>, grid, dim3(64), 0, st, a);
        }
        FK_HIP_LAUNCH(hipGetLastError());
    }
    return DPOSER_OK;
}
