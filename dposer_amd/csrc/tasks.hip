// Task loops that span both halves of the library (score network + body model) as ONE C call.
//
// dposer_motion_denoise_optimize -- MotionDenoise.optimize, reference run/motion_denoising.py:199-300 (loss weights :157-163,
// DPoser_loss :124-143, temporal / data terms :253-263): per Adam step
//     x_n   = normalise(pose)                                                     (lib/dataset/AMASS.py offline_normalize, axis-angle)
//     prior = sum(w (x_n - x0_hat)^2) / T at the step's shared t  -> d prior / d x_n  (dposer_prior_loss; x0_hat is detached)
//     v, J  = SMPL / SMPL-H / SMPL-X LBS(pose)                                     (dposer_lbs_forward)
//     temp  = mean_{t,v} || v[t] - v[t+1] ||,   data = mean_{t,j<n_obs} || J[t,j] - obs[t,j] ||   (dropped when not > 0, :261-263)
//     d pose = LBS^T (w_temp d temp / d v,  w_data d data / d J)  +  w_prior normalise^T (d prior / d x_n)   (dposer_lbs_backward)
//     torch.optim.Adam update of pose
// Nothing returns to the host between steps: the launches of all steps are queued from one loop (the Python loop around the
// same kernels spends ~3/4 of a 60-frame step in interpreter / autograd overhead and ~60 small torch kernels).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>

#include "common.h"
#include "rot_dev.h"

#define TK_HIP_LAUNCH(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
            return dposer_set_error(DPOSER_ERR_HIP, std::string(__func__) + ": " + #expr + ": " + hipGetErrorString(_e)); \
    } while (0)

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {      // 256 threads, deterministic
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// offline_normalize(from_axis=True) for the axis-angle representation (AMASS.py:126-137): mode 0 identity, 1 z-score (a = mean,
// b = std), 2 min-max (a = min, b = max)
__device__ __forceinline__ float md_norm(float p, int mode, const float* a, const float* b, int c) {
    if (mode == 1) return (p - a[c]) / b[c];
    if (mode == 2) return 2.0f * (p - a[c]) / (b[c] - a[c]) - 1.0f;
    return p;
}
__global__ void __launch_bounds__(256) k_md_normalize(const float* pose, const float* a, const float* b, int mode, float* xn, int64_t n, int D) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    xn[i] = md_norm(pose[i], mode, a, b, (int)(i % D));
}

// offline_normalize(from_axis=True) for rot_rep = 'rot6d' (AMASS.py:126-137 -> lib/utils/transforms.py:238-255): every joint's axis-angle
// becomes the first two columns of its rotation matrix, row-major (R00 R01 R10 R11 R20 R21), then the 6 J-dimensional statistics apply.
// One thread per (frame, joint).
__global__ void __launch_bounds__(256) k_md_normalize6d(const float* pose, const float* a, const float* b, int mode, float* xn, int64_t n_joints, int J) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_joints) return;
    const int j = (int)(i % J);
    const float* r = pose + i * 3;
    const Mat3 R = rodrigues(r[0], r[1], r[2]);
    const float six[6] = {R.m[0], R.m[1], R.m[3], R.m[4], R.m[6], R.m[7]};
    float* o = xn + i * 6;
#pragma unroll
    for (int e = 0; e < 6; ++e) o[e] = md_norm(six[e], mode, a, b, j * 6 + e);
}

// temporal term (:253-255): temp = mean over (T-1, V) of ||v[t] - v[t+1]||; d temp / d v[t] = (u_t - u_{t-1}) / ((T-1) V),
// u_t = (v[t] - v[t+1]) / ||v[t] - v[t+1]||  (0/0 = NaN for two identical vertices, as torch's sqrt backward gives)
// (F = frames per sequence: the frames of a batch of sequences are consecutive, neighbours never cross a sequence boundary)
// IEEE: the three quotients as IEEE divisions by the distance (what torch computes: x / sqrt(s)) instead of one v_rcp_f32 / v_rsq_f32 (1 ulp)
// and three products -- DPOSER_MD_IEEE_DIV=1 (A/B for config 5's tolerance: profiles/r06_cfg5_ieee_ab.md; the loop's own sensitivity
// to one ulp, profiles/r06_cfg5_sensitivity.md, is what bounds the agreement, not this)
template <bool IEEE> __global__ void __launch_bounds__(256) k_md_vert_grad(const float* verts, float* dverts, float* part, int F, int V, float c) {
    __shared__ float red[4];
    const int t = blockIdx.y;
    const int tf = t % F;
    const int v = blockIdx.x * 256 + threadIdx.x;
    float d_sum = 0.f;
    if (v < V) {
        const float* p = verts + ((int64_t)t * V + v) * 3;
        const float px = p[0], py = p[1], pz = p[2];
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (tf + 1 < F) {
            const float* q = p + (int64_t)V * 3;
            const float ax = px - q[0], ay = py - q[1], az = pz - q[2];
            const float d = sqrtf(ax * ax + ay * ay + az * az);
            if (IEEE) { gx = ax / d; gy = ay / d; gz = az / d; }
            else {
                const float inv = __builtin_amdgcn_rcpf(d);      // one v_rcp_f32 (1 ulp) instead of three IEEE divisions (~10 VALU each); 0 * inf = NaN as 0 / 0
                gx = ax * inv; gy = ay * inv; gz = az * inv;
            }
            d_sum = d;
        }
        if (tf > 0) {
            const float* q = p - (int64_t)V * 3;
            const float bx = q[0] - px, by = q[1] - py, bz = q[2] - pz;
            if (IEEE) { const float db = sqrtf(bx * bx + by * by + bz * bz); gx -= bx / db; gy -= by / db; gz -= bz / db; }
            else {
                const float inv = __builtin_amdgcn_rsqf(bx * bx + by * by + bz * bz);      // v_rsq_f32: this distance only feeds the gradient (rsq(0) = inf: 0 * inf = NaN)
                gx -= bx * inv; gy -= by * inv; gz -= bz * inv;
            }
        }
        float* o = dverts + ((int64_t)t * V + v) * 3;
        o[0] = c * gx; o[1] = c * gy; o[2] = c * gz;
    }
    const float tot = block_sum(d_sum, red);
    if (threadIdx.x == 0) part[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = tot;
}

// data term (:257-263) and the step's loss log.  One block: pass 1 = mean distance (decides whether the term is kept: finite
// and > 0, the reference's `if data_term > 0`), pass 2 = gradient rows.  Distances are clamped at 1e-18 before the division: a
// zero residual contributes the value ~0 and a zero gradient (sqrt'(0) = inf would poison the backward pass).
__global__ void __launch_bounds__(256) k_md_joint(const float* joints, int64_t ld, const float* obs, float* djoints, int T, int n_obs, float w_data,
                                                  const float* temp_part, int n_temp_part, float temp_scale, const float* prior_loss, float* log3) {
    // (n_temp_part = 0: the temporal term's distance sums do not exist yet -- k_md_temp_log writes log3[0] behind the LBS backward)
    __shared__ float red[4];
    // block = one sequence of T frames
    joints += (int64_t)blockIdx.x * T * ld; djoints += (int64_t)blockIdx.x * T * ld; obs += (int64_t)blockIdx.x * T * n_obs * 3;
    temp_part += (int64_t)blockIdx.x * n_temp_part;
    if (log3) log3 += 3 * (int64_t)blockIdx.x;
    const int n = T * n_obs;
    float acc = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int t = e / n_obs, j = e % n_obs;
        const float* a = joints + (int64_t)t * ld + j * 3;
        const float* o = obs + (int64_t)e * 3;
        const float dx = a[0] - o[0], dy = a[1] - o[1], dz = a[2] - o[2];
        acc += sqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-36f));
    }
    const float mean = block_sum(acc, red) / (float)n;
    const bool keep = isfinite(mean) && mean > 0.f;
    const float g = keep ? w_data / (float)n : 0.f;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int t = e / n_obs, j = e % n_obs;
        const float* a = joints + (int64_t)t * ld + j * 3;
        const float* o = obs + (int64_t)e * 3;
        const float dx = a[0] - o[0], dy = a[1] - o[1], dz = a[2] - o[2];
        const float s = dx * dx + dy * dy + dz * dz;
        float* dj = djoints + (int64_t)t * ld + j * 3;
        if (s > 1e-36f) {
            const float d = sqrtf(s);
            dj[0] = g * (dx / d); dj[1] = g * (dy / d); dj[2] = g * (dz / d);
        } else {
            dj[0] = 0.f; dj[1] = 0.f; dj[2] = 0.f;
        }
    }
    if (log3) {
        float tp = 0.f;
        for (int i = threadIdx.x; i < n_temp_part; i += 256) tp += temp_part[i];
        const float temp = block_sum(tp, red) * temp_scale;
        if (threadIdx.x == 0) { log3[0] = temp; log3[1] = keep ? mean : 0.f; log3[2] = prior_loss[0]; }
    }
}

// log3[0] of a step whose temporal term is formed inside the skinning backward (dposer_lbs_backward_temporal): the per-wave distance sums
// [frames][vb][4] arrive behind that kernel; ((p0 + p1) + p2) + p3 of an entry is k_md_vert_grad's block sum, the rest is k_md_joint's sum
__global__ void __launch_bounds__(256) k_md_temp_log(const float* part4, int n_temp_part, float temp_scale, float* log3) {
    __shared__ float red[4];
    part4 += (int64_t)blockIdx.x * n_temp_part * 4;
    float tp = 0.f;
    for (int i = threadIdx.x; i < n_temp_part; i += 256) {
        const float* p = part4 + (int64_t)i * 4;
        tp += p[0] + p[1] + p[2] + p[3];
    }
    const float temp = block_sum(tp, red) * temp_scale;
    if (threadIdx.x == 0) log3[3 * (int64_t)blockIdx.x] = temp;
}

// d pose = LBS gradient + w_prior * normalise^T(d prior / d x_n); torch.optim.Adam (single-tensor arithmetic, as k_completion_update)
struct MdUpdateArgs {
    float* pose; float* m; float* v;
    float* xn_next;      // axis-angle form, optional: the NEXT step's normalised pose (k_md_normalize's expression on the value just written: one launch less per step)
    const float* dpose; const float* gprior; const float* a; const float* b;
    int mode, D;
    int64_t n;
    float w_prior, step_size, one_minus_beta1, beta2, one_minus_beta2, bc2_sqrt, eps;
};
__global__ void __launch_bounds__(256) k_md_update(MdUpdateArgs u) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= u.n) return;
    const int c = (int)(i % u.D);
    float gp = u.gprior[i] * u.w_prior;
    if (u.mode == 1) gp = gp / u.b[c];
    else if (u.mode == 2) gp = (gp / (u.b[c] - u.a[c])) * 2.0f;
    const float g = u.dpose[i] + gp;
    float m = u.m[i], v = u.v[i];
    m = m + (g - m) * u.one_minus_beta1;
    v = v * u.beta2 + u.one_minus_beta2 * (g * g);
    const float denom = sqrtf(v) / u.bc2_sqrt + u.eps;
    const float p = u.pose[i] - u.step_size * (m / denom);
    u.pose[i] = p;
    u.m[i] = m;
    u.v[i] = v;
    if (u.xn_next) u.xn_next[i] = md_norm(p, u.mode, u.a, u.b, c);
}

// the same update for rot_rep = 'rot6d': the prior gradient arrives in the 6 J normalised coordinates; normalise^T, then the
// vector-Jacobian product of axis-angle -> (R00 R01 R10 R11 R20 R21) brings it to the joint's three pose parameters.  One thread per
// (frame, joint).
__global__ void __launch_bounds__(256) k_md_update6d(MdUpdateArgs u, int64_t n_joints, int J) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_joints) return;
    const int j = (int)(i % J);
    float g6[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) {
        const int c = j * 6 + e;
        float gp = u.gprior[i * 6 + e] * u.w_prior;
        if (u.mode == 1) gp = gp / u.b[c];
        else if (u.mode == 2) gp = (gp / (u.b[c] - u.a[c])) * 2.0f;
        g6[e] = gp;
    }
    const float dR[9] = {g6[0], g6[1], 0.f, g6[2], g6[3], 0.f, g6[4], g6[5], 0.f};
    float* q = u.pose + i * 3;
    float gq[3];
    rodrigues_bwd(q[0], q[1], q[2], dR, gq);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int64_t k = i * 3 + c;
        const float g = u.dpose[k] + gq[c];
        float m = u.m[k], v = u.v[k];
        m = m + (g - m) * u.one_minus_beta1;
        v = v * u.beta2 + u.one_minus_beta2 * (g * g);
        const float denom = sqrtf(v) / u.bc2_sqrt + u.eps;
        q[c] = q[c] - u.step_size * (m / denom);
        u.m[k] = m;
        u.v[k] = v;
    }
}

struct Scratch {
    float *xn, *gprior, *loss1, *verts, *joints, *dverts, *djoints, *dpose, *part;
    int64_t bytes;
};
Scratch layout(char* base, int64_t T, int D, int V, int n_joint_rows, int n_part) {
    Scratch s;
    char* p = base;
    auto take = [&](int64_t nfloat) { float* r = (float*)p; p += round_up(nfloat * 4, 256); return r; };
    s.xn = take(T * D * 2); s.gprior = take(T * D * 2); s.loss1 = take(64);      // (2 D: the 6-D representation of rot_rep = 'rot6d')
    s.verts = take(T * V * 3); s.joints = take(T * n_joint_rows * 3);
    s.dverts = take(T * V * 3); s.djoints = take(T * n_joint_rows * 3);
    s.dpose = take(T * D); s.part = take((int64_t)n_part * 4);      // (x 4: per-wave sums when the temporal term is formed inside the skinning backward)
    s.bytes = p - base;
    return s;
}

}   // namespace

extern "C" int64_t dposer_motion_denoise_scratch_bytes(int64_t frames, int32_t pose_dim, int32_t num_vertices, int32_t joint_rows) {
    if (frames <= 0 || pose_dim <= 0 || num_vertices <= 0 || joint_rows <= 0) return -1;
    const int n_part = (int)(ceil_div(num_vertices, 256) * frames);
    return layout(nullptr, frames, pose_dim, num_vertices, joint_rows, n_part).bytes;
}

extern "C" int dposer_motion_denoise_optimize(const dposer_motion_denoise_args* a, void* stream) {
    DP_RANGE();
    DP_CHECK_ARG(a, "null argument");
    DP_CHECK_ARG(a->net && a->flat_params && a->packed && a->net_ws && a->sde && a->freq && a->sigmas, "null score-network argument");
    DP_CHECK_ARG(a->body && a->lbs_ws_fwd && a->lbs_ws_bwd && a->posedirs_packed && a->posedirs_bwd_packed && a->j_rest && a->v_shaped && a->skin_idx &&
                     a->skin_w && a->joint_ptr && a->joint_vidx && a->joint_w && a->segment_joints_host, "null body-model argument");
    DP_CHECK_ARG(a->pose && a->adam_m && a->adam_v && a->joints_obs && a->scratch, "null problem argument");
    DP_CHECK_ARG(a->t_host && a->w_temp_host && a->w_data_host && a->w_prior_host && a->n_steps >= 0, "null / bad schedule");
    const int64_t F = a->frames_per_sequence > 0 ? a->frames_per_sequence : a->frames;
    DP_CHECK_ARG(F >= 2 && a->frames >= F && a->frames % F == 0, "frames must be a whole number of sequences of >= 2 frames (the temporal term couples neighbours)");
    const int64_t n_seq = a->frames / F;
    DP_CHECK_ARG(a->frames <= 65535, "at most 65535 frames per call (one grid row per frame in the skinning kernels)");
    DP_CHECK_ARG(a->num_segments >= 1 && a->num_segments <= 8 && a->body_segment >= 0 && a->body_segment < a->num_segments, "bad pose segments");
    DP_CHECK_ARG(a->norm_mode == 0 || ((a->norm_mode == 1 || a->norm_mode == 2) && a->norm_a && a->norm_b), "bad normaliser");
    DP_CHECK_ARG(a->num_vertices > 0 && a->num_joints > 0 && a->joint_rows >= a->num_joints && a->n_obs_joints >= 1 && a->n_obs_joints <= a->num_joints,
                 "bad body-model sizes");
    DP_CHECK_ARG(((uintptr_t)a->scratch & 255) == 0, "scratch must be 256-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int64_t T = a->frames;
    const int D = a->segment_joints_host[a->body_segment] * 3;
    const int V = a->num_vertices;
    const int vb = (int)ceil_div(V, 256);
    const int n_part = vb * (int)T;
    Scratch s = layout((char*)a->scratch, T, D, V, a->joint_rows, n_part);
    const int64_t n = T * D;
    const bool rot6d = a->rot6d != 0;                       // the network lives in the 6-D representation: 2 D coordinates per frame
    const int64_t n_net = rot6d ? 2 * n : n;
    {   // the network's input width (pre_dense.weight is [H, D_net], its bias [H]) must be the representation's: the noise and the prior
        // gradient are indexed with it
        const int64_t wn = dposer_scorefc_tensor_numel(a->net, 0), bn = dposer_scorefc_tensor_numel(a->net, 1);
        DP_CHECK_ARG(wn > 0 && bn > 0 && wn / bn == (rot6d ? 2 * D : D), "the score network's input width is not the pose representation's (3 J axis-angle, 6 J with rot6d)");
    }

    const float* segs[8];
    float* dsegs[8];
    for (int i = 0; i < 8; ++i) { segs[i] = nullptr; dsegs[i] = nullptr; }
    segs[a->body_segment] = a->pose;
    dsegs[a->body_segment] = s.dpose;
    // d joints: only the observed joints' columns are ever non-zero
    TK_HIP_LAUNCH(hipMemsetAsync(s.djoints, 0, (size_t)T * a->joint_rows * 3 * 4, st));

    // DPOSER_MD_FUSED_TEMPORAL = 0 / 1 forces the two-kernel / fused form of skinning + temporal gradient (default: fused from
    // DPOSER_MD_FUSED_TEMPORAL_MIN_SEQ = 4 sequences per call: a workgroup of the fused kernel walks >= 10 frames one after the other;
    // measured 0.60 vs 0.67 ms per step at 8 sequences of 60 frames, a tie at one)
    bool fused_temporal = a->skin_k == 4 && F <= 4096 && n_seq <= 16384;
    // DPOSER_MD_FUSED_TEMPORAL = 2 (default from 960 frames per call, where dposer_lbs_temporal_in_backward_ok): the temporal term's
    // gradient is formed INSIDE the skinning backward from the forward's transforms and offsets -- no skinning kernel, no vertices and no
    // vertex gradient in HBM (k_skin_temporal wrote 967 MB at 7680 frames that k_skin_bwd_mfma read back)
    // (measured, profiles/r06_md_ab.md: 2.63 -> 2.51 ms per step at 7680 frames, 0.83 -> 0.79 at 1920, a loss at 480 -- the backward kernel is
    //  VALU-bound once it skins three frames per pose -- hence DPOSER_MD_TEMPORAL_IN_BWD_MIN_FRAMES = 960)
    bool temporal_in_backward = dposer_lbs_temporal_in_backward_ok(a->body, a->skin_k, T) != 0;
    {
        const char* m = getenv("DPOSER_MD_TEMPORAL_IN_BWD_MIN_FRAMES");
        const char* e = getenv("DPOSER_MD_FUSED_TEMPORAL");
        if (!(e && e[0] == '2') && T < (m ? atoll(m) : 960)) temporal_in_backward = false;
    }
    bool ieee_div = false;
    { const char* e = getenv("DPOSER_MD_IEEE_DIV"); ieee_div = e && e[0] == '1'; }
    if (ieee_div) { fused_temporal = false; temporal_in_backward = false; }      // (the A/B switch exists in the two-kernel form only)
    {
        const char* e = getenv("DPOSER_MD_FUSED_TEMPORAL");
        const char* m = getenv("DPOSER_MD_FUSED_TEMPORAL_MIN_SEQ");
        if (e && e[0] == '0') { fused_temporal = false; temporal_in_backward = false; }
        else if (e && e[0] == '1') temporal_in_backward = false;
        else if (!(e && e[0] == '2')) fused_temporal = fused_temporal && n_seq >= (m ? atoll(m) : 4);
    }

    // time-bias rows of all steps: two small GEMMs once instead of per step
    if (a->n_steps > 0) DP_TRY(dposer_prior_table_build_sde(a->net, a->flat_params, a->packed, a->net_ws, a->sde, a->t_host, a->n_steps, a->freq, T, stream));
    for (int k = 0; k < a->n_steps; ++k) {
        if (rot6d) hipLaunchKernelGGL(k_md_normalize6d, dim3((unsigned)ceil_div(n / 3, 256)), dim3(256), 0, st, (const float*)a->pose, a->norm_a, a->norm_b, a->norm_mode, s.xn, n / 3, D / 3);
        else if (k == 0) hipLaunchKernelGGL(k_md_normalize, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, (const float*)a->pose, a->norm_a, a->norm_b, a->norm_mode, s.xn, n, D);
        // (axis-angle, k > 0: the previous step's update kernel has written s.xn already)
        TK_HIP_LAUNCH(hipGetLastError());
        DP_TRY(dposer_prior_loss_tabled(a->net, a->flat_params, a->packed, a->net_ws, a->sde, s.xn, a->noise ? a->noise + (int64_t)k * n_net : nullptr,
                                        a->t_host[k], k, a->n_steps, a->weighted, 1.0f / (float)F, nullptr, s.gprior, s.loss1, a->seed,
                                        a->step0 + (uint32_t)k, a->sigmas, T, stream));
        const float c_temp = a->w_temp_host[k] / ((float)(F - 1) * (float)V);
        if (temporal_in_backward) {
            // FK + pose-blend offsets only: the skinned vertices exist in registers of the backward kernel, nowhere else
            DP_TRY(dposer_lbs_forward_front(a->body, a->lbs_ws_fwd, a->posedirs_packed, segs, a->segment_joints_host, a->num_segments, a->j_rest,
                                            a->rest_batched, nullptr, s.joints, T, stream));
        } else if (fused_temporal) {
            // the vertices only feed the temporal term: skinning and that term's gradient in one pass, no vertices in HBM (the joints
            // beyond the kinematic tree -- extra vertices, landmarks -- are not formed: the data term reads observed tree joints only)
            DP_TRY(dposer_lbs_forward_temporal_grad(a->body, a->lbs_ws_fwd, a->posedirs_packed, segs, a->segment_joints_host, a->num_segments, a->j_rest,
                                                    a->rest_batched, a->v_shaped, a->rest_batched, a->skin_idx, a->skin_w, a->skin_k, nullptr, F, c_temp,
                                                    s.dverts, s.part, s.joints, T, stream));
        } else {
            DP_TRY(dposer_lbs_forward(a->body, a->lbs_ws_fwd, a->posedirs_packed, segs, a->segment_joints_host, a->num_segments, a->j_rest, a->rest_batched,
                                      a->v_shaped, a->rest_batched, a->skin_idx, a->skin_w, a->skin_k, nullptr, nullptr, nullptr, nullptr,      // (the data term reads tree joints only)
                                      s.verts, s.joints, T, stream));
            if (ieee_div) hipLaunchKernelGGL(k_md_vert_grad<true>, dim3((unsigned)vb, (unsigned)T), dim3(256), 0, st, (const float*)s.verts, s.dverts, s.part, (int)F, V, c_temp);
            else hipLaunchKernelGGL(k_md_vert_grad<false>, dim3((unsigned)vb, (unsigned)T), dim3(256), 0, st, (const float*)s.verts, s.dverts, s.part, (int)F, V, c_temp);
            TK_HIP_LAUNCH(hipGetLastError());
        }
        hipLaunchKernelGGL(k_md_joint, dim3((unsigned)n_seq), dim3(256), 0, st, (const float*)s.joints, (int64_t)a->joint_rows * 3, a->joints_obs, s.djoints,
                           (int)F, a->n_obs_joints, a->w_data_host[k], (const float*)s.part, temporal_in_backward ? 0 : vb * (int)F,
                           1.0f / ((float)(F - 1) * (float)V), (const float*)s.loss1, a->loss_log ? a->loss_log + 3 * (int64_t)k * n_seq : nullptr);
        TK_HIP_LAUNCH(hipGetLastError());
        if (temporal_in_backward) {
            DP_TRY(dposer_lbs_backward_temporal(a->body, a->lbs_ws_fwd, a->lbs_ws_bwd, a->posedirs_bwd_packed, segs, a->segment_joints_host, a->num_segments,
                                                a->j_rest, a->rest_batched, a->v_shaped, a->rest_batched, a->skin_idx, a->skin_w, a->skin_k, a->joint_ptr,
                                                a->joint_vidx, a->joint_w, F, c_temp, s.part, s.djoints, (int64_t)a->joint_rows * 3, dsegs, T, stream));
            if (a->loss_log) {
                hipLaunchKernelGGL(k_md_temp_log, dim3((unsigned)n_seq), dim3(256), 0, st, (const float*)s.part, vb * (int)F, 1.0f / ((float)(F - 1) * (float)V),
                                   a->loss_log + 3 * (int64_t)k * n_seq);
                TK_HIP_LAUNCH(hipGetLastError());
            }
        } else
        DP_TRY(dposer_lbs_backward(a->body, a->lbs_ws_fwd, a->lbs_ws_bwd, a->posedirs_bwd_packed, segs, a->segment_joints_host, a->num_segments, a->j_rest,
                                   a->rest_batched, a->v_shaped, a->rest_batched, a->skin_idx, a->skin_w, a->skin_k, a->joint_ptr, a->joint_vidx, a->joint_w,
                                   s.dverts, s.djoints, (int64_t)a->joint_rows * 3, dsegs, nullptr, nullptr, T, stream));
        // torch.optim.Adam scalars of optimiser step adam_step0 + k + 1 (python doubles, rounded once)
        const double stepno = (double)a->adam_step0 + k + 1;
        const double bc1 = 1.0 - std::pow(a->beta1, stepno), bc2 = 1.0 - std::pow(a->beta2, stepno);
        MdUpdateArgs u;
        u.xn_next = (!rot6d && k + 1 < a->n_steps) ? s.xn : nullptr;
        u.pose = a->pose; u.m = a->adam_m; u.v = a->adam_v; u.dpose = s.dpose; u.gprior = s.gprior; u.a = a->norm_a; u.b = a->norm_b;
        u.mode = a->norm_mode; u.D = D; u.n = n; u.w_prior = a->w_prior_host[k];
        u.step_size = (float)(a->lr / bc1); u.one_minus_beta1 = (float)(1.0 - a->beta1); u.beta2 = (float)a->beta2;
        u.one_minus_beta2 = (float)(1.0 - a->beta2); u.bc2_sqrt = (float)std::sqrt(bc2); u.eps = (float)a->eps;
        if (rot6d) hipLaunchKernelGGL(k_md_update6d, dim3((unsigned)ceil_div(n / 3, 256)), dim3(256), 0, st, u, n / 3, D / 3);
        else hipLaunchKernelGGL(k_md_update, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, u);
        TK_HIP_LAUNCH(hipGetLastError());
    }
    return DPOSER_OK;
}
