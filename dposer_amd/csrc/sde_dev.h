// Device-side mirrors of the SDE scalars (reference lib/algorithms/advanced/sde_lib.py) and of the sigma lookup of
// ScoreModelFC.forward (model.py:159,194), shared by the elementwise kernels and the fused sampler epilogue.  fp32 operation
// order follows the reference so that `sigmas[(t * 999).long()]` indexes bit-exactly: every function body switches contraction
// off itself (the GEMM translation units are compiled with contraction allowed, and whether hipcc then fuses
// `m2b0 * t - db * (t * t)` depends on the code around it -- round 4 found the sampler's three forms one ulp apart in g at t = 0.334).
#pragma once
#include <cmath>
#include "common.h"

// ---- SDE description (reference lib/algorithms/advanced/sde_lib.py) --------------------------------
enum : int { SDE_SUBVP = 0, SDE_VP = 1 };
struct SdeCfg {
    int kind;
    float beta_0, beta_1;
    int N;
    float T;
};

struct SdeDev {
    int kind;
    float b0, db, m2b0;   // beta_0, (beta_1 - beta_0), -2*beta_0  (python doubles rounded to fp32)
    float dt, sqrt_mdt;   // -1/N, sqrt(1/N)
};
static inline SdeDev make_sde_dev(const SdeCfg& s) {
    SdeDev d;
    d.kind = s.kind;
    d.b0 = (float)(double)s.beta_0;
    d.db = (float)((double)s.beta_1 - (double)s.beta_0);
    d.m2b0 = (float)(-2.0 * (double)s.beta_0);
    d.dt = (float)(-1.0 / (double)s.N);
    d.sqrt_mdt = (float)sqrt(1.0 / (double)s.N);
    return d;
}
__device__ __forceinline__ float sde_lmc(const SdeDev& s, float t) {          // sde_lib.py:214
#pragma clang fp contract(off)
    return (-0.25f * (t * t)) * s.db - (0.5f * t) * s.b0;
}
__device__ __forceinline__ float sde_std(const SdeDev& s, float lmc) {         // :216 (subVP) / :155 (VP)
#pragma clang fp contract(off)
    const float v = 1.0f - expf(2.0f * lmc);
    return s.kind == SDE_SUBVP ? v : sqrtf(v);
}
__device__ __forceinline__ float sde_beta(const SdeDev& s, float t) {          // :207
#pragma clang fp contract(off)
    return s.b0 + t * s.db;
}
__device__ __forceinline__ float sde_diffusion(const SdeDev& s, float t) {     // :209-210 / :149
#pragma clang fp contract(off)
    const float beta = sde_beta(s, t);
    if (s.kind == SDE_VP) return sqrtf(beta);
    const float discount = 1.0f - expf(s.m2b0 * t - s.db * (t * t));
    return sqrtf(beta * discount);
}

// ------------------------------------------------------------------------------------------------
// output stages
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float used_sigma(const float* sigmas, int num_scales, float label, int fourier) {
    if (fourier) return label;                                  // model.py:152
    int idx = (int)label;                                       // model.py:159  t.long() truncates
    idx = idx < 0 ? 0 : (idx >= num_scales ? num_scales - 1 : idx);
    return sigmas[idx];
}
