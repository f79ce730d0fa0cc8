// Device-side mirrors of the SDE scalars (reference lib/algorithms/advanced/sde_lib.py) and of the sigma lookup of
// ScoreModelFC.forward (model.py:159,194), shared by the elementwise kernels and the fused sampler epilogue.  fp32 operation
// order follows the reference so that `sigmas[(t * 999).long()]` indexes bit-exactly: every function body switches contraction
// off itself (the GEMM translation units are compiled with contraction allowed, and whether hipcc then fuses
// `m2b0 * t - db * (t * t)` depends on the code around it -- round 4 found the sampler's three forms one ulp apart in g at t = 0.334).
#pragma once
#include <cmath>
#include "common.h"

// ---- SDE description (reference lib/algorithms/advanced/sde_lib.py) --------------------------------
enum : int { SDE_SUBVP = 0, SDE_VP = 1, SDE_VE = 2 };
struct SdeCfg {
    int kind;
    int discrete = 0;        // the discrete score function: SDE_VE label round((T - t)(N - 1)) instead of sigma(t) (utils.py:175-178); SDE_VP label t (N - 1) and
                             // the DDPM table's std (utils.py:157-160)
    double beta_0, beta_1;   // python floats of the reference's constructor; VE: sigma_min, sigma_max (sde_lib.py:235-247)
    int N;
    float T;
};

struct SdeDev {
    int kind;
    float b0, db, m2b0;   // beta_0, (beta_1 - beta_0), -2*beta_0  (python doubles rounded to fp32)
    float dt, sqrt_mdt;   // -1/N, sqrt(1/N)
    float smin, ratio, gk;   // VE: sigma_min, sigma_max / sigma_min, sqrt(fp32(2 (ln sigma_max - ln sigma_min)))   sde_lib.py:260-264
    int disc;                // VE / VP: discrete score function
    float Tf, nm1;           // discrete: T and N - 1 as the fp32 scalars torch multiplies with
    float sd_disc;           // VP, discrete: sqrt_1m_alphas_cumprod[(t (N - 1)).long()] at the launch's shared t (make_sde_dev_at)
};
static inline SdeDev make_sde_dev(const SdeCfg& s) {
    SdeDev d;
    d.kind = s.kind;
    const double b0 = (double)(float)s.beta_0, b1 = (double)(float)s.beta_1;      // (sub-VP / VP: the fp32-rounded constructor arguments, as rounds 1-4 formed them)
    d.b0 = (float)b0;
    d.db = (float)(b1 - b0);
    d.m2b0 = (float)(-2.0 * b0);
    d.dt = (float)(-1.0 / (double)s.N);
    d.sqrt_mdt = (float)sqrt(1.0 / (double)s.N);
    d.smin = d.ratio = d.gk = 0.f;
    d.disc = (s.kind == SDE_VE || s.kind == SDE_VP) ? s.discrete : 0;
    d.Tf = s.T;
    d.nm1 = (float)(s.N - 1);
    d.sd_disc = 0.f;
    if (s.kind == SDE_VE) {
        d.smin = (float)s.beta_0;                                           // `self.sigma_min * tensor`: the python float enters as an fp32 scalar
        d.ratio = (float)(s.beta_1 / s.beta_0);                             // `(self.sigma_max / self.sigma_min) ** t`: python-float quotient, fp32 pow
        d.gk = sqrtf((float)(2.0 * (log(s.beta_1) - log(s.beta_0))));       // torch.sqrt(torch.tensor(2 * (np.log(smax) - np.log(smin)))): fp32 tensor, fp32 sqrt
    }
    return d;
}
// VPSDE.sqrt_1m_alphas_cumprod[k] (sde_lib.py:134-139): discrete_betas = linspace(beta_min / N, beta_max / N, N) as torch forms it in fp32 (from both
// ends: start + step i below the middle, end - step (N - 1 - i) above), alphas = 1 - betas, a running fp32 product, sqrt(1 - product)
static inline float sde_vp_sqrt_1m_alphas_cumprod(const SdeCfg& s, int k) {
    const int N = s.N;
    if (N < 1) return 0.f;
    k = k < 0 ? 0 : (k >= N ? N - 1 : k);
    const float start = (float)(s.beta_0 / (double)N), end = (float)(s.beta_1 / (double)N);
    const float step = N > 1 ? (end - start) / (float)(N - 1) : 0.f;
    float prod = 1.0f;
    for (int i = 0; i <= k; ++i) {
        const float beta = i < N / 2 ? start + step * (float)i : end - step * (float)(N - 1 - i);
        prod = prod * (1.0f - beta);
    }
    return sqrtf(1.0f - prod);
}
// the descriptor of a launch at ONE shared time t: for the discrete VP score function the table entry travels by value
static inline SdeDev make_sde_dev_at(const SdeCfg& s, float t) {
    SdeDev d = make_sde_dev(s);
    if (d.kind == SDE_VP && d.disc) d.sd_disc = sde_vp_sqrt_1m_alphas_cumprod(s, (int)(t * d.nm1));
    return d;
}
// sigma(t) of the VE SDE (sde_lib.py:260,267): the same expression on host (time-table labels) and device
__host__ __device__ __forceinline__ float sde_ve_sigma(float smin, float ratio, float t) {
#pragma clang fp contract(off)
    return smin * powf(ratio, t);
}
// label of the discrete VE score function (utils.py:176-178): `labels = sde.T - t; labels *= sde.N - 1; labels = torch.round(labels)` -- two fp32
// operations and a round-half-to-even, the same bits on host and device
__host__ __device__ __forceinline__ float sde_ve_discrete_label(float Tf, float nm1, float t) {
#pragma clang fp contract(off)
    return rintf((Tf - t) * nm1);
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

// Everything the kernels need of the forward SDE at one t, for all three kinds (the sub-VP / VP members are the functions above, evaluated
// in the order the call sites always used):
//   mc, sd  marginal_prob: mean = mc * x, std = sd        (VE: 1, sigma(t))
//   beta, g sde: drift = -1/2 beta x, diffusion g         (VE: 0, sigma(t) * sqrt(2 ln(sigma_max / sigma_min)))
//   label   what the network is conditioned on            (t * 999, utils.py:152; VE: sigma(t), utils.py:173, or round((T - t)(N - 1)), :176-178)
struct SdeAt { float mc, sd, beta, g, label, sd_score; };      // sd_score: the std the score is formed with (utils.py:155 / :160); sd: marginal_prob's
__device__ __forceinline__ SdeAt sde_at(const SdeDev& s, float t) {
#pragma clang fp contract(off)
    SdeAt r;
    if (s.kind == SDE_VE) {
        const float sig = sde_ve_sigma(s.smin, s.ratio, t);
        r.mc = 1.0f; r.sd = sig; r.beta = 0.0f; r.g = sig * s.gk; r.label = s.disc ? sde_ve_discrete_label(s.Tf, s.nm1, t) : sig;
        r.sd_score = sig;
    } else {
        const float lmc = sde_lmc(s, t);
        r.mc = expf(lmc); r.sd = sde_std(s, lmc); r.beta = sde_beta(s, t); r.g = sde_diffusion(s, t); r.label = t * 999.0f;
        r.sd_score = r.sd;
        if (s.kind == SDE_VP && s.disc) { r.label = t * s.nm1; r.sd_score = s.sd_disc; }      // utils.py:158-160
    }
    return r;
}
// score from the network output `model` (already divided by used_sigmas): -model / std for sub-VP / VP (utils.py:162), the output itself for VE (:180)
__device__ __forceinline__ float sde_score(const SdeDev& s, float model, float sd) {
#pragma clang fp contract(off)
    return s.kind == SDE_VE ? model : -model / sd;
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
