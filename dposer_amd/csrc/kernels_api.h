// Elementwise / layout / optimizer kernels of the score path (implemented in elementwise.hip).
#pragma once
#include "common.h"
#include "wgrad_batch.h"
#include "gemm_api.h"

// ---- weight packing -----------------------------------------------------------------------------
struct PackJob {
    int64_t dst_off;      // byte offset inside the packed workspace
    int64_t src_off;      // element offset inside the flat fp32 parameter buffer
    int ktot;             // columns of the destination FT matrix
    int koff;             // first destination column written by this job
    int rows_pad, kpad;   // extent written (multiples of 32 / KBS), zero-filled outside the source
    int rows_valid, cols_valid, ld;
    int trans;            // 0: dst[r][koff+k] = src[r*ld + k]     1: dst[r][koff+k] = src[k*ld + r]
    int f32;              // destination element type: 1 fp32, 0 bf16
    int split;            // bf16 destinations: 0 = bf16(v), 1 = high part bf16(v), 2 = low part bf16(v - float(bf16(v)))
};
constexpr int MAX_PACK_JOBS = 40;
struct PackJobs {
    PackJob job[MAX_PACK_JOBS];
    int n;
};
hipError_t launch_pack(const PackJobs& jobs, const float* flat, void* packed, hipStream_t st);
// dst[i] = flat[a_off[i/H] + i%H] + flat[b_off[..]]
struct BiasCatJobs {
    int64_t a_off[8], b_off[8];
    int n, H;
};
hipError_t launch_bias_cat(const BiasCatJobs& j, const float* flat, float* dst, hipStream_t st);

#include "sde_dev.h"   // SdeCfg, SdeDev

// ---- input preparation ---------------------------------------------------------------------------
struct PrepArgs {
    const float* x;        // [B][D] fp32 row-major
    const float* labels;   // [B] (per-sample t path) or null
    const float* freq;     // [E/2] positional frequencies, or fourier W
    void* xin;             // FT [Bpad][Dpad]
    void* emb;             // FT [Bpad][E] or null
    int64_t B, Bpad;
    int D, Dpad, E;
    int fourier;
    int f32;
};
hipError_t launch_prep_infer(const PrepArgs& a, hipStream_t st);

struct PrepTrainArgs {
    const float* x0;       // [B][D] clean (normalised) poses
    const float* t_in;     // [B] injected t or null (-> Philox STREAM_TRAIN_T)
    const float* z_in;     // [B][D] injected z or null (-> Philox STREAM_TRAIN_Z)
    const float* freq;
    void* xin;             // FT [Bpad][Dpad]  perturbed data
    void* emb;             // FT [Bpad][E]
    float* t_out;          // [Bpad]
    float* z_out;          // [Bpad][Dpad] fp32 row-major
    int64_t B, Bpad;
    int D, Dpad, E;
    int fourier, f32;
    SdeCfg sde;
    float eps;             // smallest t (1e-5)
    uint64_t seed;
    uint32_t step;
};
hipError_t launch_prep_train(const PrepTrainArgs& a, hipStream_t st);

// time embedding rows for a list of labels (labels == null: every row uses label0): emb FT32 [Npad][E]
hipError_t launch_time_embed(const float* labels, float label0, int64_t n, int64_t npad, const float* freq, int E, int fourier, float* emb_ft32, hipStream_t st);
// VE labels in place: t[i] -> sigma(t[i]) = sigma_min * (sigma_max / sigma_min)^t with the DEVICE's powf -- the value every kernel that perturbs with /
// divides by sigma(t) forms for itself (host libm's powf may differ by an ulp)
hipError_t launch_ve_labels(float* t_inout, int n, float smin, float ratio, hipStream_t st);

// ---- output stages ---------------------------------------------------------------------------------
struct OutModelArgs {      // ScoreModelFC.forward tail: res / used_sigmas   (model.py:189-196)
    const float* res;      // [Bpad][Cp]
    const float* labels;   // [B]
    const float* sigmas;   // [num_scales]
    float* out;            // [B][D]
    int64_t B;
    int D, Cp, num_scales;
    int scale_by_sigma, fourier;
};
hipError_t launch_out_model(const OutModelArgs& a, hipStream_t st);

struct EmUpdateArgs {      // EulerMaruyamaPredictor.update_fn + imputation (sampling.py:182-188, 416-420)
    const float* res;      // [Bpad][Cp] post_dense output of this step (null: no predictor step, only impute A + pack)
    float* x;              // [B][D] state, updated in place
    float* x_mean;         // [B][D]
    void* xin;             // FT [Bpad][Dpad]: next step's network input
    float* traj;           // [B][D] slot for this step or null
    float* x_ft;           // optional: the new state again as fp32 FT [Bpad][Dpad] (start of the fused sampler fast path)
    const float* sigmas;
    const float* obs;      // completion: observation [B][D] or null
    const float* mask;     // completion: mask [B][D]
    const float* z_pred;   // injected predictor noise [B][D] or null (-> Philox)
    const float* z_impB;   // injected imputation noise after the predictor, or null
    const float* z_impA;   // injected imputation noise before the NEXT predictor step, or null
    float t, t_next;       // current / next timestep (t_next < 0: last step, no look-ahead imputation)
    int64_t B, Bpad;
    int D, Dpad, Cp, num_scales;
    int f32;
    int scale_by_sigma;
    SdeCfg sde;
    uint64_t seed;
    uint32_t step;
};
hipError_t launch_em_update(const EmUpdateArgs& a, hipStream_t st);
// fp32 FT [Bpad][Dpad] -> row-major [B][D] (end of the fused sampler fast path); any of the two pairs may be null
hipError_t launch_ft_to_rows(const float* a_ft, float* a, const float* b_ft, float* b, int64_t B, int64_t Bpad, int D, int Dpad, hipStream_t st);

struct DenoiseArgs {       // one_step_denoise + prior loss (run/completion.py:105-149, run/smplify.py:69-107)
    const float* res;      // [Bpad][Cp]
    const float* x0;       // [B][D]
    const float* xt;       // [B][Dpad] perturbed data (fp32 row-major)
    const float* sigmas;
    float* x0_hat;         // [B][D] or null
    float* grad;           // [B][D] d loss / d x0 (analytic, x0_hat detached) or null
    float* loss_part;      // per-block partial sums
    float t;
    float inv_n;           // 1/n of the reduction (mean: 1/(B*D), sum/batch: 1/B)
    int weighted;
    int64_t B;
    int D, Dpad, Cp, num_scales, scale_by_sigma;
    SdeCfg sde;
};
hipError_t launch_denoise(const DenoiseArgs& a, int* nblocks, hipStream_t st);

struct PerturbSharedArgs { // x_t = mean + std*z at one shared t; writes xin (FT) and xt (fp32)
    const float* x0;
    const float* z_in;     // injected or null (-> Philox STREAM_PRIOR)
    void* xin;
    float* xt;             // [Bpad][Dpad]
    float t;
    int64_t B, Bpad;
    int D, Dpad, f32;
    SdeCfg sde;
    uint64_t seed;
    uint32_t step;
};
hipError_t launch_perturb_shared(const PerturbSharedArgs& a, hipStream_t st);

struct CompletionUpdateArgs {   // one optimisation step of DPoserComp.optimize (run/completion.py:167-207) after the network evaluation
    const float* res;      // [Bpad][Cp] model output at x_t
    const float* xt;       // [Bpad][Dpad] perturbed data
    const float* obs;      // [B][D]
    const float* mask;     // [B][D]
    const float* sigmas;
    float* x;              // [B][D] optimisation variable, updated in place
    float* m;              // [B][D] Adam first moment
    float* v;              // [B][D] Adam second moment
    float t;
    float inv_n;           // 1 / (B * D): both losses are means over the batch (completion.py:147, nn.MSELoss)
    float w_prior, w_data; // loss weights of this step (completion.py:151-155)
    int weighted;
    float step_size, one_minus_beta1, beta2, one_minus_beta2, bc2_sqrt, eps;   // torch.optim.Adam scalars of this step
    int64_t B;
    int D, Dpad, Cp, num_scales, scale_by_sigma;
    SdeCfg sde;
};
hipError_t launch_completion_update(const CompletionUpdateArgs& a, hipStream_t st);

struct LangevinArgs {           // LangevinCorrector.update_fn (sampling.py:282-302) around one network evaluation at a shared t
    const float* res;      // [Bpad][Cp] model output at x
    const float* noise;    // injected noise [B][D] or null (-> Philox STREAM_LANGEVIN)
    const float* sigmas;
    float* x;              // [B][D] state (update phase: in/out)
    float* x_mean;         // [B][D] out (update phase)
    void* xin;             // FT [Bpad][Dpad]: the updated state packed as the next network input (update phase) or null
    float* part;           // norms phase: per-block partial sums, [2][nblocks]
    const float* norm_sums;// update phase: [2] = sum_b ||grad_b||, sum_b ||noise_b|| over the GLOBAL batch
    float t, alpha, snr, inv_global_batch;
    int64_t B, Bpad;
    int D, Dpad, Cp, num_scales, scale_by_sigma, f32;
    SdeCfg sde;
    uint64_t seed;
    uint32_t step;
};
hipError_t launch_langevin_norms(const LangevinArgs& a, int* nblocks, hipStream_t st);
hipError_t launch_sum_partials2(const float* part, int n, float* out2, hipStream_t st);   // out2[k] = sum part[k*n .. k*n+n)
hipError_t launch_langevin_update(const LangevinArgs& a, hipStream_t st);
hipError_t launch_pack_rows(const float* x, void* xin, int64_t B, int64_t Bpad, int D, int Dpad, int f32, hipStream_t st);

struct DsmArgs {           // get_sde_loss_fn tail (losses.py:121-131) + d loss / d res
    const float* res;      // [Bpad][Cp]
    const float* t;        // [Bpad]
    const float* z;        // [Bpad][Dpad]
    const float* sigmas;
    void* dres;            // FT [Bpad][Cp]
    float* loss_part;      // per-block partial sums of the (already normalised) loss
    float* cs_part;        // optional [blocks][Cp]: per-block column sums of dres (post_dense bias gradient partials)
    int64_t B, Bpad;
    int D, Dpad, Cp, num_scales, scale_by_sigma, f32, fourier;
    float grad_scale;      // 1/(B*D) for reduce_mean
    SdeCfg sde;
};
hipError_t launch_dsm(const DsmArgs& a, int* nblocks, hipStream_t st);

struct SiLUBwdReduceArgs {  // dU = (sum over the k-splits of the time-branch dgrad) * act'(u); column sums of the stored dU per block
    const float* part;     // [nsplit][Spad][N] fp32 FT (EpiPartialFT)
    int nsplit;
    int64_t split_stride;
    const void* pre;       // u, FT [Spad][N]
    void* out;             // dU, FT [Spad][N]
    float* cs_part;        // [blocks][N]
    int N, act, f32;
    int64_t B, Spad;
};
hipError_t launch_silu_bwd_reduce(const SiLUBwdReduceArgs& a, int max_blocks, int* nblocks, hipStream_t st);
// the same reduction walked in FRAGMENT-TILE order (fp32 storage; bf16x3 mode at every batch size): every access is a coalesced 1-KiB run
// (the row-major walk above touches one 16-byte chunk per 1-KiB block); optionally writes dU's bf16 operand planes too.  Column sums:
// one partial row per row-block group (`*nblocks` rows of N floats in cs_part).
hipError_t launch_silu_bwd_reduce_ft(const SiLUBwdReduceArgs& a, void* out_hi, void* out_lo, int max_blocks, int* nblocks, hipStream_t st);

struct DresArgs {          // d res = d out / used_sigmas  (backward of model.py:192-194), FT store
    const float* dout;     // [B][D]
    const float* labels;   // [B]
    const float* sigmas;
    void* dres;            // FT [Bpad][Cp]
    int64_t B, Bpad;
    int D, Cp, num_scales, scale_by_sigma, fourier, f32;
};
hipError_t launch_dres_from_dout(const DresArgs& a, hipStream_t st);

// ---- layout helpers ---------------------------------------------------------------------------------
// out FT [Cpad][Spad] = transpose of in FT [Spad][C]
hipError_t launch_ft_transpose(int f32, const void* in, void* out, int64_t Spad, int C, hipStream_t st);
// fp32 FT [rows_pad][K] -> bf16 FT planes hi = bf16(x), lo = bf16(x - hi) of the same shape (bf16 x 3 precision mode)
hipError_t launch_split_ft32(const void* src, void* hi, void* lo, int64_t rows_pad, int K, hipStream_t st);
// part[chunk][c] = sum over the chunk's samples of in[s][c];  returns number of chunks.  `rider` (optional): a list of partials summed
// into out[0] by one extra block of the same launch (the DSM loss of the step: k_sum_partials' summation order, no launch of its own)
struct SumJob {
    const float* part;
    int n;
    float* out;
};
hipError_t launch_colsum(int f32, const void* in, float* part, int64_t Spad, int C, int* nchunks, hipStream_t st, const SumJob* rider = nullptr);

// ---- gradient finalisation and optimizer --------------------------------------------------------------
struct ReduceJob {
    int64_t dst_off;       // element offset in the flat gradient
    int64_t count;
    int64_t src_off;       // element offset in the scratch buffer
    int64_t src_stride;
    int nsrc;              // 0: fill [dst_off, dst_off + count) with zeros (parameters that never get a gradient);
                           // -1: ReduceJobs::alt[0] = sum of scratch[src_off .. src_off + count) in k_sum_partials' order (the step's loss)
};
constexpr int MAX_REDUCE_JOBS = 48;
struct ReduceJobs {
    ReduceJob job[MAX_REDUCE_JOBS];
    int n;
    float* alt;            // destination of the nsrc = -1 job (outside the flat gradient)
};
hipError_t launch_reduce_grads(const ReduceJobs& jobs, const float* scratch, float* flat_grad, hipStream_t st);
hipError_t launch_reduce_wgrad_tiles(const WgradBatchArgs& a, float* flat_grad, hipStream_t st);   // wgrad_batch.h: partial tiles -> flat gradient
// both of the above in one launch (the partial tiles of a lane launch + the small jobs behind it)
hipError_t launch_reduce_all(const WgradBatchArgs& a, const ReduceJobs& jobs, const float* scratch, float* flat_grad, hipStream_t st);
hipError_t launch_sum_partials(const float* part, int n, float* out, hipStream_t st);             // out[0] = sum(part[0..n))
hipError_t launch_sqnorm(const float* g, int64_t n, float* part, int* nblocks, hipStream_t st);   // partial sums of g^2

struct AdamArgs {          // losses.py:44-58 optimize_fn + torch.optim.Adam + ema.py:32-51, one pass
    float* p;
    const float* g;
    float* m;
    float* v;
    float* ema;            // may be null
    int64_t n;
    int64_t skip_lo[2], skip_hi[2];   // parameter ranges without gradient (Adam skipped, EMA still applied)
    float* sqnorm;         // device scalars: [0] sum of g^2 over the whole flat gradient (before grad_scale), [1] dropped-step counter
    const float* sq_part;  // n_part > 0: per-block partials of k_sqnorm, summed by the kernel itself (and published in sqnorm[0])
    int n_part;
    float grad_scale;      // applied to g before anything else (1/world_size)
    float grad_clip;       // < 0: disabled
    float step_size;       // lr / (1 - beta1^t)
    float one_minus_beta1, beta2, one_minus_beta2, eps;
    float bc2_sqrt;        // sqrt(1 - beta2^t)
    float ema_one_minus_decay;
    float weight_decay;    // torch.optim.Adam(weight_decay): grad += weight_decay * param (after the clip, before the moments); 0 = off
};
hipError_t launch_adam_ema(const AdamArgs& a, hipStream_t st);

// ---- the same update fused with the re-packing of the weights it changes (one pass over the optimizer state) --------------------
// A weight matrix W [R][K] (row-major, leading dimension ld) of the flat buffer and the packed copies the GEMMs read: every 64 x 64
// tile of W is updated by one block, which then writes the tile's part of each copy from LDS -- instead of k_adam_ema writing the
// parameters and k_pack / k_bias_cat of the NEXT step reading them back (33 MB read twice or three times, 2 launches).
struct AdamPackDst {
    int64_t off;          // byte offset in the packed workspace; < 0: unused
    int ktot, koff;       // destination FT matrix has ktot columns, this tensor starts at column koff
    int trans;            // 1: destination rows = columns of W (dgrad copies)
    int f32;              // destination element type
};
struct AdamPackTensor {
    int64_t src_off;      // element offset of W in the flat buffers
    int R, K, ld;
    int tile0;            // index of the tensor's first tile among all tiles of the launch
    AdamPackDst dst[3];
};
struct AdamPackElems {    // a range that is not a packed matrix (biases, GroupNorm affine, dead parameters, ...)
    int64_t off_a, off_b; // off_b >= 0: a second range of the same length updated by the same thread (dense bias | dense_t bias) ...
    int64_t cat_off;      // ... whose sums a[i] + b[i] go to the fp32 bias table at this float offset of the packed workspace
    int len;
    int block0;           // index of the range's first block among the element blocks
};
constexpr int ADAMPACK_MAX_TENSORS = 16, ADAMPACK_MAX_ELEMS = 28;
struct AdamPackArgs {
    AdamPackTensor tensor[ADAMPACK_MAX_TENSORS];      // (first member: fetched through the kernel-argument segment)
    AdamPackElems elems[ADAMPACK_MAX_ELEMS];
    int n_tensors, n_elems, n_tiles, n_elem_blocks;
    AdamArgs a;
    unsigned char* packed;
    int64_t bias_cat_off; // byte offset of the bias table in `packed`
    int write_through;    // 1: parameters / moments / EMA leave as write-through stores (no dirty L2 lines behind the launch)
};
hipError_t launch_adam_pack(const AdamPackArgs& a, hipStream_t st);
