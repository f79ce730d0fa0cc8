/*
 * dposer_hip.h -- C ABI of libdposer_hip.so: the MI355X (gfx950) implementation of DPoser's
 * diffusion hot path.
 *
 * The reference (moonbow721/DPoser) has NO native boundary: the path sits behind Python call
 * sites.  Each entry point below therefore names the reference *Python*
 * function it replaces (paths relative to the reference root); dposer_amd/ binds them with
 * ctypes (dposer_amd/_C.py) and INTEGRATION.md shows the binding a reference maintainer adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name ends in _host;
 *   - matrices are fp32 row-major exactly as the reference's torch tensors ([B, D] poses, ...);
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *     kernels are only enqueued, never synchronised, no allocation happens inside a call;
 *   - scratch memory is caller-owned: query *_bytes(), allocate once (e.g. a torch uint8
 *     tensor), pass the pointer;
 *   - return value: 0 = ok, <0 = error (DPOSER_ERR_*), message via dposer_last_error().
 */
#ifndef DPOSER_HIP_H
#define DPOSER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPOSER_ABI_VERSION 1

enum {
    DPOSER_OK = 0,
    DPOSER_ERR_BAD_ARG = -1,
    DPOSER_ERR_UNSUPPORTED = -2,
    DPOSER_ERR_HIP = -3,
    DPOSER_ERR_WORKSPACE = -4
};

int dposer_abi_version(void);
const char* dposer_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Score network  (lib/algorithms/advanced/model.py:93-196  class ScoreModelFC)
 * ---------------------------------------------------------------------------------------- */
typedef struct dposer_scorefc_s* dposer_scorefc_t;

/* DPOSER_PREC_BF16X3: reference-precision results ON the bf16 matrix pipe -- activations and weights are split into two bf16 terms
 * (x = hi + lo, 16 mantissa bits), every GEMM accumulates hi*hi + lo*hi + hi*lo in fp32 (three v_mfma_f32_32x32x16_bf16 per product
 * instead of sixteen-times-slower exact-fp32 MFMAs), everything outside the GEMMs (GroupNorm, SiLU, loss, optimizer, stored
 * activations) is the fp32 mode's.  hidden_dim 1024 / swish. */
enum { DPOSER_PREC_BF16 = 0, DPOSER_PREC_FP32 = 1, DPOSER_PREC_BF16X3 = 2 };
enum { DPOSER_EMB_POSITIONAL = 0, DPOSER_EMB_FOURIER = 1 };
/* config.model.nonlinearity (model.py:54-66): swish = SiLU; lrelu = LeakyReLU(0.2); elu = ELU(alpha = 1).  Swish runs on every
 * tiling; the other three on the 128-wide tilings (any batch), hidden_dim 1024 only */
enum { DPOSER_ACT_SWISH = 0, DPOSER_ACT_ELU = 1, DPOSER_ACT_RELU = 2, DPOSER_ACT_LRELU = 3 };
/* DPOSER_SDE_VE (sde_lib.py:234-292): beta_min / beta_max of dposer_sde_desc carry sigma_min / sigma_max; the network is conditioned on
 * sigma(t) (continuous VE score function, utils.py:164-181) and its output is the score itself.
 * DPOSER_SDE_VE_DISCRETE: the same SDE with the DISCRETE score function (get_score_fn(..., continuous=False), utils.py:175-181): the network is
 * conditioned on the label round((T - t) (N - 1)) -- an index into `sigmas` for scale_by_sigma and the argument of the positional embedding --
 * everything else as DPOSER_SDE_VE.  Shared-t evaluation only (sampler, Langevin step, prior loss and the two fitting loops): the training
 * loss of a discrete model is the legacy SMLD loss, the probability-flow ODE is always continuous; those entry points refuse this kind.
 * DPOSER_SDE_VP_DISCRETE: the VP SDE with its discrete score function (utils.py:157-162): label t (N - 1) and
 * score = -model / sqrt_1m_alphas_cumprod[label.long()] with the DDPM table of sde_lib.py:134-139 for THIS N (linspace(beta_min / N, beta_max / N, N),
 * cumulative product, in fp32); perturbation and reverse SDE stay the continuous ones.  Shared-t evaluation only, like the VE one. */
enum { DPOSER_SDE_SUBVP = 0, DPOSER_SDE_VP = 1, DPOSER_SDE_VE = 2, DPOSER_SDE_VE_DISCRETE = 3, DPOSER_SDE_VP_DISCRETE = 4 };

typedef struct {
    int32_t data_dim;        /* n_poses * pose_dim: 63 (axis-angle) or 126 (rot6d), 1..512  model.py:109 */
    int32_t hidden_dim;      /* 512, 1024 (the shipped configuration) or 2048: GroupNorm(32, H) groups of 16 / 32 / 64 channels  model.py:112 */
    int32_t embed_dim;       /* multiple of 128 */
    int32_t n_blocks;        /* 1..3 */
    int32_t embedding;       /* DPOSER_EMB_*      config.model.embedding_type  model.py:116-122 */
    int32_t scale_by_sigma;  /* config.model.scale_by_sigma  model.py:192 */
    int32_t num_scales;      /* length of the `sigmas` buffer  model.py:128 */
    int32_t precision;       /* DPOSER_PREC_*: bf16 MFMA (throughput), fp32 MFMA (parity) or bf16 x 3 (parity on the bf16 pipe) */
    float dropout_p;         /* config.model.dropout  model.py:113 */
    int32_t activation;      /* DPOSER_ACT_*: config.model.nonlinearity (model.py:54-66 get_act); swish is the shipped one */
} dposer_scorefc_desc;

typedef struct {             /* lib/algorithms/advanced/sde_lib.py:122-231 */
    int32_t kind;            /* DPOSER_SDE_* */
    int32_t N;               /* sde.N (mutable after construction in the reference) */
    double beta_min, beta_max, T;
} dposer_sde_desc;

int dposer_scorefc_create(const dposer_scorefc_desc* desc, dposer_scorefc_t* out);
void dposer_scorefc_destroy(dposer_scorefc_t h);

/* Flat parameter buffer: all tensors of ScoreModelFC.parameters() (model.py:98-139), in that
 * order, contiguous fp32.  The Python module keeps its nn.Parameters as views into it. */
int64_t dposer_scorefc_num_params(dposer_scorefc_t h);
int32_t dposer_scorefc_num_tensors(dposer_scorefc_t h);
int64_t dposer_scorefc_tensor_offset(dposer_scorefc_t h, int32_t idx);
int64_t dposer_scorefc_tensor_numel(dposer_scorefc_t h, int32_t idx);
/* ranges of the flat buffer that never receive a gradient (pre_dense_cond, gauss_proj.W) */
int32_t dposer_scorefc_nograd_ranges(dposer_scorefc_t h, int64_t lo[2], int64_t hi[2]);

/* Re-tile the fp32 master weights into MFMA-fragment order (bf16 or fp32) for the forward
 * GEMMs and, when with_backward != 0, transposed copies for dgrad. */
int64_t dposer_scorefc_packed_bytes(dposer_scorefc_t h, int32_t with_backward);
int dposer_scorefc_pack(dposer_scorefc_t h, const float* flat_params, void* packed, int32_t with_backward, void* stream);
/* Re-reads the A/B environment switches of the score path (DPOSER_BIG_MIN_BATCH, DPOSER_GNBWD_BIG, DPOSER_WGRAD_BIG, DPOSER_WGRAD_TR,
 * DPOSER_WGRAD_STREAM, DPOSER_WGRAD_BATCHED, DPOSER_SMALL_TILE_MAX, DPOSER_WGRAD_LAYER_LANES, DPOSER_WGRAD_GROUPS, DPOSER_FINAL_SMALL_MAX,
 * DPOSER_SAMPLER_PERSISTENT[_MIN], DPOSER_ADAM_WT, DPOSER_DSM_FUSED, DPOSER_SILU_SPLIT_MAX); they are otherwise read ONCE per
 * process (first use), never per call. */
void dposer_scorefc_tuning_reload(void);
/* TEST HOOK: dropout keep decisions of every site as bytes [n_layers][batch][hidden_dim] (device memory, NULL = back to the Philox
 * streams), used by the training epilogues of calls with exactly this batch size (hidden_dim 1024).  It lets a parity test run the
 * masks the reference drew with torch's generator through the fused training step. */
int dposer_scorefc_debug_set_dropout_masks(dposer_scorefc_t h, const unsigned char* keep, int64_t batch);

enum { DPOSER_WS_INFER = 0, DPOSER_WS_SHARED_T = 1, DPOSER_WS_TRAIN = 2 };
int64_t dposer_scorefc_workspace_bytes(dposer_scorefc_t h, int64_t batch, int32_t mode, int32_t n_steps);

/* ScoreModelFC.forward(batch, t) in eval mode -- model.py:141-196.
 *   x [B, D], labels [B] (what the reference calls `t` inside the model, i.e. t*999 from
 *   get_score_fn utils.py:152), freq [E/2] = exp(arange(E/2) * -ln(1e4)/(E/2-1)) (model.py:39-46)
 *   or gauss_proj.W for the fourier embedding, sigmas [num_scales] (model.py:128), out [B, D]. */
int dposer_scorefc_forward(dposer_scorefc_t h, const float* flat_params, const void* packed, void* ws,
                           const float* x, const float* labels, const float* freq, const float* sigmas,
                           float* out, int64_t batch, void* stream);

/* pc_sampler with the EulerMaruyama predictor, corrector 'none', optional completion imputation
 * -- lib/algorithms/advanced/sampling.py:177-188, 375-468 (score via utils.py:141-163, reverse
 * SDE sde_lib.py:75-109).  Runs loop indices [start_step, sde.N) on-device:
 *   x [B, D] in: initial state (prior sample or z), out: final x;  x_mean [B, D] out;
 *   timesteps_host [sde.N] = torch.linspace(sde.T, eps, sde.N) (sampling.py:449), HOST pointer;
 *   observation/mask [B, D] or NULL (args.task == 'completion', sampling.py:416-420);
 *   noise [n_steps][k][B][D] injected draws in the reference's draw order per step
 *     (k = 1: predictor z;  k = 3 with completion: impute-after-corrector, z, impute-after-predictor)
 *     or NULL -> in-kernel Philox4x32-10 keyed by `seed`;
 *   traj [ceil(n_steps/traj_stride)][B][D] or NULL: state after steps start+traj_stride-1, ...
 * Both time embeddings (here and in every shared-t entry point below): `freq` = the positional frequencies, or gauss_proj.W for
 * DPOSER_EMB_FOURIER (embedding of log(labels), output divided by the labels: model.py:152-155,192-194). */
int dposer_em_sampler(dposer_scorefc_t h, const float* flat_params, const void* packed, void* ws,
                      const dposer_sde_desc* sde, float* x, float* x_mean, const float* timesteps_host,
                      int32_t start_step, const float* observation, const float* mask, const float* noise,
                      uint64_t seed, float* traj, int32_t traj_stride, const float* freq, const float* sigmas,
                      int64_t batch, void* stream);

/* DPoser prior: perturb -> one_step_denoise -> weighted L2 -- run/completion.py:105-149,
 * run/smplify.py:69-107, run/motion_denoising.py:99-143.  All samples share time t.
 *   x0 [B, D]; z [B, D] injected noise or NULL; x0_hat [B, D] or NULL; grad [B, D] = d loss/d x0
 *   (x0_hat is detached in the reference) or NULL; loss [1];
 *   inv_n = 1/(B*D) for torch.mean (completion.py:147), 1/batch_size for smplify.py:105. */
int dposer_prior_loss(dposer_scorefc_t h, const float* flat_params, const void* packed, void* ws,
                      const dposer_sde_desc* sde, const float* x0, const float* z, float t, int32_t weighted,
                      float inv_n, float* x0_hat, float* grad, float* loss, uint64_t seed, uint32_t step,
                      const float* freq, const float* sigmas, int64_t batch, void* stream);

/* The same evaluation with the time-bias rows of MANY steps built once: dposer_prior_table_build(t_host [n_rows]) writes the
 * table into a DPOSER_WS_SHARED_T workspace laid out for n_rows rows; dposer_prior_loss_tabled(row, table_rows = n_rows) then
 * evaluates the prior at t = t_host[row] (the caller passes the same t for the SDE scalars).  What the task loops use
 * (run/completion.py:183-201, run/motion_denoising.py:240-252 draw a new t every optimisation step). */
int dposer_prior_table_build(dposer_scorefc_t h, const float* flat_params, const void* packed, void* ws, const float* t_host,
                             int32_t n_rows, const float* freq, int64_t batch, void* stream);
/* ... with the SDE given: the rows are built for the labels the SDE's score function conditions the network on (t * 999; VE: sigma(t)).
 * sde == NULL is dposer_prior_table_build. */
int dposer_prior_table_build_sde(dposer_scorefc_t h, const float* flat_params, const void* packed, void* ws, const dposer_sde_desc* sde,
                                 const float* t_host, int32_t n_rows, const float* freq, int64_t batch, void* stream);
int dposer_prior_loss_tabled(dposer_scorefc_t h, const float* flat_params, const void* packed, void* ws, const dposer_sde_desc* sde,
                             const float* x0, const float* z, float t, int32_t row, int32_t table_rows, int32_t weighted, float inv_n,
                             float* x0_hat, float* grad, float* loss, uint64_t seed, uint32_t step, const float* sigmas,
                             int64_t batch, void* stream);

/* ---- TimeMLPs (lib/algorithms/advanced/model.py:69-90): the reference's secondary score model -------------------------------------
 *   net = Linear(in_dim, H), act, [Linear(H, H), act, Dropout(p)] x n_blocks, Linear(H, out_dim)      (in_dim = data_dim + 1: [x, t])
 * Flat parameter buffer: net.parameters() in order (weight [out][in] row-major, bias) per Linear, contiguous fp32.  Every Linear + act
 * is one MFMA GEMM launch of the score network's kernel family; the hidden width is padded to 128 channels inside the library. */
typedef struct dposer_mlp_s* dposer_mlp_t;
typedef struct {
    int32_t in_dim;          /* n_poses * pose_dim + 1 */
    int32_t out_dim;         /* n_poses * pose_dim */
    int32_t hidden_dim;      /* config.model.HIDDEN_DIM */
    int32_t n_blocks;        /* config.model.N_BLOCKS (0..8) */
    int32_t precision;       /* DPOSER_PREC_BF16 | DPOSER_PREC_FP32 */
    int32_t activation;      /* DPOSER_ACT_* (get_act, model.py:54-66) */
    float dropout_p;         /* config.model.dropout */
} dposer_mlp_desc;
int dposer_mlp_create(const dposer_mlp_desc* desc, dposer_mlp_t* out);
void dposer_mlp_destroy(dposer_mlp_t h);
int64_t dposer_mlp_num_params(dposer_mlp_t h);
int64_t dposer_mlp_packed_bytes(dposer_mlp_t h);
int64_t dposer_mlp_workspace_bytes(dposer_mlp_t h, int64_t batch);
int dposer_mlp_pack(dposer_mlp_t h, const float* flat_params, void* packed, void* stream);
/* out [B, out_dim] = net(x [B, in_dim]).  train_mode != 0: the Dropout modules draw from Philox(seed, step) (site = block index, same
 * counter layout as the score network's dropout); keep_for_backward != 0 keeps the pre-activations in `ws` for dposer_mlp_backward. */
int dposer_mlp_forward(dposer_mlp_t h, const float* flat_params, const void* packed, void* ws, const float* x, float* out, int64_t batch,
                       int32_t train_mode, int32_t keep_for_backward, uint64_t seed, uint32_t step, void* stream);
/* Backward of the forward that last ran on `ws` with keep_for_backward != 0 (same train_mode / seed / step): flat_grad [num_params]
 * (every element written; NULL: input gradient only) and dx [B, in_dim] (NULL: none). */
int dposer_mlp_backward(dposer_mlp_t h, const float* flat_params, const void* packed, void* ws, const float* dout, float* flat_grad,
                        float* dx, int64_t batch, int32_t train_mode, uint64_t seed, uint32_t step, void* stream);


/* get_sde_loss_fn.loss_fn + loss.backward() -- lib/algorithms/advanced/losses.py:80-137, 260
 * (continuous=True, reduce_mean=True, likelihood_weighting=False, model.train()).
 *   batch [B, D]; t [B] / z [B, D] injected draws or NULL (Philox: t = u*(T-eps)+eps, z ~ N(0,I));
 *   flat_grad [num_params] out: d loss / d params (no-grad ranges zeroed); loss [1] out. */
int dposer_dsm_loss_fwd_bwd(dposer_scorefc_t h, const float* flat_params, const void* packed, void* ws,
                            const dposer_sde_desc* sde, const float* batch_x, const float* t, const float* z,
                            float eps, uint64_t seed, uint32_t step, const float* freq, const float* sigmas,
                            float* flat_grad, float* loss, int64_t batch, void* stream);

/* Same step with the flat gradient delivered in BUCKETS so that the data-parallel all-reduce (RCCL, issued by the host
 * through torch.distributed) overlaps the rest of the backward pass -- the reference trains on one device
 * (run/train.py:150-170); this is the MI355X data-parallel extension BASELINE.json's north star asks for.
 *   dposer_scorefc_grad_buckets: number of buckets (n_layers + 1); [lo[b], hi[b]) are disjoint ranges of the flat buffer that
 *     cover every parameter with a gradient (the dead pre_dense_cond range, always zero in flat_grad, is in no bucket), listed in
 *     the order the backward pass completes them: last GN layer + post_dense first, ..., layer 0's weights, then layer 0's
 *     GroupNorm affine + the shared time embedding.
 *   bucket_events[b] (from dposer_event_create) is recorded on `stream` once bucket b of flat_grad is final; a
 *     communication stream waits on it with dposer_stream_wait_event before reducing that range. */
int32_t dposer_scorefc_grad_buckets(dposer_scorefc_t h, int64_t* lo, int64_t* hi, int32_t max_buckets);
int dposer_event_create(void** event);
void dposer_event_destroy(void* event);
int dposer_stream_wait_event(void* stream, void* event);
int dposer_dsm_loss_fwd_bwd_bucketed(dposer_scorefc_t h, const float* flat_params, const void* packed, void* ws,
                                     const dposer_sde_desc* sde, const float* batch, const float* t, const float* z,
                                     float eps, uint64_t seed, uint32_t step, const float* freq, const float* sigmas,
                                     float* flat_grad, float* loss, int64_t batch_size, void* const* bucket_events,
                                     int32_t n_events, void* stream);
/* The same step, telling the caller about final buckets WHILE the call is still queueing work: buckets that become final together
 * (a layer group of the backward pass: see dposer_scorefc_tuning_reload / DPOSER_WGRAD_GROUPS) are announced by ONE event --
 * bucket_events[first_bucket], recorded on `stream` -- followed at once by a call of `notify` from the calling thread with the
 * group's flat ranges (neighbouring buckets merged; n_ranges <= n_layers + 1).  The callback typically makes a communication stream
 * wait for `event` and enqueues the all-reduce of each range there: the host-side cost of issuing the collectives then overlaps
 * with GPU work that is already queued, and each collective with the rest of the backward pass.  `notify` must not synchronise
 * `stream`.  n_events >= dposer_scorefc_grad_buckets(). */
typedef void (*dposer_ranges_final_fn)(void* user, int32_t first_bucket, int32_t n_ranges, const int64_t* lo, const int64_t* hi, void* event);
int dposer_dsm_loss_fwd_bwd_notify(dposer_scorefc_t h, const float* flat_params, const void* packed, void* ws,
                                   const dposer_sde_desc* sde, const float* batch, const float* t, const float* z,
                                   float eps, uint64_t seed, uint32_t step, const float* freq, const float* sigmas,
                                   float* flat_grad, float* loss, int64_t batch_size, void* const* bucket_events,
                                   int32_t n_events, dposer_ranges_final_fn notify, void* user, void* stream);

/* Differentiable ScoreModelFC.forward for torch.autograd (model.py:141-196): forward keeps every layer
 * input / normalised activation in `ws` (DPOSER_WS_TRAIN layout); backward consumes the same `ws`.
 *   train_mode != 0: dropout active (model.train()), mask = Philox(seed, step), recomputed in backward;
 *   dout [B, D] upstream gradient; flat_grad [num_params] or NULL; dx [B, D] or NULL. */
int dposer_scorefc_forward_train(dposer_scorefc_t h, const float* flat_params, const void* packed, void* ws,
                                 const float* x, const float* labels, const float* freq, const float* sigmas,
                                 float* out, int64_t batch, int32_t train_mode, uint64_t seed, uint32_t step, void* stream);
int dposer_scorefc_backward(dposer_scorefc_t h, const float* flat_params, const void* packed, void* ws,
                            const float* labels, const float* sigmas, const float* dout, float* flat_grad, float* dx,
                            int64_t batch, int32_t train_mode, uint64_t seed, uint32_t step, void* stream);

/* optimize_fn + ema.update -- losses.py:44-58 (the lr warm-up value is computed by the caller),
 * torch.optim.Adam (losses.py:31-41), lib/algorithms/ema.py:32-51 -- ONE pass over flat fp32 buffers of n
 * elements (parameters, gradient, exp_avg, exp_avg_sq, EMA shadow).
 *   skip_lo/hi_host [n_skip <= 2]: HOST arrays of element ranges that have no gradient (Adam skipped,
 *   EMA still applied: pre_dense_cond);  grad_scale: multiplied into the gradient first (1/world_size
 *   after an all-reduce SUM);  grad_clip < 0 disables clip_grad_norm_;  adam_step = optimizer step count
 *   AFTER this update (>= 1);  ema may be NULL;  ema_one_minus_decay = 1 - min(decay, (1+n)/(10+n));
 *   scratch: >= 8 KiB floats, caller-owned and persistent across steps: scratch[0] = squared gradient norm of this step,
 *   scratch[1] = number of steps DROPPED so far because that norm was not finite (NaN / Inf gradient: parameters, moments and
 *   EMA are left untouched on the device, no host round trip; zero scratch[1] once, before the first step). */
int dposer_adam_ema_clip_step(float* flat_params, const float* flat_grad, float* exp_avg, float* exp_avg_sq, float* ema,
                              int64_t n, const int64_t* skip_lo_host, const int64_t* skip_hi_host, int32_t n_skip,
                              double lr, double beta1, double beta2, double eps, double grad_clip, double grad_scale,
                              int64_t adam_step, double ema_one_minus_decay, float* scratch, void* stream);

/* Sharded optimiser step (ZeRO-1 style data parallelism): every rank owns a contiguous range of the flat buffers.
 *   dposer_grad_sqnorm: scratch[0] = sum of squares of grad[0..n) (this rank's reduce-scattered gradient range); the caller
 *   all-reduces that one float over the ranks to get the global squared norm for clip_grad_norm_ (losses.py:54-55);
 *   dposer_adam_ema_clip_step_presummed: dposer_adam_ema_clip_step on the range, taking the squared norm from scratch[0] instead
 *   of recomputing it (pointers and skip ranges are relative to the range). */
int dposer_grad_sqnorm(const float* grad, int64_t n, float* scratch, void* stream);

/* Runge-Kutta stage combination on the float64 state of the probability-flow ODE (lib/algorithms/advanced/likelihood.py:86-99,
 * sampling.py:513-530 hand the state to scipy.integrate.solve_ivp, whose RK45 forms these sums on the host):
 *   out[i] = (y ? y[i] : 0) + scale * (coef[0] k[0][i] + coef[1] k[1][i] + ...), left to right in float64, no fused multiply-add.
 *   k_host / coef_host: HOST arrays of n_terms (<= 8) DEVICE pointers / coefficients; y may be NULL; out may alias nothing. */
int dposer_rk_combine_f64(double* out, const double* y, const double* const* k_host, const double* coef_host, int32_t n_terms,
                          double scale, int64_t n, void* stream);
/* Right-hand side of the probability-flow ODE around one evaluation of the score network (lib/algorithms/advanced/likelihood.py:60-65,
 * 86-95 `ode_func`; sampling.py:513-530 `ode_func`): drift = -1/2 beta(t) x - 1/2 g(t)^2 score, score = -model(x, 999 t) / std(t)
 * (sde_lib.py:100-104, utils.py:152-162), all samples at the same t, VP / sub-VP; under the VE SDE (sde_lib.py:208-253) drift =
 * -1/2 g(t)^2 score with g = sigma(t) sqrt(2 ln(sigma_max / sigma_min)), labels = sigma(t) and score = model(x, sigma(t)) (utils.py:164-175),
 * the beta terms below zero.  Two elementwise launches around the network calls:
 *   begin: x [B, D] = float(state[0 .. B*D)), labels [B] = 999 t, and -- noise != NULL, dout != NULL -- dout [B, D] = the gradient of
 *          sum(drift * noise) w.r.t. the network output (what torch.autograd hands to the network's backward in likelihood.py:29-35);
 *   end:   dstate[0 .. B*D) = double(drift(x, model_out)); with dx (the network's input gradient for `dout`) also the Hutchinson
 *          estimate dstate[B*D + b] = sum_i noise[b,i] * (dx[b,i] - 1/2 beta(t) noise[b,i]);  dx NULL: drift only (dstate [B*D]). */
int dposer_pf_ode_rhs_begin(const dposer_sde_desc* sde, float t, const double* state, const float* noise, float* x, float* labels,
                            float* dout, int64_t batch, int32_t dim, void* stream);
int dposer_pf_ode_rhs_end(const dposer_sde_desc* sde, float t, const float* x, const float* model_out, const float* dx,
                          const float* noise, double* dstate, int64_t batch, int32_t dim, void* stream);
int dposer_adam_ema_clip_step_presummed(float* flat_params, const float* flat_grad, float* exp_avg, float* exp_avg_sq, float* ema,
                                        int64_t n, const int64_t* skip_lo_host, const int64_t* skip_hi_host, int32_t n_skip,
                                        double lr, double beta1, double beta2, double eps, double grad_clip, double grad_scale,
                                        int64_t adam_step, double ema_one_minus_decay, float* scratch, void* stream);
/* the same update with torch.optim.Adam's weight_decay (losses.py:35-36 get_optimizer passes config.optim.weight_decay): the
 * clipped gradient becomes grad + weight_decay * param before the moments; presummed != 0: scratch[0] already holds the squared norm */
int dposer_adam_ema_clip_step_wd(float* flat_params, const float* flat_grad, float* exp_avg, float* exp_avg_sq, float* ema_shadow,
                                 int64_t n, const int64_t* skip_lo_host, const int64_t* skip_hi_host, int32_t n_skip, double lr,
                                 double beta1, double beta2, double eps, double weight_decay, double grad_clip, double grad_scale,
                                 int64_t adam_step, double ema_one_minus_decay, float* scratch, int32_t presummed, void* stream);
/* The same update fused with the re-packing of the weights it changes: ONE pass over the optimizer state writes parameters, moments,
 * EMA shadow AND the packed copies (MFMA fragment order, transposed dgrad copies, fp32 time-branch copies, the bias table) that the
 * next forward / backward read -- dposer_scorefc_pack at the start of the next step then has nothing to do.  `packed` must have been
 * filled by dposer_scorefc_pack(with_backward = 1) before: the zero padding of the copies is written there and never again.  All flat
 * buffers hold dposer_scorefc_num_params() floats.  Results are bitwise those of dposer_adam_ema_clip_step_wd followed by
 * dposer_scorefc_pack.  The caller owns staleness: whoever changes the parameters by other means must pack again. */
int dposer_scorefc_adam_pack_step(dposer_scorefc_t h, float* flat_params, const float* flat_grad, float* exp_avg, float* exp_avg_sq,
                                  float* ema_shadow, void* packed, const int64_t* skip_lo_host, const int64_t* skip_hi_host,
                                  int32_t n_skip, double lr, double beta1, double beta2, double eps, double weight_decay,
                                  double grad_clip, double grad_scale, int64_t adam_step, double ema_one_minus_decay, float* scratch,
                                  int32_t presummed, void* stream);

/* dposer_em_sampler restricted to the steps [start_step, start_step + n_steps) (no look-ahead imputation after the last one):
 * what a predictor-corrector loop with a corrector between the predictor calls drives (sampling.py:455-461). */
int dposer_em_sampler_steps(dposer_scorefc_t h, const float* flat_params, const void* packed, void* ws, const dposer_sde_desc* sde,
                            float* x, float* x_mean, const float* timesteps_host, int32_t start_step, int32_t n_steps,
                            const float* observation, const float* mask, const float* noise, uint64_t seed, float* traj,
                            int32_t traj_stride, const float* freq, const float* sigmas, int64_t batch, void* stream);

/* LangevinCorrector.update_fn -- sampling.py:282-302, one corrector step at a shared t (alpha = sde.alphas[timestep] for VP,
 * 1 for sub-VP, :290-294), in two phases around the batch means of :296-297:
 *   phase 0: evaluates the network at x and writes norm_sums[0] = sum_b ||grad_b||, norm_sums[1] = sum_b ||noise_b|| over this
 *            call's `batch` samples (DEVICE float[2]); under data parallelism the caller all-reduces (SUM) the two floats;
 *   phase 1: x_mean = x + step * grad, x = x_mean + sqrt(2 step) * noise, step = (snr * mean||noise|| / mean||grad||)^2 * 2 * alpha
 *            with mean = norm_sums * inv_global_batch.  Same ws for both phases, nothing else on it in between.
 * noise [B, D] injected or NULL -> Philox(seed, step) (the two phases regenerate the same numbers). */
int dposer_langevin_step(dposer_scorefc_t h, const float* flat_params, const void* packed, void* ws, const dposer_sde_desc* sde,
                         float* x, float* x_mean, float t, float alpha, float snr, const float* noise, uint64_t seed, uint32_t step,
                         float* norm_sums, int32_t phase, double inv_global_batch, const float* freq, const float* sigmas,
                         int64_t batch, void* stream);

/* DPoserComp.optimize -- run/completion.py:167-207 (loss :131-149, weights :151-155): `n_steps` Adam steps on the pose batch
 * x [B, D] (in: the initial value, i.e. the observation; out: the optimised variable -- the caller applies the final
 * observation/mask blend of :205) under  w_prior[i] * mean(w (x - x0_hat)^2) + w_data[i] * MSE(x * mask, obs * mask).
 *   adam_m / adam_v [B, D]: torch.optim.Adam moments, zero on entry for a fresh optimiser (per-sample state: the loop shards
 *   over GPUs with no collective);  t_host / weighted_host / w_prior_host / w_data_host [n_steps]: HOST arrays of the
 *   per-step time, the `weighted` flag (the reference passes quan_t there, :196) and the loss weights;
 *   noise [n_steps, B, D] injected z of :133, or NULL -> Philox(seed, step0 + i).  Workspace: DPOSER_WS_SHARED_T with
 *   n_steps table rows. */
int dposer_completion_optimize(dposer_scorefc_t h, const float* flat_params, const void* packed, void* ws, const dposer_sde_desc* sde,
                               float* x, const float* observation, const float* mask, float* adam_m, float* adam_v,
                               const float* t_host, const int32_t* weighted_host, const float* w_prior_host, const float* w_data_host,
                               int32_t n_steps, double lr, double beta1, double beta2, double eps, const float* noise, uint64_t seed,
                               uint32_t step0, const float* freq, const float* sigmas, int64_t batch, void* stream);

/* Per-launch GEMM timing (HIP events recorded on the launch stream around every MFMA GEMM launch; off by
 * default).  collect() synchronises the recorded events and returns, per kernel kind, the summed
 * duration [ms], launch count and algorithmic FLOPs (2*M*N*K of the un-padded problem); arrays of
 * dposer_profile_num_kinds() entries in HOST memory.  Used by bench.py for the live roofline.
 * enable(on): 0 = off (and forget the records), -1 = pause (keep them), 1 = every GEMM launch, 2 + k = launches of epilogue kind k only (k: 0 gn_fwd, 1 gn_fwd_train,
 * 2 bias_silu, 3 rowmajor, 4 plain_ft, 5 gn_bwd_dgrad, 6 silu_bwd_dgrad, 7 wgrad, 8 post_em_step) -- two event records per
 * launch cost about 3 % of a training step when every launch is bracketed, so a timed region brackets one kind. */
void dposer_profile_enable(int32_t on);
int32_t dposer_profile_num_kinds(void);
int dposer_profile_collect(double* ms_host, int64_t* launches_host, double* flops_host);
void dposer_profile_kind_name(int32_t kind, char* out, int32_t n);

/* ------------------------------------------------------------------------------------------
 * Body model  (lib/body_model/body_model.py:68-112 -> smplx.lbs, lib/utils/transforms.py:227-235)
 * ---------------------------------------------------------------------------------------- */
/* rot6d_to_mat3x3 -- lib/utils/transforms.py:227-235: rot6d [n, 6] -> rotmat [n, 3, 3] */
int dposer_rot6d_to_rotmat(const float* rot6d, float* rotmat, int64_t n, void* stream);
/* batch_rodrigues -- smplx/lbs.py: axis-angle [n, 3] -> rotmat [n, 3, 3] */
int dposer_rodrigues(const float* axis_angle, float* rotmat, int64_t n, void* stream);

/* rot6d_to_axis_angle -- lib/utils/transforms.py:197-224 (Gram-Schmidt, then torchgeometry.rotation_matrix_to_angle_axis:
 * matrix -> unit quaternion -> angle-axis with the angle in [0, pi]; NaNs zeroed as :223); also from a rotation matrix [n, 3, 3] */
int dposer_rot6d_to_axis_angle(const float* rot6d, float* axis_angle, int64_t n, void* stream);
int dposer_rotmat_to_axis_angle(const float* rotmat, float* axis_angle, int64_t n, void* stream);

typedef struct dposer_body_s* dposer_body_t;
typedef struct {
    int32_t num_joints;      /* 24 SMPL / 52 SMPL-H / 55 SMPL-X */
    int32_t num_vertices;
    int32_t num_shape;       /* betas + expression coefficients */
    int32_t num_extra;       /* vertex-selected extra joints */
    int32_t num_landmarks;   /* barycentric landmarks */
} dposer_body_desc;
int dposer_body_create(const dposer_body_desc* desc, const int32_t* parents_host, dposer_body_t* out);
void dposer_body_destroy(dposer_body_t h);

/* Rest shape: shape blend shapes + joint regression -- smplx/lbs.py blend_shapes + vertices2joints at the head of lbs(), as
 * reached from lib/body_model/body_model.py:75-88 with betas / expression given per pose (run/smplify.py:200-260 optimises
 * betas; the task loops pass constant ones).
 *   v_shaped[b] = v_template + shapedirs . shape[b]          shapedirs [V*3, L] (the asset's [V,3,L]), shape [B, L]
 *   j_rest[b]   = J_regressor @ v_shaped[b] = j_template + jdirs . shape[b]
 * The regressor is LINEAR, so it is applied once to the template and to every blend-shape direction on the host side
 * (j_template [J*3] = J_regressor @ v_template, jdirs [J*3, L] = J_regressor @ shapedirs[..., l]); per pose the joints are an
 * L-term combination instead of a [J, V] x [V, 3] product.
 * Backward: d_shape [B, L] = d_v_shaped . shapedirs + d_j_rest . jdirs; scratch >= dposer_shape_blend_scratch_floats(...) floats. */
int dposer_shape_blend_forward(const float* v_template, const float* shapedirs, const float* j_template, const float* jdirs,
                               const float* shape, float* v_shaped, float* j_rest, int32_t num_vertices, int32_t num_joints,
                               int32_t num_shape, int64_t batch, void* stream);
int64_t dposer_shape_blend_scratch_floats(int32_t num_vertices, int32_t num_shape, int64_t batch);
int dposer_shape_blend_backward(const float* shapedirs, const float* jdirs, const float* d_v_shaped, const float* d_j_rest,
                                float* d_shape, float* scratch, int32_t num_vertices, int32_t num_joints, int32_t num_shape,
                                int64_t batch, void* stream);

/* Forward kinematics: Rodrigues + kinematic chain (smplx/lbs.py batch_rodrigues, batch_rigid_transform)
 * with the pose assembled as SMPLX.forward does (reference call site lib/body_model/body_model.py:75-88):
 *   pose_segments_host[i]: DEVICE pointer to segment i, [B, segment_joints[i]*3] axis-angle, or NULL
 *     (= zeros, what the reference gets from the module's default parameters); the host arrays
 *     themselves are HOST memory.  SMPL-X order: global(1) body(21) jaw(1) leye(1) reye(1) lhand(15) rhand(15).
 *   j_rest [J,3] (shared) or [B,J,3]: rest-pose joints (J_regressor @ v_shaped);
 *   joints [B, n_out, 3] out: posed joints [0, n_out) (+ transl [B,3] if not NULL);
 *   rel_transforms [B, n_out, 12] out or NULL: rows of the 3x4 skinning transforms A_i. */
int dposer_fk_joints(dposer_body_t h, const float* const* pose_segments_host, const int32_t* segment_joints_host,
                     int32_t num_segments, const float* j_rest, int32_t j_rest_batched, const float* transl,
                     float* joints, float* rel_transforms, int32_t n_out, int64_t batch, void* stream);

/* Linear blend skinning -- smplx/lbs.py lbs() + SMPLX.forward joint assembly, as called from
 * lib/body_model/body_model.py:75-103 (BodyModel.forward) and lib/body_model/smpl.py:67-77.
 *   posedirs_packed: dposer_lbs_pack_posedirs(posedirs [(J-1)*9, V*3]) (MFMA fragment order, fp32);
 *   v_shaped [V,3] (betas folded in, shared) or [B,V,3]; j_rest as in dposer_fk_joints;
 *   skin_idx / skin_w [V, skin_k]: ELL form of lbs_weights [V,J] (zero weights dropped or padded);
 *   extra_vertex_ids [num_extra], lmk_tri [num_landmarks,3] (= faces[lmk_faces_idx]), lmk_bary [num_landmarks,3] -- or all three NULL: the rows behind
 *   the kinematic tree's J joints are then left unwritten (callers that read tree joints only: the motion-denoising loop);
 *   verts [B,V,3] out; joints [B, J+num_extra+num_landmarks, 3] out (SMPL-X: 55+21+51 = 127). */
int64_t dposer_lbs_posedirs_packed_bytes(dposer_body_t h);
int dposer_lbs_pack_posedirs(dposer_body_t h, const float* posedirs, void* packed, void* stream);
int64_t dposer_lbs_workspace_bytes(dposer_body_t h, int64_t batch);
int dposer_lbs_forward(dposer_body_t h, void* ws, const void* posedirs_packed, const float* const* pose_segments_host,
                       const int32_t* segment_joints_host, int32_t num_segments, const float* j_rest, int32_t j_rest_batched,
                       const float* v_shaped, int32_t v_shaped_batched, const int32_t* skin_idx, const float* skin_w,
                       int32_t skin_k, const float* transl, const int32_t* extra_vertex_ids, const int32_t* lmk_tri,
                       const float* lmk_bary, float* verts, float* joints, int64_t batch, void* stream);

/* dposer_lbs_forward for callers that need the vertices ONLY for the temporal smoothness term of run/motion_denoising.py:253-255
 * (`torch.mean(torch.norm(verts[:-1] - verts[1:], dim=-1))` and its autograd): skinning and that term's gradient in one pass, the
 * vertices never reach HBM.  batch = whole sequences of frames_per_sequence consecutive frames (neighbours never cross a sequence);
 *   d_verts [B,V,3] out = scale * (u_t - u_{t-1}), u_t = (v[t] - v[t+1]) / ||v[t] - v[t+1]|| (0/0 = NaN like torch's sqrt backward);
 *   dist_part [B, ceil(V/256)] out: sums of ||v[t] - v[t+1]|| over blocks of 256 vertices (0 for the last frame of a sequence) --
 *   their sum / ((F-1) V) is the term's value;  joints [B, J+num_extra+num_landmarks, 3]: only [:, :J] is written (the extra
 *   vertices / landmarks would need the vertices).  skin_k must be 4.  Other arguments as dposer_lbs_forward. */
int dposer_lbs_forward_temporal_grad(dposer_body_t h, void* ws, const void* posedirs_packed, const float* const* pose_segments_host,
                                     const int32_t* segment_joints_host, int32_t num_segments, const float* j_rest, int32_t j_rest_batched,
                                     const float* v_shaped, int32_t v_shaped_batched, const int32_t* skin_idx, const float* skin_w,
                                     int32_t skin_k, const float* transl, int64_t frames_per_sequence, float scale, float* d_verts,
                                     float* dist_part, float* joints, int64_t batch, void* stream);

/* The same step with NO vertex-sized array in HBM at all (round 6; run/motion_denoising.py:217-218,253-267):
 *   dposer_lbs_forward_front          FK + pose-blend offsets of dposer_lbs_forward without its skinning kernel: joints[:, :J] and the
 *                                     workspace `ws` (pose feature, skinning transforms, offsets);
 *   dposer_lbs_backward_temporal      dposer_lbs_backward whose vertex gradient is the temporal term's -- scale * (u_t - u_{t-1}) -- formed
 *                                     inside the skinning-backward kernel from `ws_fwd` (the vertices of a frame and of its two neighbours
 *                                     are skinned in registers); d_joints carries the data term.  dist_part4 [B, ceil(V/256), 4] out: per-wave
 *                                     sums of ||v[t] - v[t+1]||, ((p0 + p1) + p2) + p3 of an entry = dist_part of dposer_lbs_forward_temporal_grad.
 *                                     The vertices and the vertex gradient carry the bits of dposer_lbs_forward + the two-kernel gradient; the backward half agrees with
 *                                     dposer_lbs_backward to fp32 rounding (FMA-contracted transform blend: 2e-6 on the optimised poses).
 *   dposer_lbs_temporal_in_backward_ok(h, skin_k, batch): 1 when dposer_lbs_backward_temporal can run (prepared joint lists with the
 *                                     matrix-pipe tables, four influences per vertex, batch >= DPOSER_LBS_JOINT_STREAM_MIN, bf16 x 3 blend);
 *                                     dposer_motion_denoise_optimize asks it and falls back to the forms above. */
int32_t dposer_lbs_temporal_in_backward_ok(dposer_body_t h, int32_t skin_k, int64_t batch);
int dposer_lbs_forward_front(dposer_body_t h, void* ws, const void* posedirs_packed, const float* const* pose_segments_host,
                             const int32_t* segment_joints_host, int32_t num_segments, const float* j_rest, int32_t j_rest_batched,
                             const float* transl, float* joints, int64_t batch, void* stream);
int dposer_lbs_backward_temporal(dposer_body_t h, const void* ws_fwd, void* ws_bwd, const void* posedirs_bwd_packed,
                                 const float* const* pose_segments_host, const int32_t* segment_joints_host, int32_t num_segments,
                                 const float* j_rest, int32_t j_rest_batched, const float* v_shaped, int32_t v_shaped_batched,
                                 const int32_t* skin_idx, const float* skin_w, int32_t skin_k, const int32_t* joint_ptr,
                                 const int32_t* joint_vidx, const float* joint_w, int64_t frames_per_sequence, float scale,
                                 float* dist_part4, const float* d_joints, int64_t d_joints_ld, float* const* d_pose_segments_host,
                                 int64_t batch, void* stream);

/* Backward of dposer_lbs_forward (autograd of BodyModel.forward w.r.t. pose / rest joints / v_posed; the reference
 * differentiates through smplx in run/motion_denoising.py:217-218,255-267 and run/smplify.py:200-260).
 *   ws_fwd: the workspace of the matching forward call (unmodified since);  posedirs_bwd_packed:
 *   dposer_lbs_pack_posedirs_bwd(posedirs);  joint_ptr [J+1] / joint_vidx / joint_w: CSR-by-joint form of lbs_weights;
 *   d_verts [B,V,3]; d_joints [B, d_joints_ld] (first J*3 entries of a row = gradient of the posed LBS joints; extras and
 *   landmarks are vertex gathers and must be folded into d_verts by the caller);
 *   d_pose_segments_host[i]: DEVICE pointer [B, seg_joints*3] out or NULL;  d_jrest [B,J,3] out or NULL;
 *   d_vposed [B,V,3] out or NULL (= gradient w.r.t. v_shaped).
 *   dposer_lbs_prepare_joint_lists(h, joint_ptr, joint_vidx, joint_w, stream): SETUP call, once per set of lists (and again
 *   whenever their contents change): re-cuts the lists by vertex chunk for the one-pass skinning-backward kernels and stores the
 *   tables in the handle -- among them the skinning weights of every 256-vertex chunk as a dense 64 x 256 matrix in bf16 hi / lo planes,
 *   laid out as MFMA operand fragments (64 KB per chunk: the joint-transform reduction runs on the matrix pipe).  It synchronises the
 *   stream, copies the lists to the host and allocates -- dposer_lbs_backward itself does none of that; without prepared lists it runs
 *   the (pose, joint)-parallel gather kernel, which is the right kernel below ~320 poses anyway.  dposer_lbs_backward forks onto two
 *   internal streams of the handle for the three product terms of its blend-gradient GEMM and joins them before it returns control
 *   of `stream` (events only, no host synchronisation): one call at a time per handle.
 *   dposer_body_tuning_reload(): re-reads the A/B environment switches of the body-model kernels (DPOSER_FK_SMALL_MAX,
 *   DPOSER_LBS_JOINT_STREAM_MIN, DPOSER_LBS_BLEND); they are otherwise read once per process, not per call. */
int dposer_lbs_prepare_joint_lists(dposer_body_t h, const int32_t* joint_ptr, const int32_t* joint_vidx, const float* joint_w, void* stream);
void dposer_body_tuning_reload(void);
int64_t dposer_lbs_posedirs_bwd_packed_bytes(dposer_body_t h);
int dposer_lbs_pack_posedirs_bwd(dposer_body_t h, const float* posedirs, void* packed, void* stream);
/* dposer_lbs_backward_fold: dposer_lbs_backward + the backward of the joint rows that are functions of VERTICES -- smplx
 *   VertexJointSelector (row J + e = vertex extra_vertex_ids[e]) and vertices2landmarks (row J + n_extra + l = barycentric combination of
 *   the three vertices of landmark l; called from lib/body_model/body_model.py:75-88 through smplx.SMPLX.forward): their gradients
 *   d_joints[:, J:] are added to the gradient of the vertices they read.  `fold` (device tables, built once per asset by the caller):
 *     vertex_slot [V]   slot u of the vertex, or -1;      slot_vertex [n_slots]   the vertex of slot u;
 *     slot_ptr [n_slots + 1], entry_row [n], entry_weight [n]: slot u receives sum_e entry_weight[e] * d_joints[:, entry_row[e]] over
 *     e in [slot_ptr[u], slot_ptr[u + 1]), added in that order (deterministic).
 *   d_verts is only READ (the corrected rows live in ws_bwd); fold == NULL is dposer_lbs_backward. */
typedef struct dposer_lbs_joint_fold {
    const int32_t* vertex_slot;
    const int32_t* slot_vertex;
    const int32_t* slot_ptr;
    const int32_t* entry_row;
    const float* entry_weight;
    int32_t n_slots;
} dposer_lbs_joint_fold;
int dposer_lbs_backward_fold(dposer_body_t h, const void* ws_fwd, void* ws_bwd, const void* posedirs_bwd_packed,
                             const float* const* pose_segments_host, const int32_t* segment_joints_host, int32_t num_segments,
                             const float* j_rest, int32_t j_rest_batched, const float* v_shaped, int32_t v_shaped_batched,
                             const int32_t* skin_idx, const float* skin_w, int32_t skin_k, const int32_t* joint_ptr,
                             const int32_t* joint_vidx, const float* joint_w, const float* d_verts, const float* d_joints,
                             int64_t d_joints_ld, const dposer_lbs_joint_fold* fold, float* const* d_pose_segments_host, float* d_jrest,
                             float* d_vposed, int64_t batch, void* stream);
int64_t dposer_lbs_backward_workspace_bytes(dposer_body_t h, int64_t batch);
int dposer_lbs_backward(dposer_body_t h, const void* ws_fwd, void* ws_bwd, const void* posedirs_bwd_packed,
                        const float* const* pose_segments_host, const int32_t* segment_joints_host, int32_t num_segments,
                        const float* j_rest, int32_t j_rest_batched, const float* v_shaped, int32_t v_shaped_batched,
                        const int32_t* skin_idx, const float* skin_w, int32_t skin_k, const int32_t* joint_ptr,
                        const int32_t* joint_vidx, const float* joint_w, const float* d_verts, const float* d_joints,
                        int64_t d_joints_ld, float* const* d_pose_segments_host, float* d_jrest, float* d_vposed,
                        int64_t batch, void* stream);

/* MotionDenoise.optimize -- run/motion_denoising.py:199-300 (DPoser_loss :124-143, weights :157-163, temporal / data terms
 * :253-263): `n_steps` torch.optim.Adam steps on the axis-angle body pose [frames, body joints * 3] of ONE sequence under
 *   w_temp[i] * mean ||v[t] - v[t+1]|| + w_data[i] * mean ||Jtr[:, :n_obs_joints] - joints_obs|| + w_prior[i] * DPoser_loss(normalise(pose)),
 * all steps queued from one call (prior evaluation = dposer_prior_loss, body model = dposer_lbs_forward / _backward, the
 * loss gradients and the Adam update are kernels of this entry).  The data term is dropped for a step when its value is not
 * finite and > 0 (:261-263) -- decided on the device.
 *   score network: handle, parameters, packed weights, a DPOSER_WS_SHARED_T workspace for `frames` samples and n_steps table rows;
 *   body model: handle + the device tables dposer_lbs_forward / dposer_lbs_backward take, their two workspaces for `frames`
 *   poses; rest_batched: v_shaped / j_rest are [frames, ...] instead of shared; pose segments other than `body_segment` are
 *   the zero pose;  joint_rows = J + num_extra + num_landmarks (row count of the LBS joint output);
 *   norm_mode 0 none / 1 z-score (norm_a = mean, norm_b = std) / 2 min-max (norm_a = min, norm_b = max), arrays [pose dim];
 *   t_host / w_temp_host / w_data_host / w_prior_host [n_steps]: HOST arrays; weighted: the `weighted` flag of DPoser_loss
 *   (False in the reference, :124); adam_m / adam_v zero on entry for a fresh optimiser, adam_step0 = steps already taken;
 *   noise [n_steps, frames, pose dim] injected z of the prior or NULL -> Philox(seed, step0 + i);
 *   scratch: dposer_motion_denoise_scratch_bytes(frames, pose dim, V, joint_rows) bytes, 256-byte aligned;
 *   frames_per_sequence: 0 = one sequence of `frames` frames; F = a batch of frames / F sequences of F consecutive frames each
 *   (independent problems advanced together: temporal neighbours, the data-term decision and the loss means are per sequence; the
 *   in-kernel prior noise is keyed by the frame index inside the batch);
 *   loss_log: DEVICE [n_steps, sequences, 3] (temp, data values of each sequence; the prior value is the batch total, i.e. the sum
 *   over the sequences' prior terms) of every step, unweighted, or NULL. */
typedef struct dposer_motion_denoise_args {
    dposer_scorefc_t net;
    const float* flat_params;
    const void* packed;
    void* net_ws;
    const dposer_sde_desc* sde;
    const float* freq;
    const float* sigmas;
    dposer_body_t body;
    void* lbs_ws_fwd;
    void* lbs_ws_bwd;
    const void* posedirs_packed;
    const void* posedirs_bwd_packed;
    const float* j_rest;
    const float* v_shaped;
    int32_t rest_batched;
    const int32_t* skin_idx;
    const float* skin_w;
    int32_t skin_k;
    const int32_t* joint_ptr;
    const int32_t* joint_vidx;
    const float* joint_w;
    const int32_t* extra_vertex_ids;
    const int32_t* lmk_tri;
    const float* lmk_bary;
    const int32_t* segment_joints_host;
    int32_t num_segments;
    int32_t body_segment;
    int32_t num_vertices;
    int32_t num_joints;
    int32_t joint_rows;
    int64_t frames;       /* sequences x frames_per_sequence, at most 65535 per call (sequences are independent: a larger batch goes through in groups
                             of whole sequences, as MotionDenoise.optimize_sequences does) */
    int64_t frames_per_sequence;
    float* pose;
    float* adam_m;
    float* adam_v;
    const float* joints_obs;
    int32_t n_obs_joints;
    int32_t norm_mode;
    const float* norm_a;
    const float* norm_b;
    int32_t n_steps;
    int32_t weighted;
    const float* t_host;
    const float* w_temp_host;
    const float* w_data_host;
    const float* w_prior_host;
    double lr, beta1, beta2, eps;
    int32_t adam_step0;
    uint32_t step0;
    uint64_t seed;
    const float* noise;
    void* scratch;
    float* loss_log;
    int32_t rot6d;        /* != 0: the score network works on the 6-D rotation representation (rot_rep = 'rot6d': 6 J inputs, norm_a / norm_b
                             hold 6 J statistics, noise is [n_steps, frames, 6 J]); the pose being optimised stays axis-angle */
} dposer_motion_denoise_args;
int64_t dposer_motion_denoise_scratch_bytes(int64_t frames, int32_t pose_dim, int32_t num_vertices, int32_t joint_rows);
int dposer_motion_denoise_optimize(const dposer_motion_denoise_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DPOSER_HIP_H */
