// Non-template entry points of the MFMA GEMM family (instantiated in gemm_launch.hip).
#pragma once
#include "epilogues.h"
#include "gemm.h"
#include "gemm_wgrad_tr.h"

// Workgroup tilings (channels x samples); K pipeline = ring of 4 slots x 2 FT k-blocks (bf16: 128 x 32 and 64 x 32 three slots x 4, 128 x 64 / 8 waves four x 4;
// fp32 128 x 32: 2 slots x 4):
enum GemmShape : int {
    SHAPE_BIG = 0,      // 256 x 256, 8 waves (2x4), wave tile 128 x 64    -- large batches
    SHAPE_MID = 1,      // 128 x 128, 4 waves (2x2), wave tile  64 x 64
    SHAPE_SMALL = 2,    // 128 x  32, 4 waves (4x1), wave tile  32 x 32    -- small batches
    SHAPE_FINAL = 3,    //  64 x 128, 4 waves (1x4), wave tile  64 x 32    -- post_dense (N = 63 -> 64)
    SHAPE_FINAL_S = 4,  //  64 x  32, 2 waves (2x1), wave tile  32 x 32
    SHAPE_WIDE64 = 5,   // 128 x  64, 4 waves (2x2), wave tile  64 x 32    -- wgrad of pre_dense (K = 63 -> 64)
    SHAPE_SMALL64 = 6,  // 128 x  64, 8 waves (4x2), wave tile  32 x 32    -- GroupNorm layers at 1024 < samples <= 2048 (bf16; round 6)
    GEMM_NSHAPES = 7
};
static inline int shape_ct(int s) { static const int v[] = {8, 4, 4, 2, 2, 4, 4}; return v[s]; }   // 32-channel tiles / block
static inline int shape_st(int s) { static const int v[] = {8, 4, 1, 4, 1, 2, 2}; return v[s]; }   // 32-sample tiles / block
static inline int shape_ws(int s) { static const int v[] = {4, 2, 1, 4, 1, 2, 2}; return v[s]; }   // waves along samples

// PREC_BF16X3: the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16) under the fp32-storage epilogues -- the caller hands the operands as bf16
// hi / lo planes (three K segments per term against weights packed [hi | lo | hi]); instantiated in gemm_launch_x3.hip
enum : int { PREC_BF16 = 0, PREC_FP32 = 1, PREC_BF16X3 = 2 };

// gs = channels per GroupNorm group (hidden_dim / 32): 32 = the shipped tile-per-group epilogues; 16 / 64 = the generic ones (128x128 and 128x32 tilings)
hipError_t gemm_gn(int prec, bool train, int shape, const GemmArgs& g, const GNParams& p, hipStream_t st, int gs = 32);
hipError_t gemm_bias_silu(int prec, bool train, int shape, const GemmArgs& g, const BiasSiLUParams& p, hipStream_t st);
hipError_t gemm_rowmajor(int prec, int shape, const GemmArgs& g, const RowMajorParams& p, hipStream_t st);
hipError_t gemm_em_step(int prec, int shape, const GemmArgs& g, const EmStepParams& p, hipStream_t st);
hipError_t gemm_partial_ft(int prec, int shape, const GemmArgs& g, const PartialFTParams& p, hipStream_t st);   // SHAPE_MID, g.ksplit splits of segment 0
hipError_t gemm_dsm_step(int prec, int shape, const GemmArgs& g, const DsmStepParams& p, hipStream_t st);   // SHAPE_FINAL / SHAPE_FINAL_S
hipError_t gemm_plain_ft(int prec, int shape, const GemmArgs& g, const PlainFTParams& p, hipStream_t st);
hipError_t gemm_gn_bwd(int prec, int shape, const GemmArgs& g, const GNBwdParams& p, hipStream_t st, int gs = 32);
hipError_t gemm_silu_bwd(int prec, int shape, const GemmArgs& g, const SiLUBwdParams& p, hipStream_t st);
hipError_t gemm_wgrad(int prec, int shape, const GemmArgs& g, const WgradParams& p, hipStream_t st);
// bf16, any wgrad tiling (256x256, 128x128, 64x128, 128x64), operands sample-major (gemm_wgrad_tr.h): no transposed activation copies needed
hipError_t gemm_wgrad_tr(int shape, const WgradTrArgs& g, const WgradParams& p, hipStream_t st);
// every 256x256 wgrad tile of a training step in one launch (wgrad_batch.h)
hipError_t gemm_wgrad_tr_batch(const WgradBatchArgs& a, hipStream_t st);

// ---- persistent Euler-Maruyama sampler (gemm_sampler.hip): one workgroup per block of 256 samples walks every layer of every step
struct SamplerLayer {
    const void* W;            // packed layer weights (x-path prefix of the K-concatenated rows)
    const void* in;           // FT input of the layer (this step's state for layer 0)
    const void* resid;        // FT residual input or null
    void* out;                // FT output
    const float* gamma;
    const float* beta;
    int w_stride_blocks;      // k-blocks per packed weight row-block
    int kblocks;              // k-blocks of the x-path
};
struct SamplerArgs {
    const SamplerLayer* layers;   // DEVICE table [L]
    int L, H;
    int64_t Spad;
    const float* table;           // [n_steps][L][H] time-bias rows
    const float* tsteps;          // DEVICE [n_steps] t of every step
    int n_steps;
    uint32_t step0;               // global index of the first step (Philox offset)
    const void* Wpost;
    int post_kblocks;
    const void* last;             // FT output of the last GroupNorm layer
    float* x_mean_ft;             // written on the last step
    EmStepParams em;              // per-step fields (t, step, x_mean_ft) are filled in by the kernel
    // cluster form only (k_sampler_cluster): all zero before the launch
    uint32_t* progress;           // DEVICE [n_sblk] tiles finished per sample block (4 per layer / update phase)
    uint32_t* ctrl;               // DEVICE [SAMPLER_CTRL_WORDS]: [0, 8) workgroups seen per XCD, [8] error flag, [9] longest wait in polls
    int n_sblk;                   // sample blocks of 256
};
constexpr int SAMPLER_CTRL_WORDS = 16;
hipError_t launch_sampler_persistent(int prec, const SamplerArgs& a, int64_t n_sample_blocks, hipStream_t st);
// four workgroups of one XCD share a sample block: one channel tile each per layer, joined by a counter per block (sync = 0: no waits --
// a timing probe whose samples are garbage)
hipError_t launch_sampler_cluster(int prec, const SamplerArgs& a, int sync, hipStream_t st);

// ---- optional per-launch profiling (HIP events on the launch stream; off by default) ------------------
enum GemmEpiKind : int { EPI_GN = 0, EPI_GN_TRAIN, EPI_BIAS_SILU, EPI_ROWMAJOR, EPI_PLAIN_FT, EPI_GN_BWD, EPI_SILU_BWD, EPI_WGRAD, EPI_EM_STEP, EPI_DSM_STEP, EPI_KINDS };
constexpr int GEMM_PROF_KINDS = EPI_KINDS * 3 * GEMM_NSHAPES;      // (epilogue kind, precision, tiling)
// bf16x3 entry points (gemm_launch_x3.hip): the dispatchers of gemm_launch.hip forward prec == PREC_BF16X3 here
hipError_t gemm_gn_x3(bool train, int shape, const GemmArgs& g, const GNParams& p, hipStream_t st, int gs = 32);      // every activation, group sizes 16 / 32 / 64
hipError_t gemm_bias_silu_x3(bool train, int shape, const GemmArgs& g, const BiasSiLUParams& p, hipStream_t st);      // every activation
hipError_t gemm_silu_bwd_x3(int shape, const GemmArgs& g, const SiLUBwdParams& p, hipStream_t st);
hipError_t gemm_em_step_x3(int shape, const GemmArgs& g, const EmStepParams& p, hipStream_t st);
hipError_t gemm_partial_ft_x3(int shape, const GemmArgs& g, const PartialFTParams& p, hipStream_t st);
hipError_t gemm_gn_bwd_x3(int shape, const GemmArgs& g, const GNBwdParams& p, hipStream_t st, int gs = 32);
void gemm_prof_enable(int on);
// synchronises the recorded events, accumulates them per kind and clears the record list
int gemm_prof_collect(double* ms, long long* launches, double* flops);   // arrays of GEMM_PROF_KINDS
void gemm_prof_kind_name(int kind, char* out, int n);
