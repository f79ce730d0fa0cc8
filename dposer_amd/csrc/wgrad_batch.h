// All 256 x 256 weight-gradient tiles of a training step as ONE launch (bf16, sample-major operands: gemm_wgrad_tr.h).
//
// Ten split-K launches per step (4 x [1024 x 1024] split 16, 5 x [1024 x 512] split 32 at 65536 samples) each ramp up, drain and write
// 64 MB of fp32 slabs that k_reduce_grads reads back (640 MB per step).  Here the chip is 16 LANES of 16 workgroups -- XCD x holds
// lanes x and x + 8, so the 16 tiles of a lane share their operand panels through that XCD's L2 exactly like the tiles of one
// split of the old launches -- and the work is a line of "lane problems", each 16 tiles wide and `len` sample-block rows long:
//     W_x of a layer           : 4 x 4 tiles of dy_l^T h_{l-1}                                   len = S
//     W_t of two layers        : 4 x 2 tiles of dy_a^T temb next to 4 x 2 tiles of dy_b^T temb   len = S
//     W_t of a left-over layer : its 4 x 2 tiles twice, each half taking half of the rows        len = S / 2
//     shared embedding W_se    : its 2 x 2 tiles four times, each copy a quarter of the rows     len = S / 4   (mode 1)
// Lane i takes rows [i q, (i + 1) q) of that line (q = total / 16): at most WGB_MAX_SEG problems, one SEGMENT per problem, every
// workgroup of the lane the same segment of its own tile; a segment's partial sums are one dense 256 x 256 fp32 tile in `partials`.
// k_reduce_wgrad_tiles adds the partial tiles of each output tile in row order (fixed order: deterministic) into the flat gradient.
// Both kernels derive the partition from this struct -- no table in memory.
#pragma once
#include <cstdint>

constexpr int WGB_MAX_PROB = 8, WGB_MAX_SEG = 2, WGB_LANES = 16, WGB_BLOCKS = 256;   // (q <= the shortest problem: a lane meets at most two)
struct WgradLaneProblem {
    const void* dY[2];    // per half of the lane (slots 0-7 / 8-15): FT [Spad][16 nA]
    const void* H[2];     // FT [Spad][16 nB]
    int nA[2], nB[2];
    int sblk0[2];         // first column tile of the half
    int sb_off[2];        // added to the row index (row halves of a left-over W_t)
    int64_t dst_off[2];   // flat-gradient offset of the tensor the half writes
    int ld[2];
    int len;              // sample-block rows
    int split_k;          // 1: both halves are row ranges of the SAME 8 tiles (dst_off[0], ld[0])
    int mode;             // 0: two halves of 8 tiles (above);  1: 2 x 2 tiles x 4 row quarters of ONE tensor (index 0 of every field)
};
struct WgradBatchArgs {
    WgradLaneProblem prob[WGB_MAX_PROB];
    int nprob, q;
    float* partials;      // [WGB_BLOCKS][WGB_MAX_SEG][256 * 256]
    int nterm;            // reduction only: partial sets to add per output tile (0 / 1: one; bf16x3: the three product terms, one lane launch each)
    int64_t term_stride;  // floats between the partial sets
    int64_t span;         // bytes the 32-bit DMA offsets must cover
    double alg_flops;     // algorithmic FLOPs of the launch (profiling only)
};
// hardware places block b on XCD b % 8
__host__ __device__ inline int wgb_lane(int b) { return ((b >> 7) << 3) + (b & 7); }
__host__ __device__ inline int wgb_slot(int b) { return (b >> 3) & 15; }
__host__ __device__ inline int wgb_block(int lane, int slot) { return ((lane >> 3) << 7) + (lane & 7) + (slot << 3); }
