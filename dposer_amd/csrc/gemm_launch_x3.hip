// bf16 x 3 precision mode (DPOSER_PREC_BF16X3): instantiations of gemm_ft_kernel whose K loop runs on the bf16 matrix pipe
// (v_mfma_f32_32x32x16_bf16, the hand-placed ring stages of gemm_kloop_asm.h) while the epilogue is the fp32 mode's -- precise
// GroupNorm / SiLU arithmetic, activations stored as fp32 fragment tiles.  The caller (scorefc.hip) hands every operand as its two bf16
// planes (hi = bf16(x), lo = bf16(x - hi), k_split_ft32) in three K segments against weights packed [hi | lo | hi]: three matrix-pipe
// products per term, fp32 accumulation -- reference-precision results at a third of the bf16 rate instead of a sixteenth.
// Own translation unit: the instantiations compile next to gemm_launch.hip's, not behind them.
#include "gemm_api.h"

#ifndef DPOSER_RING_KB
#define DPOSER_RING_KB 2
#define DPOSER_RING_NB 4
#endif
namespace {
constexpr int RING_KB = DPOSER_RING_KB, RING_NB = DPOSER_RING_NB;
#ifndef DPOSER_SMALL_NB
#define DPOSER_SMALL_NB 3
#endif
constexpr int SMALL_NB = DPOSER_SMALL_NB;       // (gemm_launch.hip: three slots for the 128x32 tiling)
template <typename Epi>
hipError_t main_shapes(int shape, const GemmArgs& g, const typename Epi::Params& p, hipStream_t st) {
    switch (shape) {
        case SHAPE_BIG: return launch_gemm<__bf16, 2, 4, 4, 2, RING_KB, Epi, RING_NB>(g, p, st);
        case SHAPE_MID: return launch_gemm<__bf16, 2, 2, 2, 2, RING_KB, Epi, RING_NB>(g, p, st);
        case SHAPE_SMALL: return launch_gemm<__bf16, 4, 1, 1, 1, 4, Epi, SMALL_NB>(g, p, st);
    }
    return hipErrorInvalidConfiguration;
}
}   // namespace

template <typename Epi>
hipError_t narrow_shapes(int shape, const GemmArgs& g, const typename Epi::Params& p, hipStream_t st) {      // (runtime-selected activations: the 128-wide tilings only, as in gemm_launch.hip)
    switch (shape) {
        case SHAPE_MID: return launch_gemm<__bf16, 2, 2, 2, 2, RING_KB, Epi, RING_NB>(g, p, st);
        case SHAPE_SMALL: return launch_gemm<__bf16, 4, 1, 1, 1, 4, Epi, SMALL_NB>(g, p, st);
    }
    return hipErrorInvalidConfiguration;
}
// generic GroupNorm group sizes (hidden 512 / 2048): the tilings of gemm_launch.hip's by_shape_generic (a 64-channel group needs both of its
// tiles in one wave: the paired 128x32 form)
template <typename Epi, bool PAIR>
hipError_t generic_shapes(int shape, const GemmArgs& g, const typename Epi::Params& p, hipStream_t st) {
    if (shape == SHAPE_MID) return launch_gemm<__bf16, 2, 2, 2, 2, RING_KB, Epi, RING_NB>(g, p, st);
    if (shape == SHAPE_SMALL) {
        if constexpr (PAIR) return launch_gemm<__bf16, 2, 1, 2, 1, 4, Epi>(g, p, st);
        else return launch_gemm<__bf16, 4, 1, 1, 1, 4, Epi, SMALL_NB>(g, p, st);
    }
    return hipErrorInvalidConfiguration;
}
hipError_t gemm_gn_x3(bool train, int shape, const GemmArgs& g, const GNParams& p, hipStream_t st, int gs) {
    if (gs == 16) return train ? generic_shapes<EpiGNG<float, true, 16>, false>(shape, g, p, st) : generic_shapes<EpiGNG<float, false, 16>, false>(shape, g, p, st);
    if (gs == 64) return train ? generic_shapes<EpiGNG<float, true, 64>, true>(shape, g, p, st) : generic_shapes<EpiGNG<float, false, 64>, true>(shape, g, p, st);
    if (gs != 32) return hipErrorInvalidConfiguration;
    if (p.act != DP_ACT_SWISH) {       // elu / relu / lrelu (model.py:54-66)
        if (train) return narrow_shapes<EpiGN<float, true, -1, true>>(shape, g, p, st);
        return narrow_shapes<EpiGN<float, false, -1, true>>(shape, g, p, st);
    }
    if (train) return main_shapes<EpiGN<float, true>>(shape, g, p, st);
    return main_shapes<EpiGN<float, false>>(shape, g, p, st);
}
hipError_t gemm_bias_silu_x3(bool train, int shape, const GemmArgs& g, const BiasSiLUParams& p, hipStream_t st) {
    if (p.act != DP_ACT_SWISH) {       // elu / relu / lrelu (TimeMLPs, model.py:54-66)
        if (train) return narrow_shapes<EpiBiasSiLU<float, true, true>>(shape, g, p, st);
        return narrow_shapes<EpiBiasSiLU<float, false, true>>(shape, g, p, st);
    }
    if (train) return main_shapes<EpiBiasSiLU<float, true>>(shape, g, p, st);
    return main_shapes<EpiBiasSiLU<float, false>>(shape, g, p, st);
}
// dgrad of a Linear + activation layer on the bf16 planes, fp32-storage epilogue (TimeMLPs in bf16x3 mode)
hipError_t gemm_silu_bwd_x3(int shape, const GemmArgs& g, const SiLUBwdParams& p, hipStream_t st) {
    if (p.act != DP_ACT_SWISH) return narrow_shapes<EpiSiLUBwd<float, true>>(shape, g, p, st);
    return main_shapes<EpiSiLUBwd<float>>(shape, g, p, st);
}
hipError_t gemm_em_step_x3(int shape, const GemmArgs& g, const EmStepParams& p, hipStream_t st) {
    if (shape == SHAPE_FINAL) return launch_gemm<__bf16, 1, 4, 2, 1, RING_KB, EpiEmStep<float>, RING_NB>(g, p, st);
    if (shape == SHAPE_FINAL_S) return launch_gemm<__bf16, 2, 1, 1, 1, 4, EpiEmStep<float>, 3>(g, p, st);
    return hipErrorInvalidConfiguration;
}
hipError_t gemm_partial_ft_x3(int shape, const GemmArgs& g, const PartialFTParams& p, hipStream_t st) {
    return main_shapes<EpiPartialFT<__bf16>>(shape, g, p, st);     // (fp32 partial tiles whatever the operand type)
}
hipError_t gemm_gn_bwd_x3(int shape, const GemmArgs& g, const GNBwdParams& p, hipStream_t st, int gs) {
    if (gs == 16) return generic_shapes<EpiGNBwdG<float, 16>, false>(shape, g, p, st);
    if (gs == 64) return generic_shapes<EpiGNBwdG<float, 64>, true>(shape, g, p, st);
    if (gs != 32) return hipErrorInvalidConfiguration;
    if (p.act != DP_ACT_SWISH) return narrow_shapes<EpiGNBwd<float, 0, true>>(shape, g, p, st);
    return main_shapes<EpiGNBwd<float>>(shape, g, p, st);
}
