// Explicit instantiations of gemm_ft_kernel for every (precision, tiling, epilogue) the score
// network uses.  Kept in its own translation unit: it dominates compile time.
#include "gemm_api.h"


// Each epilogue only instantiates the tilings it is used with (bit i = GemmShape i).
constexpr unsigned M_BIG = 1u << SHAPE_BIG, M_MID = 1u << SHAPE_MID, M_SMALL = 1u << SHAPE_SMALL, M_FINAL = 1u << SHAPE_FINAL,
                   M_FINAL_S = 1u << SHAPE_FINAL_S, M_WIDE = 1u << SHAPE_WIDE64;

template <typename T, typename Epi, unsigned ALLOWED>
static hipError_t by_shape_masked(int shape, const GemmArgs& g, const typename Epi::Params& p, hipStream_t st) {
    switch (shape) {
        case SHAPE_BIG: if constexpr (ALLOWED & M_BIG) return launch_gemm<T, 2, 4, 4, 2, 4, Epi>(g, p, st); break;
        case SHAPE_MID: if constexpr (ALLOWED & M_MID) return launch_gemm<T, 2, 2, 2, 2, 4, Epi>(g, p, st); break;
        case SHAPE_SMALL: if constexpr (ALLOWED & M_SMALL) return launch_gemm<T, 4, 1, 1, 1, 4, Epi>(g, p, st); break;
        case SHAPE_FINAL: if constexpr (ALLOWED & M_FINAL) return launch_gemm<T, 1, 4, 2, 1, 4, Epi>(g, p, st); break;
        case SHAPE_FINAL_S: if constexpr (ALLOWED & M_FINAL_S) return launch_gemm<T, 2, 1, 1, 1, 4, Epi>(g, p, st); break;
        case SHAPE_WIDE64: if constexpr (ALLOWED & M_WIDE) return launch_gemm<T, 2, 2, 2, 1, 4, Epi>(g, p, st); break;
    }
    return hipErrorInvalidConfiguration;
}

#define DISPATCH(EPI16, EPI32, MASK)                                                   \
    return prec == PREC_FP32 ? by_shape_masked<float, EPI32, MASK>(shape, g, p, st)    \
                             : by_shape_masked<__bf16, EPI16, MASK>(shape, g, p, st)

constexpr unsigned M_MAIN = M_BIG | M_MID | M_SMALL;

hipError_t gemm_gn(int prec, bool train, int shape, const GemmArgs& g, const GNParams& p, hipStream_t st) {
    if (train) { typedef EpiGN<__bf16, true> A; typedef EpiGN<float, true> B; DISPATCH(A, B, M_MAIN); }
    typedef EpiGN<__bf16, false> A; typedef EpiGN<float, false> B; DISPATCH(A, B, M_MAIN);
}
hipError_t gemm_bias_silu(int prec, bool train, int shape, const GemmArgs& g, const BiasSiLUParams& p, hipStream_t st) {
    if (train) { typedef EpiBiasSiLU<__bf16, true> A; typedef EpiBiasSiLU<float, true> B; DISPATCH(A, B, M_MAIN); }
    typedef EpiBiasSiLU<__bf16, false> A; typedef EpiBiasSiLU<float, false> B; DISPATCH(A, B, M_MAIN);
}
hipError_t gemm_rowmajor(int prec, int shape, const GemmArgs& g, const RowMajorParams& p, hipStream_t st) {
    typedef EpiRowMajor<__bf16> A; typedef EpiRowMajor<float> B; DISPATCH(A, B, M_MID | M_SMALL | M_FINAL | M_FINAL_S);
}
hipError_t gemm_plain_ft(int prec, int shape, const GemmArgs& g, const PlainFTParams& p, hipStream_t st) {
    typedef EpiPlainFT<__bf16> A; typedef EpiPlainFT<float> B; DISPATCH(A, B, M_MAIN);
}
hipError_t gemm_gn_bwd(int prec, int shape, const GemmArgs& g, const GNBwdParams& p, hipStream_t st) {
    typedef EpiGNBwd<__bf16> A; typedef EpiGNBwd<float> B; DISPATCH(A, B, M_MID | M_SMALL);   // BIG spills: register-heavy epilogue
}
hipError_t gemm_silu_bwd(int prec, int shape, const GemmArgs& g, const SiLUBwdParams& p, hipStream_t st) {
    typedef EpiSiLUBwd<__bf16> A; typedef EpiSiLUBwd<float> B; DISPATCH(A, B, M_MAIN);
}
hipError_t gemm_wgrad(int prec, int shape, const GemmArgs& g, const WgradParams& p, hipStream_t st) {
    typedef EpiWgrad<__bf16> A; typedef EpiWgrad<float> B; DISPATCH(A, B, M_MID | M_FINAL | M_WIDE);
}
