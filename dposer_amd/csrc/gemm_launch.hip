// Explicit instantiations of gemm_ft_kernel for every (precision, tiling, epilogue) the score
// network uses.  Kept in its own translation unit: it dominates compile time.
#include "gemm_api.h"

#include <cstdio>
#include <vector>

// ------------------------------------------------------------------------------------------------
// launch profiler: one hipEvent pair per GEMM launch, recorded on the launch stream
// ------------------------------------------------------------------------------------------------
namespace {
struct ProfRec { hipEvent_t a, b; int kind; double flops; };
struct Prof {
    int enabled = 0;
    int only = -1;      // >= 0: record launches of this epilogue kind only
    std::vector<ProfRec> pool;
    size_t used = 0;
} g_prof;
struct ProfScope {
    ProfRec* r = nullptr;
    hipStream_t st;
    ProfScope(int kind, double alg_flops, hipStream_t s) : st(s) {
        if (!g_prof.enabled || (g_prof.only >= 0 && kind / (3 * GEMM_NSHAPES) != g_prof.only)) return;
        if (g_prof.used == g_prof.pool.size()) {
            ProfRec n;
            if (hipEventCreate(&n.a) != hipSuccess || hipEventCreate(&n.b) != hipSuccess) return;
            g_prof.pool.push_back(n);
        }
        r = &g_prof.pool[g_prof.used++];
        r->kind = kind;
        r->flops = alg_flops;
        (void)hipEventRecord(r->a, st);
    }
    ~ProfScope() { if (r) (void)hipEventRecord(r->b, st); }
};
const char* kEpiNames[EPI_KINDS] = {"gn_fwd", "gn_fwd_train", "bias_silu", "rowmajor", "plain_ft", "gn_bwd_dgrad", "silu_bwd_dgrad", "wgrad", "post_em_step", "post_dsm_step"};
const char* kShapeNames[GEMM_NSHAPES] = {"256x256", "128x128", "128x32", "64x128", "64x32", "128x64", "128x64w8"};
}   // namespace
void gemm_prof_enable(int on) {
    if (on < 0) { g_prof.enabled = 0; return; }      // pause: keep what was recorded so far
    g_prof.enabled = on != 0;
    g_prof.only = on >= 2 ? on - 2 : -1;
    if (!on) g_prof.used = 0;
}
int gemm_prof_collect(double* ms, long long* launches, double* flops) {
    for (int i = 0; i < GEMM_PROF_KINDS; ++i) { ms[i] = 0; launches[i] = 0; flops[i] = 0; }
    for (size_t i = 0; i < g_prof.used; ++i) {
        ProfRec& r = g_prof.pool[i];
        if (hipEventSynchronize(r.b) != hipSuccess) return -1;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return -1;
        ms[r.kind] += t; launches[r.kind] += 1; flops[r.kind] += r.flops;
    }
    g_prof.used = 0;
    return 0;
}
void gemm_prof_kind_name(int kind, char* out, int n) {
    const int epi = kind / (3 * GEMM_NSHAPES), prec = (kind / GEMM_NSHAPES) % 3, shape = kind % GEMM_NSHAPES;
    snprintf(out, n, "gemm_ft_kernel<%s,%s,%s>", prec == PREC_FP32 ? "fp32" : (prec == PREC_BF16X3 ? "bf16x3" : "bf16"), kShapeNames[shape], kEpiNames[epi]);
}
#define PROF(EPI) ProfScope _ps((EPI) * 3 * GEMM_NSHAPES + prec * GEMM_NSHAPES + shape, g.alg_flops, st)


// Each epilogue only instantiates the tilings it is used with (bit i = GemmShape i).
// K pipeline of every tiling but 128x32: ring of 4 slots x 2 k-blocks with the DMA spread between the MFMAs (gemm.h);
// 128x32 (one sample tile: its two activation blocks per stage do not split over four waves) keeps stages of 4 k-blocks.
#ifndef DPOSER_RING_KB
#define DPOSER_RING_KB 2
#define DPOSER_RING_NB 4
#endif
constexpr int RING_KB = DPOSER_RING_KB, RING_NB = DPOSER_RING_NB;
// 128x32, bf16: THREE slots of 4 k-blocks (round 6, tools/tune_gemm.hip TUNE_TINY2, profiles/r06_small_tile_pipeline.txt: -8 ... -16 % per launch from 128 to 3968
// samples against two slots; four slots = 80 KB leave one workgroup per CU and lose from 1280 samples up).  Same K order: bit-identical.
#ifndef DPOSER_SMALL_NB
#define DPOSER_SMALL_NB 3
#endif
// 64x32 (post_dense / dx / the fused Euler-Maruyama step at small batches), bf16: three slots of 4 k-blocks instead of four of 2 (16 stages for K = 1024
// instead of 32: 6.3 -> 5.7 us, profiles/r06_final_tile.txt).  Same K order.
#ifndef DPOSER_FINAL_S_KB
#define DPOSER_FINAL_S_KB 4
#define DPOSER_FINAL_S_NB 3
#endif
template <typename T> constexpr int small_nb() { return sizeof(T) == 2 ? DPOSER_SMALL_NB : 2; }
constexpr unsigned M_BIG = 1u << SHAPE_BIG, M_MID = 1u << SHAPE_MID, M_SMALL = 1u << SHAPE_SMALL, M_FINAL = 1u << SHAPE_FINAL,
                   M_FINAL_S = 1u << SHAPE_FINAL_S, M_WIDE = 1u << SHAPE_WIDE64, M_SMALL64 = 1u << SHAPE_SMALL64;

template <typename T, typename Epi, unsigned ALLOWED>
static hipError_t by_shape_masked(int shape, const GemmArgs& g, const typename Epi::Params& p, hipStream_t st) {
    switch (shape) {
        case SHAPE_BIG: if constexpr (ALLOWED & M_BIG) return launch_gemm<T, 2, 4, 4, 2, RING_KB, Epi, RING_NB>(g, p, st); break;
        case SHAPE_MID: if constexpr (ALLOWED & M_MID) return launch_gemm<T, 2, 2, 2, 2, RING_KB, Epi, RING_NB>(g, p, st); break;
        case SHAPE_SMALL: if constexpr (ALLOWED & M_SMALL) return launch_gemm<T, 4, 1, 1, 1, 4, Epi, small_nb<T>()>(g, p, st); break;
        case SHAPE_FINAL: if constexpr (ALLOWED & M_FINAL) return launch_gemm<T, 1, 4, 2, 1, RING_KB, Epi, RING_NB>(g, p, st); break;
        case SHAPE_FINAL_S: if constexpr (ALLOWED & M_FINAL_S) return launch_gemm<T, 2, 1, 1, 1, (sizeof(T) == 2 ? DPOSER_FINAL_S_KB : RING_KB), Epi, (sizeof(T) == 2 ? DPOSER_FINAL_S_NB : RING_NB)>(g, p, st); break;
        case SHAPE_WIDE64: if constexpr (ALLOWED & M_WIDE) return launch_gemm<T, 2, 2, 2, 1, RING_KB, Epi, RING_NB>(g, p, st); break;
        // 128 x 64 on eight waves, 4 slots x 4 k-blocks (bf16 only): between 1024 and 2048 samples the GroupNorm layers' launches take 7-11 % less than on
        // 128x32 and 128x128 (tools/tune_gemm.hip TUNE_WIDE, profiles/r06_wide_tile.txt); same K order: bit-identical
        case SHAPE_SMALL64: if constexpr ((ALLOWED & M_SMALL64) != 0 && sizeof(T) == 2) return launch_gemm<T, 4, 2, 1, 1, 4, Epi, 4>(g, p, st); break;
    }
    return hipErrorInvalidConfiguration;
}

#define DISPATCH(EPI16, EPI32, MASK)                                                   \
    return prec == PREC_FP32 ? by_shape_masked<float, EPI32, MASK>(shape, g, p, st)    \
                             : by_shape_masked<__bf16, EPI16, MASK>(shape, g, p, st)

constexpr unsigned M_MAIN = M_BIG | M_MID | M_SMALL;

// GroupNorm group sizes 16 / 64 (generic epilogues): 128x128 (wave tile 64 channels: both tiles of a 64-channel group) and 128x32
// (four waves of one tile for GS = 16, two waves of two tiles for GS = 64)
template <typename T, typename Epi, bool PAIR>
static hipError_t by_shape_generic(int shape, const GemmArgs& g, const typename Epi::Params& p, hipStream_t st) {
    if (shape == SHAPE_MID) return launch_gemm<T, 2, 2, 2, 2, RING_KB, Epi, RING_NB>(g, p, st);
    if (shape == SHAPE_SMALL) {
        if constexpr (PAIR) return launch_gemm<T, 2, 1, 2, 1, 4, Epi>(g, p, st);
        else return launch_gemm<T, 4, 1, 1, 1, 4, Epi, small_nb<T>()>(g, p, st);
    }
    return hipErrorInvalidConfiguration;
}
#define DISPATCH_GS(EPI, GS)                                                                                       \
    return prec == PREC_FP32 ? by_shape_generic<float, EPI(float, GS), (GS) == 64>(shape, g, p, st)                \
                             : by_shape_generic<__bf16, EPI(__bf16, GS), (GS) == 64>(shape, g, p, st)

hipError_t gemm_gn(int prec, bool train, int shape, const GemmArgs& g, const GNParams& p, hipStream_t st, int gs) {
    PROF(train ? EPI_GN_TRAIN : EPI_GN);
    if (prec == PREC_BF16X3) return gemm_gn_x3(train, shape, g, p, st, gs);
    if (gs != 32) {
#define E_TRAIN(T, GS) EpiGNG<T, true, GS>
#define E_INFER(T, GS) EpiGNG<T, false, GS>
        if (gs == 16) { if (train) { DISPATCH_GS(E_TRAIN, 16); } DISPATCH_GS(E_INFER, 16); }
        if (gs == 64) { if (train) { DISPATCH_GS(E_TRAIN, 64); } DISPATCH_GS(E_INFER, 64); }
        return hipErrorInvalidConfiguration;
    }
    if (p.act != DP_ACT_SWISH) {     // elu / relu / lrelu: runtime-selected activation, 128-wide tilings only
        if (train) { typedef EpiGN<__bf16, true, -1, true> A; typedef EpiGN<float, true, -1, true> B; DISPATCH(A, B, M_MID | M_SMALL); }
        typedef EpiGN<__bf16, false, -1, true> A; typedef EpiGN<float, false, -1, true> B; DISPATCH(A, B, M_MID | M_SMALL);
    }
    if (train) { typedef EpiGN<__bf16, true> A; typedef EpiGN<float, true> B; DISPATCH(A, B, M_MAIN | M_SMALL64); }
    typedef EpiGN<__bf16, false> A; typedef EpiGN<float, false> B; DISPATCH(A, B, M_MAIN | M_SMALL64);
}
hipError_t gemm_bias_silu(int prec, bool train, int shape, const GemmArgs& g, const BiasSiLUParams& p, hipStream_t st) {
    PROF(EPI_BIAS_SILU);
    if (prec == PREC_BF16X3) return gemm_bias_silu_x3(train, shape, g, p, st);
    if (p.act != DP_ACT_SWISH) {
        if (train) { typedef EpiBiasSiLU<__bf16, true, true> A; typedef EpiBiasSiLU<float, true, true> B; DISPATCH(A, B, M_MID | M_SMALL); }
        typedef EpiBiasSiLU<__bf16, false, true> A; typedef EpiBiasSiLU<float, false, true> B; DISPATCH(A, B, M_MID | M_SMALL);
    }
    if (train) { typedef EpiBiasSiLU<__bf16, true> A; typedef EpiBiasSiLU<float, true> B; DISPATCH(A, B, M_MAIN); }
    typedef EpiBiasSiLU<__bf16, false> A; typedef EpiBiasSiLU<float, false> B; DISPATCH(A, B, M_MAIN);
}
hipError_t gemm_rowmajor(int prec, int shape, const GemmArgs& g, const RowMajorParams& p, hipStream_t st) {
    PROF(EPI_ROWMAJOR);
    if (prec == PREC_BF16X3) return by_shape_masked<__bf16, EpiRowMajor<__bf16>, M_MID | M_SMALL | M_FINAL | M_FINAL_S>(shape, g, p, st);   // (fp32 row-major output whatever the operands: the bf16 instantiation)
    typedef EpiRowMajor<__bf16> A; typedef EpiRowMajor<float> B; DISPATCH(A, B, M_MID | M_SMALL | M_FINAL | M_FINAL_S);
}
hipError_t gemm_em_step(int prec, int shape, const GemmArgs& g, const EmStepParams& p, hipStream_t st) {
    PROF(EPI_EM_STEP);
    if (prec == PREC_BF16X3) return gemm_em_step_x3(shape, g, p, st);
    typedef EpiEmStep<__bf16> A; typedef EpiEmStep<float> B; DISPATCH(A, B, M_FINAL | M_FINAL_S);
}
hipError_t gemm_partial_ft(int prec, int shape, const GemmArgs& g, const PartialFTParams& p, hipStream_t st) {
    PROF(EPI_PLAIN_FT);
    if (prec == PREC_BF16X3) return gemm_partial_ft_x3(shape, g, p, st);
    typedef EpiPartialFT<__bf16> A; typedef EpiPartialFT<float> B; DISPATCH(A, B, M_MID | M_SMALL);
}
hipError_t gemm_dsm_step(int prec, int shape, const GemmArgs& g, const DsmStepParams& p, hipStream_t st) {
    PROF(EPI_DSM_STEP);
    typedef EpiDsm<__bf16> A; typedef EpiDsm<float> B; DISPATCH(A, B, M_FINAL | M_FINAL_S);
}
hipError_t gemm_plain_ft(int prec, int shape, const GemmArgs& g, const PlainFTParams& p, hipStream_t st) {
    PROF(EPI_PLAIN_FT);
    typedef EpiPlainFT<__bf16> A; typedef EpiPlainFT<float> B; DISPATCH(A, B, M_MAIN);
}
hipError_t gemm_gn_bwd(int prec, int shape, const GemmArgs& g, const GNBwdParams& p, hipStream_t st, int gs) {
    PROF(EPI_GN_BWD);
    if (prec == PREC_BF16X3) return gemm_gn_bwd_x3(shape, g, p, st, gs);
    if (gs != 32) {
#define E_BWD(T, GS) EpiGNBwdG<T, GS>
        if (gs == 16) { DISPATCH_GS(E_BWD, 16); }
        if (gs == 64) { DISPATCH_GS(E_BWD, 64); }
        return hipErrorInvalidConfiguration;
    }
    if (p.act != DP_ACT_SWISH) { typedef EpiGNBwd<__bf16, 0, true> A; typedef EpiGNBwd<float, 0, true> B; DISPATCH(A, B, M_MID | M_SMALL); }
    typedef EpiGNBwd<__bf16> A; typedef EpiGNBwd<float> B; DISPATCH(A, B, M_MAIN | M_SMALL64);
}
hipError_t gemm_silu_bwd(int prec, int shape, const GemmArgs& g, const SiLUBwdParams& p, hipStream_t st) {
    PROF(EPI_SILU_BWD);
    if (prec == PREC_BF16X3) return gemm_silu_bwd_x3(shape, g, p, st);
    if (p.act != DP_ACT_SWISH) { typedef EpiSiLUBwd<__bf16, true> A; typedef EpiSiLUBwd<float, true> B; DISPATCH(A, B, M_MID | M_SMALL); }
    typedef EpiSiLUBwd<__bf16> A; typedef EpiSiLUBwd<float> B; DISPATCH(A, B, M_MAIN);
}
hipError_t gemm_wgrad_tr_batch(const WgradBatchArgs& a, hipStream_t st) {
    ProfScope _ps(EPI_WGRAD * 3 * GEMM_NSHAPES + SHAPE_BIG, a.alg_flops, st);
    return launch_wgrad_tr_batch<2, 4, 4, 2, 4>(a, st);
}
hipError_t gemm_wgrad_tr(int shape, const WgradTrArgs& g, const WgradParams& p, hipStream_t st) {
    ProfScope _ps(EPI_WGRAD * 3 * GEMM_NSHAPES + shape, g.alg_flops, st);
    if (shape == SHAPE_BIG) return launch_wgrad_tr<2, 4, 4, 2, 4>(g, p, st);
    if (shape == SHAPE_MID) return launch_wgrad_tr<2, 2, 2, 2, 4>(g, p, st);
    if (shape == SHAPE_FINAL) return launch_wgrad_tr<1, 4, 2, 1, 4>(g, p, st);
    if (shape == SHAPE_WIDE64) return launch_wgrad_tr<2, 2, 2, 1, 4>(g, p, st);
    return hipErrorInvalidValue;
}
hipError_t gemm_wgrad(int prec, int shape, const GemmArgs& g, const WgradParams& p, hipStream_t st) {
    PROF(EPI_WGRAD);
    typedef EpiWgrad<__bf16> A; typedef EpiWgrad<float> B; DISPATCH(A, B, M_BIG | M_MID | M_FINAL | M_WIDE);
}
