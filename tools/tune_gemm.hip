// Tile-shape / epilogue-cost tuner for the score-network GEMM (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idposer_amd/csrc -Iinclude -Itools tools/tune_gemm.hip -o tools/bin/tune_gemm
//   tools/bin/tune_gemm [S]            default table        TUNE_K=1536 ...    reduction length of the layer GEMM
//   focused case lists (see main), one per run:
//     TUNE_RING      2-slot K loop vs ring pipeline per epilogue        TUNE_TAILS   ring prologue / tail paths, bit for bit
//     TUNE_WTR       wgrad from sample-major operands (tr reads) vs the plain kernel on transposed copies, bit for bit + time
//     TUNE_RESID     residual input of the GroupNorm forward epilogues  TUNE_DROPCOST  Philox dropout draws
//     TUNE_2WG       4-wave tiles, two workgroups per CU                TUNE_FINAL (+ TUNE_C=64)  64-channel output tilings
//     TUNE_GNBWD / TUNE_PLAIN / TUNE_SMALLB / TUNE_PIPE / TUNE_PROFILE  earlier studies (tilings, pipelined kernel, PMC runs)
// Times one layer GEMM (C = 1024 channels, K = 1024, S = 65536 samples) per (tiling, epilogue); the cases of a run are
// interleaved over 7 rounds and min / median are reported, because clocks drift by several percent within a process.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "epilogues.h"
#include "gemm.h"
#include "experimental/gemm_pipe.h"
#include "gemm_wgrad_tr.h"

int dposer_set_error(int code, const std::string&) { return code; }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

#include <algorithm>
#include <functional>
struct Case { std::string name; std::function<void()> launch; double flops; std::vector<double> us; };
static std::vector<Case> g_cases;
static float g_drop_p = 0.f;
static int g_ksplit = 1;                // TUNE_SPLITK: the plain cases split their reduction over blockIdx.y (every split writes the same tile: timing only)
static const void* g_resid = nullptr;   // TUNE_RESID cases: residual input of the GroupNorm forward epilogue   // TUNE_DROP=0.1: the gn-train cases draw dropout masks

template <int WC, int WS, int TC, int TS, int KB, int GLDS>
void add_plain(const char* name, int64_t S, int C, int K, void* W, void* X, void* out) {
    typedef GemmCfg<__bf16, WC, WS, TC, TS, KB> Cfg;
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.W = W; g.w_stride_blocks = K / 16; g.src[0] = X; g.seg_kblocks[0] = K / 16; g.nseg = 1; g.ktot_blocks = K / 16;
    g.n_cblk = C / (Cfg::CT * 32); g.n_sblk = (int)(S / (Cfg::ST * 32)); g.ksplit = g_ksplit;
   
    PlainFTParams p;
    p.out = out; p.N = C;
    char buf[160];
    snprintf(buf, sizeof buf, "%-8s %3dx%-3d w%d KB%d NB%d plain", name, Cfg::CT * 32, Cfg::ST * 32, Cfg::NW, KB, GLDS < 2 ? 2 : GLDS);
    g_cases.push_back({buf, [=] { CK((launch_gemm<__bf16, WC, WS, TC, TS, KB, EpiPlainFT<__bf16>, (GLDS < 2 ? 2 : GLDS)>(g, p, 0))); }, 2.0 * S * C * K, {}});
}

template <int WC, int WS, int TC, int TS, int KB, int GLDS, bool TRAIN = false>
void add_gn(const char* name, int64_t S, int C, int K, void* W, void* X, void* out, float* bias, float* gamma, float* beta, void* xhat, GnAux* rstd, void* outT) {
    typedef GemmCfg<__bf16, WC, WS, TC, TS, KB> Cfg;
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.W = W; g.w_stride_blocks = K / 16; g.src[0] = X; g.seg_kblocks[0] = K / 16; g.nseg = 1; g.ktot_blocks = K / 16;
    g.n_cblk = C / (Cfg::CT * 32); g.n_sblk = (int)(S / (Cfg::ST * 32)); g.ksplit = 1;
    GNParams p;
    memset(&p, 0, sizeof(p));
    p.bias = bias; p.gamma = gamma; p.beta = beta; p.out = out; p.H = C; p.Spad = S; p.resid = g_resid;
    if (TRAIN) { p.xhat = xhat; p.aux = rstd; p.outT = outT; }
    if (TRAIN && g_drop_p > 0.f) { p.drop.p = g_drop_p; p.drop.scale = 1.f / (1.f - g_drop_p); p.drop.thr = (uint32_t)((1.0 - g_drop_p) * 65536.0); p.drop.groups_x4 = C / 8; p.drop.seed = 7; }
    char buf[160];
    snprintf(buf, sizeof buf, "%-8s %3dx%-3d w%d KB%d NB%d %s", name, Cfg::CT * 32, Cfg::ST * 32, Cfg::NW, KB, GLDS < 2 ? 2 : GLDS, TRAIN ? "gn-train" : "gn");
    g_cases.push_back({buf, [=] { CK((launch_gemm<__bf16, WC, WS, TC, TS, KB, EpiGN<__bf16, TRAIN>, (GLDS < 2 ? 2 : GLDS)>(g, p, 0))); }, 2.0 * S * C * K, {}});
}

template <int WC, int WS, int TC, int TS, int KB, bool TRAIN = false>
void add_gn_pipe(const char* name, int64_t S, int C, int K, void* W, void* X, void* out, float* bias, float* gamma, float* beta, void* xhat, GnAux* rstd, void* outT) {
    typedef GemmCfg<__bf16, WC, WS, TC, TS, KB> Cfg;
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.W = W; g.w_stride_blocks = K / 16; g.src[0] = X; g.seg_kblocks[0] = K / 16; g.nseg = 1; g.ktot_blocks = K / 16;
    g.n_cblk = C / (Cfg::CT * 32); g.n_sblk = (int)(S / (Cfg::ST * 32)); g.ksplit = 1;
    GNParams p;
    memset(&p, 0, sizeof(p));
    p.bias = bias; p.gamma = gamma; p.beta = beta; p.out = out; p.H = C; p.Spad = S;
    if (TRAIN) { p.xhat = xhat; p.aux = rstd; p.outT = outT; }
    char buf[160];
    p.drop.thr = 65536; p.drop.scale = 1.f; p.drop.groups_x4 = C / 8;
    snprintf(buf, sizeof buf, "%-8s %3dx%-3d w%d KB%d PIPE %s", name, Cfg::CT * 32, Cfg::ST * 32, Cfg::NW, KB, TRAIN ? "gn-train" : "gn");
    g_cases.push_back({buf, [=] { CK((launch_gemm_pipe<__bf16, WC, WS, TC, TS, KB, EpiGN<__bf16, TRAIN, 0>>(g, p, 0))); }, 2.0 * S * C * K, {}});
}
template <int WC, int WS, int TC, int TS, int KB>
void add_plain_pipe(const char* name, int64_t S, int C, int K, void* W, void* X, void* out) {
    typedef GemmCfg<__bf16, WC, WS, TC, TS, KB> Cfg;
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.W = W; g.w_stride_blocks = K / 16; g.src[0] = X; g.seg_kblocks[0] = K / 16; g.nseg = 1; g.ktot_blocks = K / 16;
    g.n_cblk = C / (Cfg::CT * 32); g.n_sblk = (int)(S / (Cfg::ST * 32)); g.ksplit = 1;
    PlainFTParams p;
    p.out = out; p.N = C;
    char buf[160];
    snprintf(buf, sizeof buf, "%-8s %3dx%-3d w%d KB%d PIPE plain", name, Cfg::CT * 32, Cfg::ST * 32, Cfg::NW, KB);
    g_cases.push_back({buf, [=] { CK((launch_gemm_pipe<__bf16, WC, WS, TC, TS, KB, EpiPlainFT<__bf16>>(g, p, 0))); }, 2.0 * S * C * K, {}});
}

template <int WC, int WS, int TC, int TS, int KB, int GLDS, int ABL = 0>
void add_gnbwd(const char* name, int64_t S, int C, int K, void* Wt, void* dyn, void* dy, void* xhat, GnAux* rstd, float* gamma, float* beta,
               float* part, void* dyT, void* carry_in, void* carry_out, float drop_p) {
    typedef GemmCfg<__bf16, WC, WS, TC, TS, KB> Cfg;
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.W = Wt; g.w_stride_blocks = K / 16; g.src[0] = dyn; g.seg_kblocks[0] = K / 16; g.nseg = 1; g.ktot_blocks = K / 16;
    g.n_cblk = C / (Cfg::CT * 32); g.n_sblk = (int)(S / (Cfg::ST * 32)); g.ksplit = 1;
    GNBwdParams p;
    memset(&p, 0, sizeof(p));
    p.carry_in = carry_in; p.carry_out = carry_out; p.xhat = xhat; p.aux = rstd; p.gamma = gamma; p.beta = beta; p.dy = dy; p.part = part;
    p.H = C; p.S_valid = S; p.dyT = dyT; p.Spad = S;
    p.drop_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    char buf[160];
    snprintf(buf, sizeof buf, "%-8s %3dx%-3d w%d KB%d NB%d gnbwd abl%d drop%d T%d carry%d%d", name, Cfg::CT * 32, Cfg::ST * 32, Cfg::NW, KB, GLDS < 2 ? 2 : GLDS, ABL, drop_p > 0.f,
             dyT != nullptr, carry_in != nullptr, carry_out != nullptr);
    g_cases.push_back({buf, [=] { CK((launch_gemm<__bf16, WC, WS, TC, TS, KB, EpiGNBwd<__bf16, ABL>, (GLDS < 2 ? 2 : GLDS)>(g, p, 0))); }, 2.0 * S * C * K, {}});
}

static void run_all(int rounds, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (auto& c : g_cases) { c.launch(); }
    CK(hipDeviceSynchronize());
    for (int r = 0; r < rounds; ++r)
        for (auto& c : g_cases) {
            c.launch();
            CK(hipEventRecord(a, 0));
            for (int i = 0; i < reps; ++i) c.launch();
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            c.us.push_back(ms * 1e3 / reps);
        }
    for (auto& c : g_cases) {
        std::sort(c.us.begin(), c.us.end());
        const double mn = c.us.front(), md = c.us[c.us.size() / 2];
        printf("%-44s min %7.1f us (%6.0f TF)  median %7.1f us (%6.0f TF)\n", c.name.c_str(), mn, c.flops / mn * 1e-6, md, c.flops / md * 1e-6);
    }
}

int main(int argc, char** argv) {
    const int64_t S = argc > 1 ? atoll(argv[1]) : 65536;
    const int C = getenv("TUNE_C") ? atoi(getenv("TUNE_C")) : 1024, K = getenv("TUNE_K") ? atoi(getenv("TUNE_K")) : 1024;
    void *W, *X, *o0, *o1;
    float *bias, *gamma, *beta;
    CK(hipMalloc(&W, (size_t)C * K * 2)); CK(hipMalloc(&X, (size_t)S * K * 2)); CK(hipMalloc(&o0, (size_t)S * C * 2)); CK(hipMalloc(&o1, (size_t)S * C * 2));
    CK(hipMalloc(&bias, C * 4)); CK(hipMalloc(&gamma, C * 4)); CK(hipMalloc(&beta, C * 4));
    std::vector<unsigned short> hw((size_t)C * K), hx((size_t)S * K);
    srand(1);
    auto rnd = [] { float f = (rand() / (float)RAND_MAX - 0.5f) * 0.2f; unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
    for (auto& v : hw) v = rnd();
    for (auto& v : hx) v = rnd();
    CK(hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(X, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    std::vector<float> hb(C, 0.01f), hg(C, 1.0f);
    CK(hipMemcpy(bias, hb.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(gamma, hg.data(), C * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(beta, hb.data(), C * 4, hipMemcpyHostToDevice));
    printf("forward GN layer GEMM: S=%lld C=%d K=%d bf16\n", (long long)S, C, K);
    void *xhat; GnAux* rstd; void* outT;
    CK(hipMalloc(&xhat, (size_t)S * C * 2)); CK(hipMalloc(&rstd, (size_t)(S / 32) * (C / 32) * 64 * sizeof(GnAux))); CK(hipMalloc(&outT, (size_t)S * C * 2));
#define GN(WC, WS, TC, TS, KB, G) add_gn<WC, WS, TC, TS, KB, G>(#WC "," #WS "," #TC "," #TS, S, C, K, W, X, o1, bias, gamma, beta, xhat, rstd, outT)
#define GNT(WC, WS, TC, TS, KB, G) add_gn<WC, WS, TC, TS, KB, G, true>(#WC "," #WS "," #TC "," #TS, S, C, K, W, X, o1, bias, gamma, beta, xhat, rstd, outT)
#define PL(WC, WS, TC, TS, KB, G) add_plain<WC, WS, TC, TS, KB, G>(#WC "," #WS "," #TC "," #TS, S, C, K, W, X, o1)
    float* part; void *cin, *cout;
    CK(hipMalloc(&part, (size_t)(S / 32) * 3 * C * 4)); CK(hipMalloc(&cin, (size_t)S * C * 2)); CK(hipMalloc(&cout, (size_t)S * C * 2));
    CK(hipMemset(rstd, 0x3f, (size_t)(S / 32) * (C / 32) * 64 * sizeof(GnAux))); CK(hipMemset(xhat, 0, (size_t)S * C * 2)); CK(hipMemset(cin, 0, (size_t)S * C * 2));
#define GB(WC, WS, TC, TS, KB, G, DROP, DYT, CI, CO) add_gnbwd<WC, WS, TC, TS, KB, G>(#WC "," #WS "," #TC "," #TS, S, C, K, W, X, o1, xhat, rstd, gamma, beta, part, DYT ? outT : nullptr, CI ? cin : nullptr, CO ? cout : nullptr, DROP ? 0.1f : 0.f)
    if (getenv("TUNE_R3")) {         // round 3: training epilogues (dropout draw, residual / record prefetch by asm-issued DMA); run old and new builds
        g_drop_p = 0.1f;
        PL(2, 4, 4, 2, 2, 4);
        GNT(2, 4, 4, 2, 2, 4); g_cases.back().name += " +dropout";
        g_resid = cin;
        GNT(2, 4, 4, 2, 2, 4); g_cases.back().name += " +dropout +resid";
        g_resid = nullptr;
        GB(2, 4, 4, 2, 2, 4, 1, 0, 0, 0);
        GB(2, 4, 4, 2, 2, 4, 1, 0, 1, 0);
        GB(2, 4, 4, 2, 2, 4, 1, 0, 1, 1);
        if (getenv("TUNE_GNBWD_PACK")) {       // packed (default) vs scalar inner loop of the GroupNorm-backward epilogue (ABL bit 32), same binary
            add_gnbwd<2, 4, 4, 2, 2, 4, 32>("2,4,4,2 scalar-loop", S, C, K, W, X, o1, xhat, rstd, gamma, beta, part, nullptr, nullptr, nullptr, 0.1f);
            add_gnbwd<2, 4, 4, 2, 2, 4, 32>("2,4,4,2 scalar-loop carry11", S, C, K, W, X, o1, xhat, rstd, gamma, beta, part, nullptr, cin, cout, 0.1f);
        }
        run_all(7, 10);
        return 0;
    }
    if (getenv("TUNE_PIPE")) {
        // correctness: pipelined vs reference kernel, bit for bit
        {
            add_gn<2, 4, 4, 2, 4, 1>("ref", S, C, K, W, X, o0, bias, gamma, beta, xhat, rstd, outT);
            add_gn_pipe<2, 2, 4, 2, 4>("pipe", S, C, K, W, X, o1, bias, gamma, beta, xhat, rstd, outT);
            CK(hipMemset(o0, 0, (size_t)S * C * 2)); CK(hipMemset(o1, 0xff, (size_t)S * C * 2));
            g_cases[0].launch(); g_cases[1].launch();
            CK(hipDeviceSynchronize());
            std::vector<unsigned short> h0((size_t)S * C), h1((size_t)S * C);
            CK(hipMemcpy(h0.data(), o0, h0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), o1, h1.size() * 2, hipMemcpyDeviceToHost));
            size_t bad = 0, first = 0;
            for (size_t i = 0; i < h0.size(); ++i) if (h0[i] != h1[i]) { if (!bad) first = i; ++bad; }
            printf("pipelined vs reference GN output: %zu mismatches of %zu (first at %zu)\n", bad, h0.size(), first);
            g_cases.clear();
        }
#define GNP(WC, WS, TC, TS, KB) add_gn_pipe<WC, WS, TC, TS, KB>(#WC "," #WS "," #TC "," #TS, S, C, K, W, X, o1, bias, gamma, beta, xhat, rstd, outT)
#define GNPT(WC, WS, TC, TS, KB) add_gn_pipe<WC, WS, TC, TS, KB, true>(#WC "," #WS "," #TC "," #TS, S, C, K, W, X, o1, bias, gamma, beta, xhat, rstd, outT)
#define PLP(WC, WS, TC, TS, KB) add_plain_pipe<WC, WS, TC, TS, KB>(#WC "," #WS "," #TC "," #TS, S, C, K, W, X, o1)
        GN(2, 4, 4, 2, 4, 1);
        GNP(2, 2, 4, 2, 4);
        GNP(1, 4, 4, 2, 4);
        GNP(4, 2, 2, 2, 4);
        GNP(2, 4, 2, 2, 4);
        GNT(2, 4, 4, 2, 4, 1);
        GNPT(2, 2, 4, 2, 4);
        GNPT(4, 2, 2, 2, 4);
        PL(2, 4, 4, 2, 4, 1);
        PLP(2, 2, 4, 2, 4);
        PLP(4, 2, 2, 2, 4);
        run_all(7, 10);
        return 0;
    }
    if (getenv("TUNE_TINY")) {       // round 2: latency-bound layers (run with S = 128 / 512 / 2048): K pipeline depth and tilings of the small shapes
        GN(4, 1, 1, 1, 4, 2);        // shipped 128x32: 2 slots x 4 k-blocks
        GN(4, 1, 1, 1, 4, 4);        // ring of 4 slots x 4 k-blocks
        GN(2, 1, 1, 1, 4, 4);        // 64x32, 2 waves
        GN(2, 1, 1, 1, 2, 4);
        GN(2, 2, 1, 1, 2, 4);        // 64x64, 4 waves
        GN(1, 2, 1, 1, 2, 4);        // 32x64, 2 waves
        PL(4, 1, 1, 1, 4, 2);
        PL(4, 1, 1, 1, 4, 4);
        run_all(9, 50);
        return 0;
    }
    if (getenv("TUNE_LDS")) {        // round 6: the 128x128 tiling moves 1.5 KB through LDS per MFMA (1.0 fragment reads + 0.5 DMA), the CU's LDS delivers 1 KB per
                                     // MFMA slot -- four-wave tiles with 128x64 / 64x128 wave tiles (0.75 + 0.375) for the one-round batches (8192 / 16384)?
        g_drop_p = 0.1f;
        GNT(2, 2, 2, 2, 2, 4);       // shipped: 128 x 128 (asm stage)
        GNT(2, 2, 4, 2, 2, 4);       // 256 ch x 128 s, 4 waves, wave tile 128 x 64
        GNT(2, 2, 4, 2, 2, 3);
        GNT(2, 2, 2, 4, 2, 4);       // 128 ch x 256 s, 4 waves, wave tile 64 x 128
        GNT(2, 2, 2, 4, 2, 3);
        GNT(2, 4, 4, 2, 2, 4);       // 256 x 256 (asm stage)
        GB(2, 2, 2, 2, 2, 4, 1, 0, 0, 0);
        GB(2, 2, 4, 2, 2, 4, 1, 0, 0, 0);
        GB(2, 2, 4, 2, 2, 3, 1, 0, 0, 0);
        GB(2, 2, 2, 4, 2, 4, 1, 0, 0, 0);
        GB(2, 2, 2, 4, 2, 3, 1, 0, 0, 0);
        GB(2, 4, 4, 2, 2, 4, 1, 0, 0, 0);
        PL(2, 2, 2, 2, 2, 4); PL(2, 2, 4, 2, 2, 4); PL(2, 2, 4, 2, 2, 3); PL(2, 2, 2, 4, 2, 4); PL(2, 2, 2, 4, 2, 3); PL(2, 4, 4, 2, 2, 4);
        run_all(9, 20);
        return 0;
    }
    if (getenv("TUNE_WIDE")) {       // round 6: 128 x 64 tilings for 1024 ... 4096 samples against the shipped 128x32 (3 slots) and 128x128 (asm)
        GNT(4, 1, 1, 1, 4, 3);       // shipped <= 2048
        GNT(2, 2, 2, 2, 2, 4);       // shipped above
        GNT(4, 2, 1, 1, 4, 3);       // 128 x 64, 8 waves, wave tile 32 x 32
        GNT(4, 2, 1, 1, 4, 4);
        GNT(2, 2, 2, 1, 2, 4);       // 128 x 64, 4 waves, wave tile 64 x 32 (SHAPE_WIDE64's pipeline)
        GNT(2, 2, 2, 1, 4, 3);
        GNT(2, 2, 2, 1, 4, 4);
        GNT(2, 1, 2, 2, 4, 3);       // 128 x 64, 2 waves, wave tile 64 x 64
        GB(4, 1, 1, 1, 4, 3, 1, 0, 0, 0);
        GB(2, 2, 2, 2, 2, 4, 1, 0, 0, 0);
        GB(4, 2, 1, 1, 4, 3, 1, 0, 0, 0);
        GB(4, 2, 1, 1, 4, 4, 1, 0, 0, 0);
        GB(2, 2, 2, 1, 2, 4, 1, 0, 0, 0);
        GB(2, 2, 2, 1, 4, 3, 1, 0, 0, 0);
        GB(2, 2, 2, 1, 4, 4, 1, 0, 0, 0);
        GB(2, 1, 2, 2, 4, 3, 1, 0, 0, 0);
        GN(4, 1, 1, 1, 4, 3);
        GN(2, 2, 2, 2, 2, 4);
        GN(4, 2, 1, 1, 4, 3);
        GN(2, 2, 2, 1, 4, 3);
        GN(2, 2, 2, 1, 2, 4);
        run_all(9, 50);
        return 0;
    }
    if (getenv("TUNE_TINY2")) {      // round 6: the 128x32 tiling of the small batches (sampler at 500 samples, training at <= 1280): deeper K pipelines, wider tiles
        GN(4, 1, 1, 1, 4, 2);        // shipped until round 6: 2 slots x 4 k-blocks
        GN(4, 1, 1, 1, 4, 3);
        GN(4, 2, 1, 1, 4, 3);        // 128 x 64, 8 waves
        GN(4, 2, 1, 1, 4, 4);
        GN(2, 2, 1, 1, 4, 3);        // 64 x 64, 4 waves
        GN(2, 2, 2, 2, 2, 4);        // 128 x 128 (asm stage)
        GNT(4, 1, 1, 1, 4, 2);
        GNT(4, 1, 1, 1, 4, 3);
        GNT(4, 2, 1, 1, 4, 3);
        GNT(4, 2, 1, 1, 4, 4);
        GNT(2, 2, 1, 1, 4, 3);
        GNT(2, 2, 2, 2, 2, 4);
        GB(4, 1, 1, 1, 4, 2, 1, 0, 0, 0);
        GB(4, 1, 1, 1, 4, 3, 1, 0, 0, 0);
        GB(4, 2, 1, 1, 4, 3, 1, 0, 0, 0);
        GB(4, 2, 1, 1, 4, 4, 1, 0, 0, 0);
        GB(2, 2, 1, 1, 4, 3, 1, 0, 0, 0);
        GB(2, 2, 2, 2, 2, 4, 1, 0, 0, 0);
        run_all(9, 50);
        return 0;
    }
    if (getenv("TUNE_SPLITK")) {     // round 4: the time-branch dgrad at small batches is 40 ... 256 tiles with K = 5120 (run with TUNE_C=512 TUNE_K=5120 and
                                     // S = 1280 / 4096 / 8192): what would splitting the reduction over the five layers' segments buy?  (timing probe: the
                                     // splits overwrite one another's tile)
        for (int ks : {1, 2, 5, 10}) {
            g_ksplit = ks;
            PL(2, 2, 2, 2, 2, 4);
            g_cases.back().name += " ksplit " + std::to_string(ks);
        }
        g_ksplit = 1;
        run_all(9, 20);
        return 0;
    }
    if (getenv("TUNE_TALL")) {       // round 4: one-round batches (S = 8192 / 16384) are L2-bound on the 128x128 tiling (64 FLOP per byte staged): do 256x128 /
                                     // 128x256 tiles (85 FLOP per byte; 256 / 512 tiles) on the compiler's ring loop beat the hand-placed 128x128 stage?
        g_drop_p = 0.1f;
        GNT(2, 2, 2, 2, 2, 4);   // shipped below 16384 samples: 128 x 128, 4 waves, two workgroups per CU, asm stage
        GNT(4, 2, 2, 2, 2, 4);   // 256 ch x 128 s, 8 waves, wave tile 64 x 64
        GNT(2, 4, 2, 2, 2, 4);   // 128 ch x 256 s, 8 waves
        GNT(2, 4, 4, 1, 2, 4);   // 256 ch x 128 s, 8 waves, wave tile 128 x 32
        GNT(4, 2, 2, 2, 2, 3);   // three slots: two workgroups per CU
        GNT(2, 4, 4, 2, 2, 4);   // 256 x 256
        GB(2, 2, 2, 2, 2, 4, 1, 0, 0, 0);
        GB(4, 2, 2, 2, 2, 4, 1, 0, 0, 0);
        GB(2, 4, 2, 2, 2, 4, 1, 0, 0, 0);
        GB(2, 4, 4, 1, 2, 4, 1, 0, 0, 0);
        GB(2, 4, 4, 2, 2, 4, 1, 0, 0, 0);
        PL(2, 2, 2, 2, 2, 4); PL(4, 2, 2, 2, 2, 4); PL(2, 4, 2, 2, 2, 4); PL(2, 4, 4, 1, 2, 4); PL(2, 4, 4, 2, 2, 4);
        run_all(9, 20);
        return 0;
    }
    if (getenv("TUNE_SMALLB")) {     // which tiling for one-round problem sizes (run with S = 8192 / 16384)?
        GNT(2, 2, 2, 2, 4, 1);
        GNT(2, 2, 2, 1, 4, 1);
        GNT(4, 1, 1, 2, 4, 1);
        GNT(2, 4, 4, 2, 4, 1);
        GB(2, 2, 2, 2, 4, 1, 1, 1, 0, 0);
        GB(2, 2, 2, 1, 4, 1, 1, 1, 0, 0);
        GB(4, 1, 1, 2, 4, 1, 1, 1, 0, 0);
        run_all(9, 20);
        return 0;
    }
    if (getenv("TUNE_2WGT")) {       // round 2: training-forward epilogue (three output streams) with two 4-wave workgroups per CU
        g_drop_p = 0.1f;
        GNT(2, 4, 4, 2, 2, 4);
        GNT(2, 2, 4, 2, 2, 3);
        GNT(1, 4, 4, 2, 2, 3);
        GNT(4, 1, 2, 4, 2, 3);
        GNT(2, 2, 2, 4, 2, 3);
        g_resid = cin;
        GNT(2, 4, 4, 2, 2, 4); g_cases.back().name += " +resid";
        GNT(2, 2, 4, 2, 2, 3); g_cases.back().name += " +resid";
        run_all(7, 10);
        return 0;
    }
    if (getenv("TUNE_4W")) {         // round 2: one wave per SIMD with 128x128 wave tiles (0.5 LDS fragment reads per MFMA instead of 0.75)
        PL(2, 4, 4, 2, 2, 4);   // shipped: 8 waves, wave tile 128 x 64
        PL(2, 2, 4, 4, 2, 4);   // 4 waves, wave tile 128 x 128, ring of 4
        PL(2, 2, 4, 4, 2, 3);
        PL(2, 2, 4, 4, 4, 2);   // 2-slot loop, 4 k-blocks per stage
        GN(2, 4, 4, 2, 2, 4);
        GN(2, 2, 4, 4, 2, 4);
        run_all(7, 10);
        return 0;
    }
    if (getenv("TUNE_GNBWD2")) {     // round 2: where the GroupNorm-backward epilogue spends its time (ablations) and the trimmed version
#define GBA(WC, WS, TC, TS, ABL, DROP, CI, CO) add_gnbwd<WC, WS, TC, TS, 2, 4, ABL>(#WC "," #WS "," #TC "," #TS, S, C, K, W, X, o1, xhat, rstd, gamma, beta, part, nullptr, CI ? cin : nullptr, CO ? cout : nullptr, DROP ? 0.1f : 0.f)
        PL(2, 4, 4, 2, 2, 4);
        GBA(2, 4, 4, 2, 0, 1, 0, 0);      // shipped tiling, trimmed epilogue
        GBA(2, 4, 4, 2, 4, 1, 0, 0);      // ... with the ds_bpermute butterflies
        GBA(2, 4, 4, 2, 1, 1, 0, 0);      // no parameter-gradient sums
        GBA(2, 4, 4, 2, 2, 1, 0, 0);      // no SiLU'
        GBA(2, 4, 4, 2, 3, 1, 0, 0);      // neither
        GBA(2, 4, 4, 2, 8, 1, 0, 0);      // loads / stores only
        GBA(4, 2, 2, 4, 0, 1, 0, 0);      // wave tile 64 channels x 128 samples: half as many butterflies
        GBA(2, 4, 4, 2, 0, 1, 1, 1);      // with the residual carry in and out
        GBA(4, 2, 2, 4, 0, 1, 1, 1);
        GNT(2, 4, 4, 2, 2, 4);
        g_drop_p = 0.1f;
        GNT(2, 4, 4, 2, 2, 4); g_cases.back().name += " +dropout";
        run_all(7, 10);
        return 0;
    }
    if (getenv("TUNE_GNBWD")) {
        for (int rep = 0; rep < 2; ++rep) {
            GB(2, 2, 2, 2, 4, 1, 1, 1, 0, 0);
            GB(4, 2, 2, 2, 4, 1, 1, 1, 0, 0);
            GB(2, 4, 2, 2, 4, 1, 1, 1, 0, 0);
            GB(4, 2, 2, 2, 2, 1, 1, 1, 0, 0);
            GB(2, 2, 2, 2, 2, 1, 1, 1, 0, 0);
        }
        PL(2, 2, 2, 2, 4, 1);
        PL(4, 2, 2, 2, 4, 1);
        run_all(7, 10);
        return 0;
    }
    if (getenv("TUNE_TAILS")) {      // ring prologue / tail paths: 1..6 stages, ring (NB 4 and 3) against the 2-slot loop, bit for bit
        size_t total_bad = 0;
        const int kt_max = getenv("TUNE_TAILS_MAXK") ? atoi(getenv("TUNE_TAILS_MAXK")) : 192;      // e.g. 1536: covers the hand-placed stage groups (from 7 stages up)
        for (int Kt = 32; Kt <= kt_max && Kt <= K; Kt += 32) {
            g_cases.clear();
            add_plain<2, 2, 2, 2, 2, 2>("ref", S, C, Kt, W, X, o0);
            add_plain<2, 2, 2, 2, 2, 4>("ring4", S, C, Kt, W, X, o1);
            add_plain<2, 4, 4, 2, 2, 3>("ring3", S, C, Kt, W, X, xhat);
            CK(hipMemset(o0, 0, (size_t)S * C * 2)); CK(hipMemset(o1, 0xff, (size_t)S * C * 2)); CK(hipMemset(xhat, 0xee, (size_t)S * C * 2));
            for (auto& c : g_cases) c.launch();
            CK(hipDeviceSynchronize());
            std::vector<unsigned short> h0((size_t)S * C), h1((size_t)S * C), h2((size_t)S * C);
            CK(hipMemcpy(h0.data(), o0, h0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), o1, h1.size() * 2, hipMemcpyDeviceToHost));
            CK(hipMemcpy(h2.data(), xhat, h2.size() * 2, hipMemcpyDeviceToHost));
            size_t bad1 = 0, bad2 = 0, nz = 0;
            for (size_t i = 0; i < h0.size(); ++i) { bad1 += h0[i] != h1[i]; bad2 += h0[i] != h2[i]; nz += h0[i] != 0; }
            if (bad1 && getenv("TUNE_TAILS_DEBUG")) {        // where do the mismatches sit inside the 128x128 tile (32x32 sub-tiles: sample block x channel block)?
                size_t hist[4][4] = {};
                for (int64_t s_ = 0; s_ < S; ++s_)
                    for (int c = 0; c < C; ++c) {
                        const size_t i = (size_t)FT<__bf16>::index(s_, c, C);
                        if (h0[i] != h1[i]) hist[(s_ % 128) / 32][(c % 128) / 32]++;
                    }
                for (int a_ = 0; a_ < 4; ++a_) printf("   sample block %d: %zu %zu %zu %zu (channel blocks 0..3)\n", a_, hist[a_][0], hist[a_][1], hist[a_][2], hist[a_][3]);
            }
            // the shipped 256x256 ring (NB = 4: hand-placed steady-state stages, gemm_kloop_asm.h)
            g_cases.clear();
            add_plain<2, 4, 4, 2, 2, 4>("ring4 256", S, C, Kt, W, X, o1);
            add_plain<2, 2, 2, 2, 2, 4>("ring4 128 (again, into xhat)", S, C, Kt, W, X, xhat);
            CK(hipMemset(o1, 0xdd, (size_t)S * C * 2));
            for (auto& c : g_cases) c.launch();
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(h1.data(), o1, h1.size() * 2, hipMemcpyDeviceToHost));
            size_t bad3 = 0;
            for (size_t i = 0; i < h0.size(); ++i) bad3 += h0[i] != h1[i];
            printf("K = %4d (%2d stages): ring NB4 128x128 %zu, ring NB3 256x256 %zu, ring NB4 256x256 %zu mismatches of %zu (%zu non-zero)\n", Kt, Kt / 32, bad1, bad2, bad3, h0.size(), nz);
            total_bad += bad1 + bad2 + bad3;
        }
        printf(total_bad ? "TAILS FAILED\n" : "TAILS OK\n");
        return total_bad != 0;
    }
    if (getenv("TUNE_WTR")) {        // wgrad from un-transposed operands (ds_read_b64_tr_b16) vs the plain kernel on transposed copies
        const int N = 1024, Kc = C;                    // dW [N][Kc], contraction over S samples
        std::vector<unsigned short> hy((size_t)S * N), hyT((size_t)S * N), hh((size_t)S * Kc), hhT((size_t)S * Kc);
        for (auto& v : hy) v = rnd();
        for (auto& v : hh) v = rnd();
        for (int64_t s_ = 0; s_ < S; ++s_) {
            for (int n = 0; n < N; ++n) hyT[FT<__bf16>::index(n, (int)s_, (int)S)] = hy[FT<__bf16>::index(s_, n, N)];
            for (int k = 0; k < Kc; ++k) hhT[FT<__bf16>::index(k, (int)s_, (int)S)] = hh[FT<__bf16>::index(s_, k, Kc)];
        }
        void *dy, *dyT_, *hb, *hbT;
        CK(hipMalloc(&dy, hy.size() * 2)); CK(hipMalloc(&dyT_, hy.size() * 2)); CK(hipMalloc(&hb, hh.size() * 2)); CK(hipMalloc(&hbT, hh.size() * 2));
        CK(hipMemcpy(dy, hy.data(), hy.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dyT_, hyT.data(), hy.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(hb, hh.data(), hh.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(hbT, hhT.data(), hh.size() * 2, hipMemcpyHostToDevice));
        const int ks = 16;
        float *slab0, *slab1;
        CK(hipMalloc(&slab0, (size_t)ks * N * Kc * 4)); CK(hipMalloc(&slab1, (size_t)ks * N * Kc * 4));
        CK(hipMemset(slab0, 0, (size_t)ks * N * Kc * 4)); CK(hipMemset(slab1, 0xff, (size_t)ks * N * Kc * 4));
        WgradParams wp0, wp1;
        wp0.slab = slab0; wp0.slab_stride = (int64_t)N * Kc; wp0.ld = Kc; wp0.N_valid = N; wp0.K_valid = Kc;
        wp1 = wp0; wp1.slab = slab1;
        GemmArgs g0;
        memset(&g0, 0, sizeof(g0));
        g0.W = dyT_; g0.w_stride_blocks = (int)(S / 16); g0.src[0] = hbT; g0.seg_kblocks[0] = (int)(S / 16); g0.nseg = 1; g0.ktot_blocks = (int)(S / 16);
        const bool mid = getenv("TUNE_WTR_MID") != nullptr;            // the 128x128 / 4-wave tiling of both kernels
        const int tile = mid ? 128 : 256;
        g0.n_cblk = N / tile; g0.n_sblk = Kc / tile; g0.ksplit = ks;
        WgradTrArgs g1;
        memset(&g1, 0, sizeof(g1));
        g1.dY = dy; g1.H = hb; g1.N = N; g1.Kc = Kc; g1.n_cblk = N / tile; g1.n_sblk = Kc / tile; g1.sblocks = (int)(S / 32); g1.ksplit = ks;
        const double fl = 2.0 * S * N * Kc;
        if (mid) {
            g_cases.push_back({"wgrad 128x128, transposed copies (plain kernel)", [=] { CK((launch_gemm<__bf16, 2, 2, 2, 2, 2, EpiWgrad<__bf16>, 4>(g0, wp0, 0))); }, fl, {}});
            g_cases.push_back({"wgrad 128x128, sample-major operands (tr reads)", [=] { CK((launch_wgrad_tr<2, 2, 2, 2, 4>(g1, wp1, 0))); }, fl, {}});
        } else {
        g_cases.push_back({"wgrad 256x256, transposed copies (plain kernel)", [=] { CK((launch_gemm<__bf16, 2, 4, 4, 2, 2, EpiWgrad<__bf16>, 4>(g0, wp0, 0))); }, fl, {}});
        g_cases.push_back({"wgrad 256x256, sample-major operands (tr reads)", [=] { CK((launch_wgrad_tr<2, 4, 4, 2, 4>(g1, wp1, 0))); }, fl, {}});
        }
        for (auto& c : g_cases) c.launch();
        CK(hipDeviceSynchronize());
        {
            std::vector<float> a((size_t)ks * N * Kc), b((size_t)ks * N * Kc);
            CK(hipMemcpy(a.data(), slab0, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), slab1, b.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0, nz = 0;
            for (size_t i = 0; i < a.size(); ++i) { bad += memcmp(&a[i], &b[i], 4) != 0; nz += a[i] != 0.f; }
            printf("tr-read wgrad vs plain wgrad slabs: %zu mismatches of %zu (%zu non-zero), e.g. %g vs %g\n", bad, a.size(), nz, a[12345], b[12345]);
        }
        run_all(7, 10);
        return 0;
    }
    if (getenv("TUNE_FINAL")) {      // 64-channel outputs (post_dense, dx; run with TUNE_C=64): K pipeline of the 64x128 / 64x32 tilings
        PL(1, 4, 2, 1, 2, 4);
        PL(1, 4, 2, 1, 4, 2);
        PL(1, 4, 2, 1, 4, 3);
        PL(1, 4, 2, 1, 4, 4);
        PL(1, 4, 2, 1, 8, 2);
        PL(1, 4, 2, 1, 8, 3);
        PL(2, 1, 1, 1, 2, 4);
        PL(2, 1, 1, 1, 4, 4);
        PL(2, 1, 1, 1, 4, 3);
        PL(2, 1, 1, 1, 8, 3);
        PL(2, 1, 1, 1, 8, 4);
        PL(2, 1, 1, 1, 16, 2);
        PL(2, 2, 1, 1, 2, 4);
        PL(2, 2, 1, 1, 4, 4);
        PL(2, 2, 1, 2, 2, 4);
        PL(2, 2, 1, 2, 4, 4);
        run_all(7, 20);
        return 0;
    }
    if (getenv("TUNE_RESID")) {      // cost of the residual input in the GroupNorm forward epilogues
        GN(2, 4, 4, 2, 2, 4); GNT(2, 4, 4, 2, 2, 4); GN(2, 2, 2, 2, 2, 4);
        g_resid = cin;
        GN(2, 4, 4, 2, 2, 4); g_cases.back().name += " +resid";
        GNT(2, 4, 4, 2, 2, 4); g_cases.back().name += " +resid";
        GN(2, 2, 2, 2, 2, 4); g_cases.back().name += " +resid";
        run_all(7, 10);
        return 0;
    }
    if (getenv("TUNE_2WG")) {        // 4-wave tiles with 128-register accumulators, two workgroups per CU (ring NB = 3), vs 256x256
        GN(2, 4, 4, 2, 2, 4);  PL(2, 4, 4, 2, 2, 4);
        GN(1, 4, 4, 2, 2, 3);  PL(1, 4, 4, 2, 2, 3);
        GN(2, 2, 2, 4, 2, 3);  PL(2, 2, 2, 4, 2, 3);
        GN(2, 2, 4, 2, 2, 3);  PL(2, 2, 4, 2, 2, 3);
        GN(4, 1, 2, 4, 2, 3);  PL(4, 1, 2, 4, 2, 3);
        GB(2, 4, 4, 2, 2, 4, 1, 1, 0, 0);
        GB(4, 1, 2, 4, 2, 3, 1, 1, 0, 0);
        GB(2, 2, 4, 2, 2, 3, 1, 1, 0, 0);
        GB(1, 4, 4, 2, 2, 3, 1, 1, 0, 0);
        run_all(7, 10);
        return 0;
    }
    if (getenv("TUNE_DROPCOST")) {   // what do the Philox dropout draws cost in the training-forward epilogue?
        GNT(2, 4, 4, 2, 2, 4);
        g_drop_p = 0.1f;
        GNT(2, 4, 4, 2, 2, 4);
        g_cases.back().name += " +dropout";
        run_all(7, 10);
        return 0;
    }
    if (getenv("TUNE_RING")) {       // 2-slot loop vs ring pipeline, per epilogue and tiling
        PL(2, 4, 4, 2, 4, 1);  PL(2, 4, 4, 2, 2, 4);  PL(2, 4, 4, 2, 2, 3);
        PL(2, 2, 2, 2, 4, 1);  PL(2, 2, 2, 2, 2, 4);
        GN(2, 4, 4, 2, 4, 1);  GN(2, 4, 4, 2, 2, 4);
        GN(2, 2, 2, 2, 4, 1);  GN(2, 2, 2, 2, 2, 4);
        GNT(2, 4, 4, 2, 4, 1); GNT(2, 4, 4, 2, 2, 4);
        GNT(2, 2, 2, 2, 4, 1); GNT(2, 2, 2, 2, 2, 4);
        GB(2, 4, 4, 2, 4, 1, 1, 1, 0, 0); GB(2, 4, 4, 2, 2, 4, 1, 1, 0, 0);
        GB(2, 2, 2, 2, 4, 1, 1, 1, 0, 0); GB(2, 2, 2, 2, 2, 4, 1, 1, 0, 0);
        run_all(7, 10);
        return 0;
    }
    if (getenv("TUNE_PLAIN")) {
        PL(2, 2, 2, 2, 4, 1);
        PL(2, 4, 4, 2, 4, 1);
        PL(2, 2, 2, 2, 2, 1);
        GN(2, 2, 2, 2, 4, 1);
        GN(2, 4, 4, 2, 4, 1);
        run_all(7, 20);
        return 0;
    }
    if (getenv("TUNE_PROFILE")) {
        GB(2, 2, 2, 2, 4, 1, 1, 1, 0, 0);
        GB(2, 2, 2, 2, 4, 1, 0, 0, 0, 0);
        PL(2, 2, 2, 2, 4, 1);
        GNT(2, 4, 4, 2, 4, 1);
        GN(2, 4, 4, 2, 4, 1);
        PL(2, 4, 4, 2, 4, 1);
        run_all(1, 2);
        return 0;
    }
    // default table: the shipped tilings (256x256 / 8 waves, 128x128 / 4 waves) and the 4-wave two-workgroups-per-CU alternatives
    PL(2, 4, 4, 2, 4, 1);
    PL(2, 2, 2, 2, 4, 1);
    GN(2, 4, 4, 2, 4, 1);
    GN(2, 2, 2, 2, 4, 1);
    GN(2, 2, 2, 4, 2, 1);
    GN(4, 1, 2, 4, 2, 1);
    GNT(2, 4, 4, 2, 4, 1);
    GNT(2, 2, 2, 4, 2, 1);
    GNT(4, 1, 2, 4, 2, 1);
    GB(2, 2, 2, 2, 4, 1, 1, 1, 0, 0);
    GB(2, 2, 2, 2, 4, 1, 0, 1, 0, 0);
    GB(2, 2, 2, 2, 4, 1, 1, 0, 0, 0);
    GB(2, 2, 2, 2, 4, 1, 0, 0, 0, 0);
    GB(2, 2, 2, 2, 4, 1, 1, 1, 1, 1);
    GB(4, 2, 2, 2, 4, 1, 1, 1, 0, 0);
    run_all(7, 10);
    return 0;
}
