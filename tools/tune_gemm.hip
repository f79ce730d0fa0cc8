// Tile-shape / staging tuner for the score-network GEMM (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idposer_amd/csrc tools/tune_gemm.hip -o gpurun_out/tune_gemm && gpurun_out/tune_gemm
// Times the forward GroupNorm layer GEMM (C = 1024 channels, K = 1024, S samples) for several workgroup tilings.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "epilogues.h"
#include "gemm.h"

int dposer_set_error(int code, const std::string&) { return code; }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int WC, int WS, int TC, int TS, int KB, int GLDS>
void run_plain(const char* name, int64_t S, int C, int K, void* W, void* X, void* out) {
    typedef GemmCfg<__bf16, WC, WS, TC, TS, KB> Cfg;
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.W = W; g.w_stride_blocks = K / 16; g.src[0] = X; g.seg_kblocks[0] = K / 16; g.nseg = 1; g.ktot_blocks = K / 16;
    g.n_cblk = C / (Cfg::CT * 32); g.n_sblk = (int)(S / (Cfg::ST * 32)); g.ksplit = 1;
    PlainFTParams p;
    p.out = out; p.N = C;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) CK((launch_gemm<__bf16, WC, WS, TC, TS, KB, EpiPlainFT<__bf16>, GLDS>(g, p, 0)));
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) CK((launch_gemm<__bf16, WC, WS, TC, TS, KB, EpiPlainFT<__bf16>, GLDS>(g, p, 0)));
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps;
    printf("%-34s tile %3dx%-3d waves %d KB %d glds %d PLAIN-STORE epilogue : %8.1f us  %7.1f TF\n", name, Cfg::CT * 32, Cfg::ST * 32, Cfg::NW, KB,
           (int)GLDS, us, 2.0 * S * C * K / (us * 1e-6) / 1e12);
}

template <int WC, int WS, int TC, int TS, int KB, int GLDS>
void run(const char* name, int64_t S, int C, int K, void* W, void* X, void* out, float* bias, float* gamma, float* beta, void* ref_out) {
    typedef GemmCfg<__bf16, WC, WS, TC, TS, KB> Cfg;
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.W = W; g.w_stride_blocks = K / 16; g.src[0] = X; g.seg_kblocks[0] = K / 16; g.nseg = 1; g.ktot_blocks = K / 16;
    g.n_cblk = C / (Cfg::CT * 32); g.n_sblk = (int)(S / (Cfg::ST * 32)); g.ksplit = 1;
    GNParams p;
    memset(&p, 0, sizeof(p));
    p.bias = bias; p.gamma = gamma; p.beta = beta; p.out = out; p.H = C;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) CK((launch_gemm<__bf16, WC, WS, TC, TS, KB, EpiGN<__bf16, false>, GLDS>(g, p, 0)));
    CK(hipDeviceSynchronize());
    const int reps = 20;
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) CK((launch_gemm<__bf16, WC, WS, TC, TS, KB, EpiGN<__bf16, false>, GLDS>(g, p, 0)));
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / reps;
    const double tf = 2.0 * S * C * K / (us * 1e-6) / 1e12;
    // correctness vs the first configuration's output
    std::vector<unsigned short> h0(1 << 16), h1(1 << 16);
    int bad = -1;
    if (ref_out != out) {
        CK(hipMemcpy(h0.data(), ref_out, h0.size() * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h1.data(), out, h1.size() * 2, hipMemcpyDeviceToHost));
        bad = 0;
        for (size_t i = 0; i < h0.size(); ++i) bad += (h0[i] != h1[i]);
    }
    printf("%-34s tile %3dx%-3d waves %d KB %d lds %3d KiB glds %d : %8.1f us  %7.1f TF  mismatches %d\n", name, Cfg::CT * 32, Cfg::ST * 32,
           Cfg::NW, KB, Cfg::LDS_BYTES / 1024, (int)GLDS, us, tf, bad);
}

int main(int argc, char** argv) {
    const int64_t S = argc > 1 ? atoll(argv[1]) : 65536;
    const int C = 1024, K = 1024;
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
#define RUN(WC, WS, TC, TS, KB, G, OUT) run<WC, WS, TC, TS, KB, G>(#WC "," #WS "," #TC "," #TS, S, C, K, W, X, OUT, bias, gamma, beta, o0)
#define RUNP(WC, WS, TC, TS, KB, G) run_plain<WC, WS, TC, TS, KB, G>(#WC "," #WS "," #TC "," #TS, S, C, K, W, X, o1)
    if (argc > 2) {   // profiling mode: only the two kernels of interest
        RUN(2, 4, 4, 2, 4, 1, o0);
        RUNP(2, 4, 4, 2, 4, 1);
        return 0;
    }
    RUNP(2, 4, 4, 2, 4, 1);
    RUNP(2, 4, 4, 2, 4, 1);
    RUNP(2, 2, 2, 2, 4, 1);
    return 0;
}
