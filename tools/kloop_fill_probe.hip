// Issue-slot probe of the SHIPPED 256x256 / 8-wave asm K-loop stage (csrc/gemm_kloop_asm.h): how many independent VALU instructions and
// tile stores fit into the gaps between its MFMAs before the loop slows down?  ("a probe in tools/ that runs the
// shipped asm K-loop stage with n = 0..6 independent VALU fillers and m = 0..2 global_store per MFMA gap and reports cycles / MFMA".)
//
// One binary per (n, m): the filler text is spliced into the stage statement through the DP_RS_Gi hooks (tools/gen_kloop_fill.py writes
// them; the library build defines them empty).  Build + run all of them: tools/kloop_fill_probe.sh.  Each binary times the plain-store
// 256x256 GEMM (65536 samples x 1024 channels) at K = 1024 / 1536 / 2048 and prints, per K:
//   us per launch (HIP events, median of the rounds), and from s_memtime stamps around the asm stages of every wave:
//   cycles per MFMA of one wave (= K-loop cycles / MFMAs the wave issued); two waves share a SIMD's matrix pipe, so the pipe-bound
//   floor is 64 cycles per own MFMA (2 x 32) and 8 issue slots of ~4 cycles per MFMA of the PAIR.
// Output is garbage where filler stores land (they go to their own scratch) -- timing only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "epilogues.h"
#include "gemm.h"

int dposer_set_error(int code, const std::string&) { return code; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

#ifndef DP_RS_FILL_N
#define DP_RS_FILL_N 0
#define DP_RS_FILL_M 0
#endif
#ifndef FILL_TAG
#define FILL_TAG "mix"
#endif

int main(int argc, char** argv) {
    const int64_t S = 65536;
    const int C = 1024, KMAX = 2048;
    const int rounds = argc > 1 ? atoi(argv[1]) : 5;
    void *W, *X, *out, *scratch;
    uint64_t* stamps;
    const size_t n_waves = (size_t)(S / 256) * (C / 256) * 8;
    CK(hipMalloc(&W, (size_t)C * KMAX * 2)); CK(hipMalloc(&X, (size_t)S * KMAX * 2)); CK(hipMalloc(&out, (size_t)S * C * 2));
    CK(hipMalloc(&scratch, (size_t)(KMAX / 32 + 2) * n_waves * 4096)); CK(hipMalloc(&stamps, n_waves * 8));
    {
        std::vector<unsigned short> h((size_t)S * KMAX);
        srand(1);
        for (auto& v : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 0.2f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
        CK(hipMemcpy(X, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(W, h.data() + 12345, (size_t)C * KMAX * 2, hipMemcpyHostToDevice));
    }
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int K : {1024, 1536, 2048}) {
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.W = W; g.w_stride_blocks = K / 16; g.src[0] = X; g.seg_kblocks[0] = K / 16; g.nseg = 1; g.ktot_blocks = K / 16;
        g.n_cblk = C / 256; g.n_sblk = (int)(S / 256); g.ksplit = 1;
        g.src[6] = stamps; g.src[7] = scratch;
        PlainFTParams p;
        p.out = out; p.N = C;
        auto launch = [&] { CK((launch_gemm<__bf16, 2, 4, 4, 2, 2, EpiPlainFT<__bf16>, 4>(g, p, 0))); };
        launch(); launch();
        CK(hipDeviceSynchronize());
        std::vector<double> us;
        for (int r = 0; r < rounds; ++r) {
            CK(hipEventRecord(a, 0));
            for (int i = 0; i < 10; ++i) launch();
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            us.push_back(ms * 100.0);
        }
        std::sort(us.begin(), us.end());
        std::vector<uint64_t> hs(n_waves);
        CK(hipMemcpy(hs.data(), stamps, n_waves * 8, hipMemcpyDeviceToHost));
        double sum = 0, mx = 0;
        for (auto v : hs) { sum += (double)v; mx = std::max(mx, (double)v); }
        // s_memtime ticks are shader cycles on gfx950 (MI355X_MICROARCH.md, cycle-constant table); the stamps bracket the asm stages, which
        // hold every MFMA of the wave: K / 16 k-blocks x 8 MFMAs
        const double mfma_per_wave = K / 16 * 8.0;
        const double ticks = sum / n_waves;
        printf("fill n=%d m=%d %-5s K=%4d : %7.1f us median (%7.1f min)  %6.0f TF | K-loop %9.0f cycles avg %9.0f max per wave = %6.2f cycles per own MFMA (pipe floor 64: two waves per SIMD)\n",
               DP_RS_FILL_N, DP_RS_FILL_M, FILL_TAG, K, us[us.size() / 2], us.front(), 2.0 * S * C * K / us[us.size() / 2] * 1e-6, ticks, mx, ticks / mfma_per_wave);
    }
    return 0;
}
