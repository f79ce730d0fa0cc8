// What would ONE launch of all 256x256 weight-gradient tiles of a training step cost, against the ten split-K launches?
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dposer_amd/csrc tools/wgrad_batch_probe.hip -o tools/bin/wgrad_batch_probe
// A: dW [1024][1024] over 65536 samples, split 16 (256 workgroups x 128 stages)        -- x 4 per step
// B: dW [1024][512]  over 65536 samples, split 32 (256 workgroups x 64 stages)         -- x 5 per step
// C: the same 893 GFLOP as 256 workgroups x 832 stages in one launch (emulated as dW [1024][16384] over 26624 samples, no split)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "epilogues.h"
#include "gemm.h"
#include "gemm_wgrad_tr.h"
int dposer_set_error(int code, const std::string&) { return code; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
int main() {
    const size_t nY = (size_t)65536 * 1024, nH = (size_t)26624 * 16384;
    std::vector<unsigned short> h(nH);
    unsigned s = 12345;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3c00 + ((s >> 16) & 0x3ff)) | ((s >> 3) & 0x8000); }
    void *dy, *hb;
    float* slab;
    CK(hipMalloc(&dy, nY * 2)); CK(hipMalloc(&hb, nH * 2)); CK(hipMalloc(&slab, (size_t)32 * 1024 * 1024 * 4));
    CK(hipMemcpy(dy, h.data(), nY * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(hb, h.data(), nH * 2, hipMemcpyHostToDevice));
    auto mk = [&](int Kc, int S, int ks, WgradTrArgs& g, WgradParams& wp) {
        memset(&g, 0, sizeof(g));
        g.dY = dy; g.H = hb; g.N = 1024; g.Kc = Kc; g.n_cblk = 4; g.n_sblk = Kc / 256; g.sblocks = S / 32; g.ksplit = ks;
        wp.slab = slab; wp.slab_stride = (int64_t)1024 * (Kc > 1024 ? 1024 : Kc); wp.ld = Kc; wp.N_valid = 1024; wp.K_valid = Kc;
    };
    WgradTrArgs gA, gB, gC; WgradParams pA, pB, pC;
    mk(1024, 65536, 16, gA, pA); mk(512, 65536, 32, gB, pB); mk(16384, 26624, 1, gC, pC);
    pC.slab_stride = 0;                               // (C writes 64 MB once: its 256 tiles)
    CK(hipFree(slab)); CK(hipMalloc(&slab, (size_t)1024 * 16384 * 4)); pA.slab = pB.slab = pC.slab = slab;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto fn, double flops) {
        for (int i = 0; i < 3; ++i) fn();
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int r = 0; r < 7; ++r) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 10; ++i) fn();
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms / 10 < best ? ms / 10 : best;
        }
        printf("%-70s %8.1f us  %6.0f TFLOP/s\n", name, best * 1e3, flops / best * 1e-9);
    };
    for (int rep = 0; rep < 2; ++rep) {
        timeit("A  [1024x1024] split 16: 256 WG x 128 stages", [&] { CK((launch_wgrad_tr<2, 4, 4, 2, 4>(gA, pA, 0))); }, 2.0 * 65536 * 1024 * 1024);
        timeit("B  [1024x512]  split 32: 256 WG x 64 stages", [&] { CK((launch_wgrad_tr<2, 4, 4, 2, 4>(gB, pB, 0))); }, 2.0 * 65536 * 1024 * 512);
        timeit("4 A + 5 B back to back (one training step's 256x256 wgrad launches)", [&] {
            for (int i = 0; i < 4; ++i) CK((launch_wgrad_tr<2, 4, 4, 2, 4>(gA, pA, 0)));
            for (int i = 0; i < 5; ++i) CK((launch_wgrad_tr<2, 4, 4, 2, 4>(gB, pB, 0))); }, 2.0 * 65536 * 1024 * (4 * 1024 + 5 * 512));
        timeit("C  the same FLOPs as ONE launch: 256 WG x 832 stages", [&] { CK((launch_wgrad_tr<2, 4, 4, 2, 4>(gC, pC, 0))); }, 2.0 * 26624 * 1024 * 16384);
    }
    return 0;
}
