// Standalone timing harness for the sampler's producer / consumer kernel (tools/experimental/em_l0_fused.h) on random operands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I tools [-DPIPE_...] tools/em_l0_pipe_bench.hip -o tools/bin/em_l0_pipe_bench
#include "experimental/em_l0_fused.h"
#include <cstdio>
#include <vector>
int main() {
    const int64_t S = 65536;
    const int H = 1024;
    void *h, *wpost, *w0, *h0;
    float *xft, *bias, *par, *sig;
    hipMalloc(&h, S * H * 2); hipMalloc(&wpost, 64 * H * 2); hipMalloc(&w0, (size_t)H * 576 * 2); hipMalloc(&h0, S * H * 2);
    hipMalloc(&xft, S * 64 * 4); hipMalloc(&bias, 64 * 4); hipMalloc(&par, 3 * H * 4); hipMalloc(&sig, 1000 * 4);
    std::vector<unsigned short> r(S * H);
    for (size_t i = 0; i < r.size(); ++i) r[i] = 0x3c00 + (unsigned short)((i * 2654435761u) >> 24);   // bf16 around 0.01
    hipMemcpy(h, r.data(), S * H * 2, hipMemcpyHostToDevice);
    hipMemcpy(wpost, r.data(), 64 * H * 2, hipMemcpyHostToDevice);
    hipMemcpy(w0, r.data(), (size_t)H * 576 * 2, hipMemcpyHostToDevice);
    hipMemset(xft, 0, S * 64 * 4); hipMemset(bias, 0, 64 * 4);
    std::vector<float> ones(3 * H, 1.0f);
    hipMemcpy(par, ones.data(), 3 * H * 4, hipMemcpyHostToDevice);
    hipMemcpy(sig, ones.data(), 1000 * 4, hipMemcpyHostToDevice);
    EmL0PipeArgs a = {};
    a.wpost = wpost; a.h = h; a.n_chunks = (int)(S / 128);
    a.p.w0 = w0; a.p.w0_stride_blocks = 36; a.p.bias0 = par; a.p.gamma0 = par + H; a.p.beta0 = par + 2 * H; a.p.h0 = h0; a.p.H = H;
    EmStepParams& p = a.p.em;
    p.bias = bias; p.x_ft = xft; p.x_mean_ft = nullptr; p.sigmas = sig; p.t = 0.5f; p.num_scales = 1000; p.scale_by_sigma = 0;
    p.D = 63; p.Cp = 64; p.QD = 16; p.S_valid = S; p.seed = 1; p.step = 3;
    SdeCfg sc; sc.kind = SDE_SUBVP; sc.beta_0 = 0.1f; sc.beta_1 = 20.f; sc.N = 1000; sc.T = 1.f;
    p.sde = make_sde_dev(sc);
    for (int rep = 0; rep < 3; ++rep) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        launch_em_l0_pipe(a, 0);
        hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) launch_em_l0_pipe(a, 0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("em_l0_pipe_kernel: %.1f us / launch (%s)\n", ms * 100.f, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
