// Where does a 256x256 tile's time go, and when do the tiles run?  (round 5: the asm K loop issues an MFMA every 33 cycles of a SIMD --
// tools/kloop_fill_probe -- yet the plain GEMM reaches half of the matrix pipe's peak.)  Built with -DDPOSER_PHASE_STAMPS: every wave
// stamps s_memtime at tile entry / first stage landed / K loop done / epilogue's stores retired, s_memrealtime (100 MHz, chip-wide) at
// entry and exit, and its XCC / HW id.  Per kernel: HIP-event time, per-phase cycles (mean over waves), and from the real-time stamps the
// timeline: kernel span, busy time per CU (union of its workgroups' intervals), gaps between consecutive workgroups on a CU, tail.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DDPOSER_PHASE_STAMPS -I dposer_amd/csrc tools/tile_phase_probe.hip -o tools/bin/tile_phase_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "epilogues.h"
#include "gemm.h"

int dposer_set_error(int code, const std::string&) { return code; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

static void report(const char* name, int K, double us_med, const std::vector<uint64_t>& st, size_t n_wg, int nw, double flops) {
    // per wave: [rt0, rt1, prologue, loop, epilogue, ids]
    double pro = 0, loop = 0, epi = 0;
    uint64_t rt_min = ~0ull, rt_max = 0;
    for (size_t i = 0; i < n_wg * nw; ++i) {
        pro += (double)st[i * 6 + 2]; loop += (double)st[i * 6 + 3]; epi += (double)st[i * 6 + 4];
        rt_min = std::min(rt_min, st[i * 6 + 0]); rt_max = std::max(rt_max, st[i * 6 + 1]);
    }
    const double n = (double)(n_wg * nw);
    // workgroup interval = [min entry, max exit] over its waves; CU key = (xcc, se, sh, cu) of wave 0
    struct Iv { uint64_t a, b; };
    std::map<uint64_t, std::vector<Iv>> cu;
    for (size_t w = 0; w < n_wg; ++w) {
        uint64_t a = ~0ull, b = 0;
        for (int k = 0; k < nw; ++k) { a = std::min(a, st[(w * nw + k) * 6 + 0]); b = std::max(b, st[(w * nw + k) * 6 + 1]); }
        const uint64_t id = st[w * nw * 6 + 5];
        const uint64_t hw = id & 0xffffffffu, xcc = id >> 32;
        const uint64_t key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
        cu[key].push_back({a, b});
    }
    double busy = 0, gap_sum = 0, first_sum = 0, last_sum = 0, wg_len = 0;
    size_t gaps = 0, max_wg = 0, min_wg = 1u << 30;
    for (auto& kv : cu) {
        auto& v = kv.second;
        std::sort(v.begin(), v.end(), [](const Iv& x, const Iv& y) { return x.a < y.a; });
        max_wg = std::max(max_wg, v.size()); min_wg = std::min(min_wg, v.size());
        uint64_t end = v[0].a;
        for (size_t i = 0; i < v.size(); ++i) {
            wg_len += (double)(v[i].b - v[i].a);
            if (v[i].a > end) { gap_sum += (double)(v[i].a - end); ++gaps; busy += (double)(v[i].b - v[i].a); }
            else if (v[i].b > end) busy += (double)(v[i].b - end);
            end = std::max(end, v[i].b);
        }
        first_sum += (double)(v[0].a - rt_min);
        last_sum += (double)(rt_max - end);
    }
    const double tick_us = 0.01;      // s_memrealtime: 100 MHz
    const double span = (double)(rt_max - rt_min) * tick_us, ncu = (double)cu.size();
    printf("%-22s K=%4d : %7.1f us (events, median)  %5.0f TF | per wave cycles: prologue %7.0f  K loop %8.0f  epilogue %7.0f  (loop share %4.1f %%)\n", name, K, us_med,
           flops / us_med * 1e-6, pro / n, loop / n, epi / n, 100.0 * loop / (pro + loop + epi));
    printf("%-22s          timeline: span %6.1f us over %3.0f CUs, workgroups per CU %zu..%zu, mean workgroup %5.1f us; per CU: busy %6.1f us, first start +%4.1f us, idle between "
           "workgroups %5.1f us (%zu gaps), idle at the end %5.1f us\n", "", span, ncu, min_wg, max_wg, wg_len / (double)n_wg * tick_us, busy / ncu * tick_us, first_sum / ncu * tick_us,
           gap_sum / ncu * tick_us, gaps, last_sum / ncu * tick_us);
}

int main(int argc, char** argv) {
    const int64_t S = argc > 1 ? atoll(argv[1]) : 65536;
    const int C = 1024, KMAX = 1536;
    void *W, *X, *out, *xhat, *resid, *cout;
    float* part;
    uint64_t* stamps;
    float *bias, *gamma, *beta;
    GnAux* aux;
    const size_t n_wg = (size_t)(S / 256) * (C / 256);
    CK(hipMalloc(&W, (size_t)C * KMAX * 2)); CK(hipMalloc(&X, (size_t)S * KMAX * 2)); CK(hipMalloc(&out, (size_t)S * C * 2));
    CK(hipMalloc(&xhat, (size_t)S * C * 2)); CK(hipMalloc(&resid, (size_t)S * C * 2)); CK(hipMalloc(&aux, (size_t)S * 64 * sizeof(GnAux)));
    CK(hipMalloc(&stamps, n_wg * 8 * 6 * 8));
    CK(hipMalloc(&cout, (size_t)S * C * 2)); CK(hipMalloc(&part, (size_t)(S / 32) * 3 * C * 4));
    CK(hipMalloc(&bias, C * 4)); CK(hipMalloc(&gamma, C * 4)); CK(hipMalloc(&beta, C * 4));
    {
        std::vector<unsigned short> h((size_t)S * KMAX);
        srand(1);
        for (auto& v : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 0.2f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
        CK(hipMemcpy(X, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(W, h.data() + 12345, (size_t)C * KMAX * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(resid, h.data() + 777, (size_t)S * C * 2, hipMemcpyHostToDevice));
        std::vector<float> f(C, 1.0f);
        CK(hipMemcpy(gamma, f.data(), C * 4, hipMemcpyHostToDevice));
        CK(hipMemset(bias, 0, C * 4)); CK(hipMemset(beta, 0, C * 4));
    }
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int K : {1024, 1536}) {
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.W = W; g.w_stride_blocks = K / 16; g.src[0] = X; g.seg_kblocks[0] = K / 16; g.nseg = 1; g.ktot_blocks = K / 16;
        g.n_cblk = C / 256; g.n_sblk = (int)(S / 256); g.ksplit = 1;
        g.src[6] = stamps;
        PlainFTParams pp;
        pp.out = out; pp.N = C;
        GNParams pg;
        memset(&pg, 0, sizeof(pg));
        pg.bias = bias; pg.gamma = gamma; pg.beta = beta; pg.out = out; pg.H = C; pg.Spad = S; pg.resid = resid;
        GNParams pt = pg;
        pt.xhat = xhat; pt.aux = aux;
        pt.drop.p = 0.1f; pt.drop.scale = 1.f / 0.9f; pt.drop.thr = (uint32_t)(0.9 * 65536.0); pt.drop.groups_x4 = C / 8; pt.drop.seed = 7;
        struct Case { const char* name; std::function<void()> launch; };
        std::vector<Case> cases;
        cases.push_back({"plain store", [&] { CK((launch_gemm<__bf16, 2, 4, 4, 2, 2, EpiPlainFT<__bf16>, 4>(g, pp, 0))); }});
        cases.push_back({"EpiGN (inference)", [&] { CK((launch_gemm<__bf16, 2, 4, 4, 2, 2, EpiGN<__bf16, false>, 4>(g, pg, 0))); }});
        cases.push_back({"EpiGN<train>", [&] { CK((launch_gemm<__bf16, 2, 4, 4, 2, 2, EpiGN<__bf16, true>, 4>(g, pt, 0))); }});
        GNParams pt0 = pt;
        memset(&pt0.drop, 0, sizeof(pt0.drop));
        cases.push_back({"EpiGN<train> p = 0", [&] { CK((launch_gemm<__bf16, 2, 4, 4, 2, 2, EpiGN<__bf16, true>, 4>(g, pt0, 0))); }});
        GNBwdParams pb;
        memset(&pb, 0, sizeof(pb));
        pb.carry_in = resid; pb.carry_out = cout; pb.xhat = xhat; pb.aux = aux; pb.gamma = gamma; pb.beta = beta; pb.dy = out; pb.part = part;
        pb.H = C; pb.S_valid = S; pb.Spad = S; pb.drop_scale = 1.f / 0.9f;
        cases.push_back({"EpiGNBwd carry in+out", [&] { CK((launch_gemm<__bf16, 2, 4, 4, 2, 2, EpiGNBwd<__bf16>, 4>(g, pb, 0))); }});
        for (auto& c : cases) {
            printf("# %s K=%d\n", c.name, K); fflush(stdout);
            c.launch(); c.launch();
            CK(hipDeviceSynchronize());
            std::vector<double> us;
            for (int r = 0; r < 7; ++r) {
                CK(hipEventRecord(a, 0));
                c.launch();
                CK(hipEventRecord(b, 0));
                CK(hipEventSynchronize(b));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, a, b));
                us.push_back(ms * 1000.0);
            }
            std::sort(us.begin(), us.end());
            std::vector<uint64_t> hs(n_wg * 8 * 6);
            CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
            report(c.name, K, us[us.size() / 2], hs, n_wg, 8, 2.0 * S * C * K);
        }
    }
    return 0;
}
