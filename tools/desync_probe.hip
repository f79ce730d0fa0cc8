// Does the GEMM epilogue phase cost chip-wide HBM bandwidth (every CU stores at the same time) or per-CU serial work?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idposer_amd/csrc -Iinclude -Itools tools/desync_probe.hip -o tools/bin/desync_probe
// The training GEMMs run as 4 rounds of 256 equal tiles: all CUs are in their K loop (MFMA-bound, ~1 TB/s of HBM) and then all
// in their epilogue (HBM-bound) at the same time.  This probe runs the same launches as TWO half-batch chains on two streams
// that own complementary halves of the CUs (hipExtStreamCreateWithCUMask), optionally phase-shifted by a spin kernel, and
// compares chain time against the full-chip launches.  If the epilogue is a chip-wide bandwidth burst, the phase-shifted pair
// is faster; if it is per-CU work, nothing changes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "epilogues.h"
#include "gemm.h"
#include "gemm_wgrad_tr.h"

int dposer_set_error(int code, const std::string&) { return code; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_spin(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
// one record per workgroup: XCC id, HW_ID register
__global__ void __launch_bounds__(256) k_census(unsigned* out, long long ticks) {
    extern __shared__ unsigned char big[];
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hwid;
        big[0] = 1;
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

struct Bufs {
    int64_t S; int C, K;
    void *W, *X, *out, *xhat, *dy, *cin;
    GnAux* aux; float *bias, *gamma, *beta, *part, *slab;
    void *wy, *wh;   // wgrad operands [S][1024], [S][1024]
};
static Bufs B;

// half = -1: whole batch; 0 / 1: that half of the sample blocks (FT layouts are sample-block-major: a half is a pointer offset)
static void launch_gnt(int half, int K, hipStream_t st) {
    const int64_t S = half < 0 ? B.S : B.S / 2, s0 = half <= 0 ? 0 : B.S / 2;
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.W = B.W; g.w_stride_blocks = K / 16; g.src[0] = (char*)B.X + s0 * K * 2; g.seg_kblocks[0] = K / 16; g.nseg = 1; g.ktot_blocks = K / 16;
    g.n_cblk = B.C / 256; g.n_sblk = (int)(S / 256); g.ksplit = 1;
    GNParams p; memset(&p, 0, sizeof(p));
    p.bias = B.bias; p.gamma = B.gamma; p.beta = B.beta; p.out = (char*)B.out + s0 * B.C * 2; p.H = B.C; p.Spad = S;
    p.xhat = (char*)B.xhat + s0 * B.C * 2; p.aux = B.aux + (s0 / 32) * (B.C / 32) * 64;
    p.drop.p = 0.1f; p.drop.scale = 1.f / 0.9f; p.drop.thr = (uint32_t)(0.9 * 65536.0); p.drop.groups_x4 = B.C / 8; p.drop.seed = 7;
    CK((launch_gemm<__bf16, 2, 4, 4, 2, 2, EpiGN<__bf16, true>, 4>(g, p, st)));
}
static void launch_gnb(int half, hipStream_t st) {
    const int K = 1024;
    const int64_t S = half < 0 ? B.S : B.S / 2, s0 = half <= 0 ? 0 : B.S / 2;
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.W = B.W; g.w_stride_blocks = K / 16; g.src[0] = (char*)B.X + s0 * K * 2; g.seg_kblocks[0] = K / 16; g.nseg = 1; g.ktot_blocks = K / 16;
    g.n_cblk = B.C / 256; g.n_sblk = (int)(S / 256); g.ksplit = 1;
    GNBwdParams p; memset(&p, 0, sizeof(p));
    p.xhat = (char*)B.xhat + s0 * B.C * 2; p.aux = B.aux + (s0 / 32) * (B.C / 32) * 64; p.gamma = B.gamma; p.beta = B.beta;
    p.dy = (char*)B.dy + s0 * B.C * 2; p.part = B.part + (s0 / 64) * 3 * B.C; p.H = B.C; p.S_valid = S; p.Spad = S; p.drop_scale = 1.f / 0.9f;
    CK((launch_gemm<__bf16, 2, 4, 4, 2, 2, EpiGNBwd<__bf16>, 4>(g, p, st)));
}
static void launch_plain(int half, int K, hipStream_t st) {
    const int64_t S = half < 0 ? B.S : B.S / 2, s0 = half <= 0 ? 0 : B.S / 2;
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.W = B.W; g.w_stride_blocks = K / 16; g.src[0] = (char*)B.X + s0 * K * 2; g.seg_kblocks[0] = K / 16; g.nseg = 1; g.ktot_blocks = K / 16;
    g.n_cblk = B.C / 256; g.n_sblk = (int)(S / 256); g.ksplit = 1;
    PlainFTParams p; p.out = (char*)B.out + s0 * B.C * 2; p.N = B.C;
    CK((launch_gemm<__bf16, 2, 4, 4, 2, 2, EpiPlainFT<__bf16>, 4>(g, p, st)));
}
// dW [1024][Kc] over the (half) batch; whole batch: ksplit 16, a half: ksplit 8 into its own 8 slabs
static void launch_wg(int half, int Kc, hipStream_t st) {
    const int64_t S = half < 0 ? B.S : B.S / 2, s0 = half <= 0 ? 0 : B.S / 2;
    const int ks = half < 0 ? 16 : 8;
    WgradTrArgs g; memset(&g, 0, sizeof(g));
    g.dY = (char*)B.wy + s0 * 1024 * 2; g.H = (char*)B.wh + s0 * Kc * 2; g.N = 1024; g.Kc = Kc; g.n_cblk = 4; g.n_sblk = Kc / 256; g.sblocks = (int)(S / 32); g.ksplit = ks;
    WgradParams p; p.slab = B.slab + (half > 0 ? (int64_t)8 * 1024 * Kc : 0); p.slab_stride = (int64_t)1024 * Kc; p.ld = Kc; p.N_valid = 1024; p.K_valid = Kc;
    CK((launch_wgrad_tr<2, 4, 4, 2, 4>(g, p, st)));
}

typedef std::function<void(int, hipStream_t)> Chain;   // (half, stream) -> launches one chain

static hipStream_t sA, sB, uA, uB;
static hipEvent_t e0, e1, eA, eB;

static double time_single(const Chain& c) {
    CK(hipEventRecord(e0, 0));
    c(-1, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3;
}
static double time_dual(const Chain& c, hipStream_t a, hipStream_t b, double offset_us, bool only_a = false) {
    CK(hipEventRecord(e0, a));
    if (!only_a) {
        CK(hipStreamWaitEvent(b, e0, 0));
        if (offset_us > 0) hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, b, (long long)(offset_us * 100));
    }
    c(0, a);
    if (!only_a) {
        c(1, b);
        CK(hipEventRecord(eB, b));
        CK(hipStreamWaitEvent(a, eB, 0));
    }
    CK(hipEventRecord(e1, a));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3;
}

int main(int argc, char** argv) {
    const int64_t S = argc > 1 ? atoll(argv[1]) : 65536;
    const int C = 1024, KM = 1536;
    B.S = S; B.C = C; B.K = KM;
    CK(hipMalloc(&B.W, (size_t)C * KM * 2)); CK(hipMalloc(&B.X, (size_t)S * KM * 2)); CK(hipMalloc(&B.out, (size_t)S * C * 2));
    CK(hipMalloc(&B.xhat, (size_t)S * C * 2)); CK(hipMalloc(&B.dy, (size_t)S * C * 2));
    CK(hipMalloc(&B.aux, (size_t)(S / 32) * (C / 32) * 64 * sizeof(GnAux))); CK(hipMalloc(&B.part, (size_t)(S / 32) * 3 * C * 4));
    CK(hipMalloc(&B.bias, C * 4)); CK(hipMalloc(&B.gamma, C * 4)); CK(hipMalloc(&B.beta, C * 4));
    CK(hipMalloc(&B.wy, (size_t)S * 1024 * 2)); CK(hipMalloc(&B.wh, (size_t)S * 1024 * 2)); CK(hipMalloc(&B.slab, (size_t)16 * 1024 * 1024 * 4));
    {
        std::vector<unsigned short> h((size_t)S * KM);
        srand(1);
        auto rnd = [] { float f = (rand() / (float)RAND_MAX - 0.5f) * 0.2f; unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); };
        for (auto& v : h) v = rnd();
        CK(hipMemcpy(B.X, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(B.W, h.data(), (size_t)C * KM * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(B.wy, h.data(), (size_t)S * 1024 * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(B.wh, h.data() + 12345, (size_t)S * 1024 * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(B.xhat, h.data() + 777, (size_t)S * C * 2, hipMemcpyHostToDevice));
        std::vector<float> hb(C, 0.01f), hg(C, 1.0f);
        CK(hipMemcpy(B.bias, hb.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B.gamma, hg.data(), C * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(B.beta, hb.data(), C * 4, hipMemcpyHostToDevice));
        CK(hipMemset(B.aux, 0x3f, (size_t)(S / 32) * (C / 32) * 64 * sizeof(GnAux)));
    }
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device: %s, %d CUs\n", prop.name, ncu);
    // masks: MODE 0 = even / odd bits, 1 = low / high half of the bit range, 2 = pairs (bits 0,1 -> A; 2,3 -> B), 3 = alternate groups of 8
    const int mode = getenv("MASK_MODE") ? atoi(getenv("MASK_MODE")) : 0;
    uint32_t mA[8] = {}, mB[8] = {};
    for (int i = 0; i < ncu; ++i) {
        bool a;
        if (mode == 0) a = (i & 1) == 0;
        else if (mode == 1) a = i < ncu / 2;
        else if (mode == 2) a = ((i >> 1) & 1) == 0;
        else a = ((i >> 3) & 1) == 0;
        (a ? mA : mB)[i >> 5] |= 1u << (i & 31);
    }
    CK(hipExtStreamCreateWithCUMask(&sA, 8, mA)); CK(hipExtStreamCreateWithCUMask(&sB, 8, mB));
    CK(hipStreamCreateWithFlags(&uA, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&uB, hipStreamNonBlocking));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&eA)); CK(hipEventCreate(&eB));

    // ---- census: where do the workgroups of a masked stream land? ---------------------------------------------------
    {
        unsigned* d; CK(hipMalloc(&d, 2 * 1024 * 4));
        CK(hipFuncSetAttribute((const void*)k_census, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
        hipStream_t ss[3] = {0, sA, sB};
        const char* nm[3] = {"unmasked", "mask A", "mask B"};
        for (int k = 0; k < 3; ++k) {
            CK(hipMemset(d, 0xff, 2 * 1024 * 4));
            hipLaunchKernelGGL(k_census, dim3(256), dim3(256), 100 * 1024, ss[k], d, 2000LL);   // 100 KB LDS: one workgroup per CU; 20 us
            CK(hipDeviceSynchronize());
            std::vector<unsigned> h(512); CK(hipMemcpy(h.data(), d, 512 * 4, hipMemcpyDeviceToHost));
            int per_xcc[8] = {}; std::vector<int> seen;
            int mism = 0;
            for (int b = 0; b < 256; ++b) {
                const unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
                per_xcc[xcc & 7]++;
                const int key = (xcc << 16) | (hw & 0xff00);   // cu_id[11:8], sh_id[12], se_id[15:13]
                if (std::find(seen.begin(), seen.end(), key) == seen.end()) seen.push_back(key);
                if ((int)(xcc & 7) != (b & 7)) ++mism;
            }
            printf("census %-9s: distinct (xcc, se, sh, cu) = %3zu; per XCC:", nm[k], seen.size());
            for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
            printf("; blocks with xcc != b %% 8: %d\n", mism);
        }
        CK(hipFree(d));
    }

    struct Item { std::string name; Chain chain; double flops; };
    std::vector<Item> items;
    const int NREP = 6;
    items.push_back({"6 x GN-train fwd K=1536", [&](int h, hipStream_t s) { for (int i = 0; i < NREP; ++i) launch_gnt(h, 1536, s); }, NREP * 2.0 * S * C * 1536});
    items.push_back({"6 x GN-bwd dgrad K=1024", [&](int h, hipStream_t s) { for (int i = 0; i < NREP; ++i) launch_gnb(h, s); }, NREP * 2.0 * S * C * 1024});
    items.push_back({"6 x plain K=1536", [&](int h, hipStream_t s) { for (int i = 0; i < NREP; ++i) launch_plain(h, 1536, s); }, NREP * 2.0 * S * C * 1536});
    items.push_back({"6 x wgrad 1024x1024", [&](int h, hipStream_t s) { for (int i = 0; i < NREP; ++i) launch_wg(h, 1024, s); }, NREP * 2.0 * S * 1024 * 1024});
    items.push_back({"mix: 2 fwd, 2 x (bwd, wgrad, wgrad512)", [&](int h, hipStream_t s) {
                         launch_gnt(h, 1536, s); launch_gnt(h, 1536, s);
                         for (int i = 0; i < 2; ++i) { launch_gnb(h, s); launch_wg(h, 1024, s); launch_wg(h, 512, s); } },
                     2 * 2.0 * S * C * 1536 + 2 * (2.0 * S * C * 1024 * 2 + 2.0 * S * C * 512)});

    const int ROUNDS = 5;
    const double offs[] = {0, 15, 30, 45};
    for (auto& it : items) {
        // warm-up of every variant
        time_single(it.chain); time_dual(it.chain, sA, sB, 0); time_dual(it.chain, uA, uB, 0); time_dual(it.chain, sA, sB, 0, true);
        std::vector<double> t_single, t_alone, t_un, t_m[4];
        for (int r = 0; r < ROUNDS; ++r) {
            t_single.push_back(time_single(it.chain));
            t_alone.push_back(time_dual(it.chain, sA, sB, 0, true));
            t_un.push_back(time_dual(it.chain, uA, uB, 0));
            for (int k = 0; k < 4; ++k) t_m[k].push_back(time_dual(it.chain, sA, sB, offs[k]));
        }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        auto mn = [](std::vector<double> v) { return *std::min_element(v.begin(), v.end()); };
        printf("\n%s  (%.0f GFLOP)\n", it.name.c_str(), it.flops * 1e-9);
        printf("  full chip, one stream                : median %8.1f us  min %8.1f  (%5.0f TF)\n", med(t_single), mn(t_single), it.flops / med(t_single) * 1e-6);
        printf("  half batch ALONE on 128 masked CUs   : median %8.1f us  min %8.1f   (x2 = %.1f)\n", med(t_alone), mn(t_alone), 2 * med(t_alone));
        printf("  two halves, two unmasked streams     : median %8.1f us  min %8.1f\n", med(t_un), mn(t_un));
        for (int k = 0; k < 4; ++k)
            printf("  two halves, masked, offset %4.0f us    : median %8.1f us  min %8.1f  (minus offset: %8.1f)\n", offs[k], med(t_m[k]), mn(t_m[k]), med(t_m[k]) - offs[k]);
    }
    return 0;
}
