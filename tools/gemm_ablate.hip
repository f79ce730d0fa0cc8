// Ablation of the 256x256 / 8-wave GEMM main loop (run on the GPU box): where do the cycles between the shipped
// ~1.1 PFLOP/s and the 2.1 PFLOP/s register-only MFMA rate go?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idposer_amd/csrc -Iinclude tools/gemm_ablate.hip -o tools/bin/gemm_ablate
// The kernel below is the product main loop (csrc/gemm.h) with pieces switched off by template flags; results of the
// ablated variants are garbage by construction -- only the time matters.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <type_traits>
#include <string>
#include <vector>

#include "epilogues.h"
#include "gemm.h"

int dposer_set_error(int code, const std::string&) { return code; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

enum : int {
    NO_GLDS = 1,     // no DMA inside the loop (prologue only)
    NO_BAR = 2,      // no stage barrier / vmcnt wait
    NO_FRAG = 4,     // no ds_read inside the loop (fragments loaded once)
    SAME_ADDR = 8,   // DMA re-fetches k-blocks 0..KB-1 every stage (L2-resident operands, same issue pattern)
    NO_EPI = 16,     // epilogue stores only when an impossible condition holds
    W_ONLY = 32,     // DMA fetches only the weight half of a stage
    X_ONLY = 64,     // DMA fetches only the activation half of a stage
    TIMING = 256,    // s_memtime stamps around the vmcnt wait, the barrier and the DMA issue (written to a debug buffer)
    CLK = 1024,      // ring: s_memtime around the whole K loop (mean cycles per tile printed) -- separates stall cycles from clock drops
    HALF_LOAD = 512, // ring: waves 0..NW/2-1 (one per SIMD) issue all DMA pieces, their SIMD partners only compute
    FINE = 128,      // DMA / ds_read instructions interleaved one-by-one with the MFMAs of the group they precede
};

__device__ unsigned long long g_timing[4096 * 8 * 4];

template <int WC, int WS, int TC, int TS, int KB, int ABL, int DLY = 0, int PRIO = 0>
__global__ void __launch_bounds__(WC* WS * 64, 1) abl_kernel(GemmArgs g, PlainFTParams ep) {
    typedef __bf16 T;
    typedef GemmCfg<T, WC, WS, TC, TS, KB> C;
    typedef typename Mma<T>::Frag Frag;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wc = wave / WS, ws = wave % WS;
    const int ntiles = g.n_cblk * g.n_sblk;
    const int L = xcd_remap(blockIdx.x, ntiles);
    const int cblk = L % g.n_cblk;
    const int sblk = L / g.n_cblk;

    f32x16 acc[TC][TS];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TS; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    int seg_kb = 0, w_kb = 0;
    const int seg_total = g.seg_kblocks[0];
    const int nstages = g.ktot_blocks / KB;
    const unsigned char* sbase = reinterpret_cast<const unsigned char*>(g.src[0]);

    auto fetch_glds = [&](int buf) __attribute__((always_inline)) {
        if constexpr (!(ABL & X_ONLY)) {
#pragma unroll
            for (int i = 0; i < C::LPW_A; ++i) {
                const int blk = wave + i * C::NW;
                const int rb = blk / KB, kb = blk % KB;
                const unsigned char* p = reinterpret_cast<const unsigned char*>(g.W) + (((int64_t)(cblk * C::CT + rb) * g.w_stride_blocks + w_kb + kb) << 10);
                __builtin_amdgcn_global_load_lds((gptr_t)(p + lane * 16), (lptr_t)(smem + buf * C::STAGE_BYTES + (blk << 10)), 16, 0, 0);
            }
        }
        if constexpr (!(ABL & W_ONLY)) {
#pragma unroll
            for (int i = 0; i < C::LPW_B; ++i) {
                const int blk = wave + i * C::NW;
                const int rb = blk / KB, kb = blk % KB;
                const unsigned char* p = sbase + (((int64_t)(sblk * C::ST + rb) * seg_total + seg_kb + kb) << 10);
                __builtin_amdgcn_global_load_lds((gptr_t)(p + lane * 16), (lptr_t)(smem + buf * C::STAGE_BYTES + ((C::CT * KB + blk) << 10)), 16, 0, 0);
            }
        }
        if constexpr (!(ABL & SAME_ADDR)) {
            seg_kb += KB;
            w_kb += KB;
        }
    };
    Frag fa[2][TC], fb[2][TS];
    auto load_frags = [&](int buf, int kb, int set) __attribute__((always_inline)) {
        const unsigned char* a_base = smem + buf * C::STAGE_BYTES + ((wc * TC * KB) << 10) + lane * 16;
        const unsigned char* b_base = smem + buf * C::STAGE_BYTES + ((C::CT * KB + ws * TS * KB) << 10) + lane * 16;
#pragma unroll
        for (int i = 0; i < TC; ++i) fa[set][i] = *reinterpret_cast<const Frag*>(a_base + ((i * KB + kb) << 10));
#pragma unroll
        for (int j = 0; j < TS; ++j) fb[set][j] = *reinterpret_cast<const Frag*>(b_base + ((j * KB + kb) << 10));
    };
    auto mma = [&](int set) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TS; ++j) Mma<T>::run(fa[set][i], fb[set][j], acc[i][j]);
    };

    if constexpr (PRIO == 1) { if (wave >= C::NW / 2) __builtin_amdgcn_s_setprio(1); }
    if constexpr (PRIO == 2) { if (wave < C::NW / 2) __builtin_amdgcn_s_setprio(1); }
    fetch_glds(0);
    if (nstages > 1) fetch_glds(1);
    __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
    __syncthreads_lds_only();
    load_frags(0, 0, 0);
    if constexpr (ABL & NO_FRAG) load_frags(0, 1, 1);
    uint64_t tm_wait = 0, tm_bar = 0, tm_dma = 0;
    // one stage; HAS_NEXT: a following stage exists (barrier + first fragments of it), HAS_DMA: refill this stage's buffer
    auto stage = [&](int t, auto has_next, auto has_dma) __attribute__((always_inline)) {
        constexpr bool HAS_NEXT = decltype(has_next)::value, HAS_DMA = decltype(has_dma)::value;
        const int buf = t & 1;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            if (kb + 1 < KB) {
                if constexpr (!(ABL & NO_FRAG)) load_frags(buf, kb + 1, (kb + 1) & 1);
            } else if constexpr (HAS_NEXT) {
                if constexpr (ABL & TIMING) {
                    __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(15));   // own ds_reads back: what remains is DMA latency
                    const uint64_t t0 = __builtin_amdgcn_s_memtime();
                    __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(0));
                    const uint64_t t1 = __builtin_amdgcn_s_memtime();
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    const uint64_t t2 = __builtin_amdgcn_s_memtime();
                    if constexpr (HAS_DMA) fetch_glds(buf);
                    const uint64_t t3 = __builtin_amdgcn_s_memtime();
                    tm_wait += t1 - t0; tm_bar += t2 - t1; tm_dma += t3 - t2;
                } else {
                if constexpr (!(ABL & NO_BAR)) {
                    __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(0));
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
                if constexpr (!(ABL & NO_GLDS) && HAS_DMA) {
                    if (DLY == 0 || wave < C::NW / 2) fetch_glds(buf);
                }
                }
                if constexpr (!(ABL & NO_FRAG)) load_frags(buf ^ 1, 0, 0);
            }
            if constexpr (!(ABL & FINE)) __builtin_amdgcn_sched_barrier(0);
            mma(kb & 1);
            if constexpr (ABL & FINE) {
                if (kb + 1 < KB || !HAS_DMA || (ABL & NO_GLDS)) {
#pragma unroll
                    for (int i = 0; i < TC * TS; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (i < TC + TS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < TC * TS; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x010, (C::LPW + TC * TS - 1) / (TC * TS), 0);
                        if (i < TC + TS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DLY == 1) {
                if (HAS_DMA && kb == KB - 1 && wave >= C::NW / 2) fetch_glds(buf);
            } else if constexpr (DLY >= 2) {
                if (HAS_NEXT && kb == DLY - 2 && wave >= C::NW / 2 && t >= 1) fetch_glds(buf ^ 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const uint64_t tm_begin = (ABL & TIMING) ? __builtin_amdgcn_s_memtime() : 0;
    {
        int t = 0;
        for (; t + 2 < nstages; ++t) stage(t, std::true_type{}, std::true_type{});
        if (t + 1 < nstages) { stage(t, std::true_type{}, std::false_type{}); ++t; }
        stage(t, std::false_type{}, std::false_type{});
    }
    if constexpr (ABL & TIMING) {
        const uint64_t tm_total = __builtin_amdgcn_s_memtime() - tm_begin;
        if (lane == 0) {
            unsigned long long* d = g_timing + ((size_t)blockIdx.x * C::NW + wave) * 4;
            d[0] = tm_total; d[1] = tm_wait; d[2] = tm_bar; d[3] = tm_dma;
        }
    }
    if constexpr (ABL & NO_EPI) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TS; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 123.456f) reinterpret_cast<float*>(ep.out)[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        EpiPlainFT<T>::template apply<TC, TS>(ep, acc, (cblk * C::CT + wc * TC) * 32, ((int64_t)sblk * C::ST + ws * TS) * 32, lane, sblk * WS + ws, 0, nullptr, 0, nullptr);
    }
}

struct Case { std::string name; std::function<void()> launch; double flops; std::vector<double> us; };
static std::vector<Case> g_cases;


// ---------------------------------------------------------------------------------------------------------------
// Ring pipeline: NB slots of KB = 2 k-blocks; the DMA of stage t+NB-1 is issued in small pieces spread over the MFMAs
// of stage t (its slot was freed by the barrier that ended stage t-1), so no wave ever sits in a burst of VMEM issue.
// ---------------------------------------------------------------------------------------------------------------
template <int WC, int WS, int TC, int TS, int NB, int ABL>
__global__ void __launch_bounds__(WC* WS * 64, 1) ring_kernel(GemmArgs g, PlainFTParams ep) {
    typedef __bf16 T;
    constexpr int KB = 2;
    typedef GemmCfg<T, WC, WS, TC, TS, KB> C;
    typedef typename Mma<T>::Frag Frag;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wc = wave / WS, ws = wave % WS;
    const int ntiles = g.n_cblk * g.n_sblk;
    const int L = xcd_remap(blockIdx.x, ntiles);
    const int cblk = L % g.n_cblk;
    const int sblk = L / g.n_cblk;

    f32x16 acc[TC][TS];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TS; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int seg_total = g.seg_kblocks[0];
    const int nstages = g.ktot_blocks / KB;
    // per-wave DMA source pointers (advance by one stage = KB KiB per call)
    const unsigned voff = lane * 16;
    constexpr int NL = (ABL & HALF_LOAD) ? C::NW / 2 : C::NW;   // loader waves (HALF_LOAD: one per SIMD; its partner only computes)
    constexpr int LA = C::CT * KB / NL, LB = C::ST * KB / NL, LP = LA + LB;
    const unsigned char* wsrc[LA];
    const unsigned char* xsrc[LB];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int blk = wave + i * NL, rb = blk / KB, kb = blk % KB;
        wsrc[i] = reinterpret_cast<const unsigned char*>(g.W) + (((int64_t)(cblk * C::CT + rb) * g.w_stride_blocks + kb) << 10);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        const int blk = wave + i * NL, rb = blk / KB, kb = blk % KB;
        xsrc[i] = reinterpret_cast<const unsigned char*>(g.src[0]) + (((int64_t)(sblk * C::ST + rb) * seg_total + kb) << 10);
    }
    auto dma_w = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + voff), (lptr_t)(smem + slot * C::STAGE_BYTES + ((wave + i * NL) << 10)), 16, 0, 0);
            if constexpr (!(ABL & SAME_ADDR)) wsrc[i] += KB << 10;
        }
    };
    auto dma_x = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)(xsrc[i] + voff), (lptr_t)(smem + slot * C::STAGE_BYTES + ((C::CT * KB + wave + i * NL) << 10)), 16, 0, 0);
            if constexpr (!(ABL & SAME_ADDR)) xsrc[i] += KB << 10;
        }
    };
    Frag fa[2][TC], fb[2][TS];
    auto load_frags = [&](int slot, int kb, int set) __attribute__((always_inline)) {
        const unsigned char* a_base = smem + slot * C::STAGE_BYTES + ((wc * TC * KB) << 10) + lane * 16;
        const unsigned char* b_base = smem + slot * C::STAGE_BYTES + ((C::CT * KB + ws * TS * KB) << 10) + lane * 16;
#pragma unroll
        for (int i = 0; i < TC; ++i) fa[set][i] = *reinterpret_cast<const Frag*>(a_base + ((i * KB + kb) << 10));
#pragma unroll
        for (int j = 0; j < TS; ++j) fb[set][j] = *reinterpret_cast<const Frag*>(b_base + ((j * KB + kb) << 10));
    };
    auto mma = [&](int set) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TS; ++j) Mma<T>::run(fa[set][i], fb[set][j], acc[i][j]);
    };
    // interleave pattern for one MFMA group: per MFMA one ds_read (while they last) and NV VMEM pieces spread evenly
    auto pattern = [&](auto nv) __attribute__((always_inline)) {
        constexpr int NV = decltype(nv)::value, NM = TC * TS, ND = TC + TS;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < ND) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (NV > 0 && (i * NV) / NM != ((i + 1) * NV) / NM) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
    };

    constexpr int PRE = NB - 1;   // stages in flight ahead of the one being computed
    auto run = [&](auto is_loader) __attribute__((always_inline)) {
    constexpr bool LOADER = decltype(is_loader)::value;
#pragma unroll
    for (int s0 = 0; s0 < PRE; ++s0)
        if (s0 < nstages && LOADER) { dma_w(s0); dma_x(s0); }
    // stage 0 landed?  (in-order VMEM return: everything but the last PRE-1 stages)
    if (nstages >= PRE) __builtin_amdgcn_s_waitcnt(waitcnt_vm((PRE - 1) * LP)); else __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
    __syncthreads_lds_only();
    load_frags(0, 0, 0);

    int slot = 0;              // slot of the stage being computed
    int fill = PRE % NB;       // slot the DMA of this stage goes to (the one stage t-1 used)
    // steady state: stage t computes, stage t+PRE is fetched
    auto stage = [&](auto dma, auto last) __attribute__((always_inline)) {
        constexpr bool DMA = decltype(dma)::value && !(ABL & NO_GLDS) && LOADER, LAST = decltype(last)::value;
        constexpr bool DMAW = DMA && !(ABL & X_ONLY), DMAX = DMA && !(ABL & W_ONLY);
        const int nslot = (slot + 1 == NB) ? 0 : slot + 1;
        // group A: fragments of k-block 1, W pieces of the new stage
        load_frags(slot, 1, 1);
        if constexpr (DMAW) dma_w(fill);
        mma(0);
        pattern(std::integral_constant<int, DMAW ? LA : 0>{});
        __builtin_amdgcn_sched_barrier(0);
        // group B: next stage must have landed for every wave, and every wave is done reading this slot
        if constexpr (!LAST) {
            if constexpr (DMAW && DMAX) __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0((PRE - 2) * LP + LA));
            else __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(0));
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            load_frags(nslot, 0, 0);
            if constexpr (DMAX) dma_x(fill);
            mma(1);
            pattern(std::integral_constant<int, DMAX ? LB : 0>{});
        } else {
            mma(1);
        }
        __builtin_amdgcn_sched_barrier(0);
        slot = nslot;
        fill = (fill + 1 == NB) ? 0 : fill + 1;
    };
    {
        int t = 0;
        for (; t + PRE < nstages; ++t) stage(std::true_type{}, std::false_type{});
        for (; t + 1 < nstages; ++t) stage(std::false_type{}, std::false_type{});
        stage(std::false_type{}, std::true_type{});
    }
    };
    const uint64_t clk0 = (ABL & CLK) ? __builtin_amdgcn_s_memtime() : 0;
    if ((ABL & HALF_LOAD) && wave >= NL) run(std::false_type{}); else run(std::true_type{});
    if constexpr (ABL & CLK) { const uint64_t d = __builtin_amdgcn_s_memtime() - clk0; if (lane == 0) g_timing[(size_t)blockIdx.x * C::NW + wave] = d; }
    if constexpr (ABL & NO_EPI) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TS; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 123.456f) reinterpret_cast<float*>(ep.out)[blockIdx.x * 512 + threadIdx.x] = s;
    } else {
        EpiPlainFT<T>::template apply<TC, TS>(ep, acc, (cblk * C::CT + wc * TC) * 32, ((int64_t)sblk * C::ST + ws * TS) * 32, lane, sblk * WS + ws, 0, nullptr, 0, nullptr);
    }
}

template <int WC, int WS, int TC, int TS, int NB, int ABL> void add_ring(const char* name, int64_t S, int Cc, int K, void* W, void* X, void* out) {
    typedef GemmCfg<__bf16, WC, WS, TC, TS, 2> Cfg;
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.W = W; g.w_stride_blocks = K / 16; g.src[0] = X; g.seg_kblocks[0] = K / 16; g.nseg = 1; g.ktot_blocks = K / 16;
    g.n_cblk = Cc / (Cfg::CT * 32); g.n_sblk = (int)(S / (Cfg::ST * 32)); g.ksplit = 1;
    PlainFTParams p;
    p.out = out; p.N = Cc;
    auto kern = ring_kernel<WC, WS, TC, TS, NB, ABL>;
    constexpr int lds = NB * Cfg::STAGE_BYTES;
    static_assert(lds <= 160 * 1024, "LDS");
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    dim3 grid(g.n_cblk * g.n_sblk);
    g_cases.push_back({name, [=] { hipLaunchKernelGGL(kern, grid, dim3(Cfg::THREADS), lds, 0, g, p); }, 2.0 * S * Cc * K, {}});
}

template <int ABL, int DLY = 0, int PRIO = 0> void add(const char* name, int64_t S, int Cc, int K, void* W, void* X, void* out) {
    constexpr int WC = 2, WS = 4, TC = 4, TS = 2, KB = 4;
    typedef GemmCfg<__bf16, WC, WS, TC, TS, KB> Cfg;
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.W = W; g.w_stride_blocks = K / 16; g.src[0] = X; g.seg_kblocks[0] = K / 16; g.nseg = 1; g.ktot_blocks = K / 16;
    g.n_cblk = Cc / (Cfg::CT * 32); g.n_sblk = (int)(S / (Cfg::ST * 32)); g.ksplit = 1;
    PlainFTParams p;
    p.out = out; p.N = Cc;
    auto kern = abl_kernel<WC, WS, TC, TS, KB, ABL, DLY, PRIO>;
    constexpr int lds = 2 * Cfg::STAGE_BYTES;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    dim3 grid(g.n_cblk * g.n_sblk);
    g_cases.push_back({name, [=] { hipLaunchKernelGGL(kern, grid, dim3(Cfg::THREADS), lds, 0, g, p); }, 2.0 * S * Cc * K, {}});
}

int main(int argc, char** argv) {
    const int64_t S = argc > 1 ? atoll(argv[1]) : 65536;
    const int Cc = 1024, K = argc > 2 ? atoi(argv[2]) : 1024;
    void *W, *X, *o, *o2;
    CK(hipMalloc(&W, (size_t)Cc * K * 2)); CK(hipMalloc(&X, (size_t)S * K * 2)); CK(hipMalloc(&o, (size_t)S * Cc * 2)); CK(hipMalloc(&o2, (size_t)S * Cc * 2));
    {   // random operands: MFMA power (and with it the sustained clock) depends on the data
        std::vector<unsigned short> h(1 << 20);
        srand(1);
        for (auto& v : h) { float f = (rand() / (float)RAND_MAX - 0.5f) * 0.2f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
        for (size_t off = 0; off < (size_t)Cc * K; off += h.size()) CK(hipMemcpy((unsigned short*)W + off, h.data(), std::min(h.size(), (size_t)Cc * K - off) * 2, hipMemcpyHostToDevice));
        for (size_t off = 0; off < (size_t)S * K; off += h.size()) CK(hipMemcpy((unsigned short*)X + off, h.data(), std::min(h.size(), (size_t)S * K - off) * 2, hipMemcpyHostToDevice));
    }
    add<0>("baseline (product loop, plain FT store)", S, Cc, K, W, X, o);
    add<NO_EPI>("no epilogue stores", S, Cc, K, W, X, o);
    add<NO_EPI | SAME_ADDR>("no epi, DMA re-fetches the same k-blocks", S, Cc, K, W, X, o);
    add<NO_EPI | W_ONLY>("no epi, DMA weights only", S, Cc, K, W, X, o);
    add<NO_EPI | X_ONLY>("no epi, DMA activations only", S, Cc, K, W, X, o);
    add<NO_EPI | NO_GLDS>("no epi, no DMA in loop", S, Cc, K, W, X, o);
    add<NO_EPI | NO_GLDS | NO_BAR>("no epi, no DMA, no barrier", S, Cc, K, W, X, o);
    add<NO_EPI | NO_GLDS | NO_BAR | NO_FRAG>("no epi, no DMA, no barrier, no ds_read", S, Cc, K, W, X, o);
    add<NO_EPI | NO_FRAG>("no epi, no ds_read (DMA + barrier + MFMA)", S, Cc, K, W, X, o);
    add<NO_EPI | NO_BAR>("no epi, no barrier (DMA racing)", S, Cc, K, W, X, o);
    add<NO_EPI, 1>("no epi, upper half DMA +1 group", S, Cc, K, W, X, o);
    add<NO_EPI, 2>("no epi, upper half DMA +2 groups", S, Cc, K, W, X, o);
    add<NO_EPI, 3>("no epi, upper half DMA +3 groups", S, Cc, K, W, X, o);
    add<NO_EPI, 4>("no epi, upper half DMA +4 groups", S, Cc, K, W, X, o);
    add<NO_EPI, 0, 1>("no epi, upper half prio 1", S, Cc, K, W, X, o);
    add<NO_EPI, 2, 1>("no epi, upper half DMA +2, upper prio 1", S, Cc, K, W, X, o);
    add<NO_EPI, 2, 2>("no epi, upper half DMA +2, lower prio 1", S, Cc, K, W, X, o);
    add<0, 2>("upper half DMA +2 groups, with epilogue", S, Cc, K, W, X, o);
    add<NO_EPI | TIMING>("no epi, timing stamps", S, Cc, K, W, X, o);
    add_ring<2, 4, 4, 2, 4, 0>("ring 4x32K, plain FT store", S, Cc, K, W, X, o2);
    add_ring<2, 4, 4, 2, 4, NO_EPI>("ring 4x32K, no epi", S, Cc, K, W, X, o2);
    add_ring<2, 4, 4, 2, 4, NO_EPI | CLK>("clk ring", S, Cc, K, W, X, o2);
    add_ring<2, 4, 4, 2, 4, NO_EPI | CLK | NO_GLDS>("clk ring no DMA", S, Cc, K, W, X, o2);
    add_ring<2, 4, 4, 2, 4, NO_EPI | CLK | SAME_ADDR>("clk ring same addr", S, Cc, K, W, X, o2);
    add_ring<2, 4, 4, 2, 4, HALF_LOAD>("ring 4x32K, half loaders, plain FT store", S, Cc, K, W, X, o2);
    add_ring<2, 4, 4, 2, 4, NO_EPI | HALF_LOAD>("ring 4x32K, half loaders, no epi", S, Cc, K, W, X, o2);
    add_ring<2, 4, 4, 2, 4, NO_EPI | NO_GLDS>("ring 4x32K, no epi, no DMA", S, Cc, K, W, X, o2);
    add_ring<2, 4, 4, 2, 4, NO_EPI | SAME_ADDR>("ring 4x32K, no epi, same addr", S, Cc, K, W, X, o2);
    add_ring<2, 4, 4, 2, 4, NO_EPI | W_ONLY>("ring 4x32K, no epi, W only", S, Cc, K, W, X, o2);
    add_ring<2, 4, 4, 2, 4, NO_EPI | X_ONLY>("ring 4x32K, no epi, X only", S, Cc, K, W, X, o2);
    add<NO_EPI | FINE>("no epi, fine interleave", S, Cc, K, W, X, o);
    add<NO_EPI | FINE | NO_GLDS>("no epi, fine interleave, no DMA", S, Cc, K, W, X, o);
    add<NO_EPI | FINE, 1>("no epi, fine interleave, upper DMA +1", S, Cc, K, W, X, o);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemset(o, 0, (size_t)S * Cc * 2)); CK(hipMemset(o2, 0xff, (size_t)S * Cc * 2));
    for (auto& c : g_cases) c.launch();
    CK(hipDeviceSynchronize());
    for (auto& c : g_cases) if (c.name.find("baseline") != std::string::npos || c.name.find("plain FT store") != std::string::npos) c.launch();
    CK(hipDeviceSynchronize());
    {
        std::vector<unsigned short> h0((size_t)S * Cc), h1((size_t)S * Cc);
        CK(hipMemcpy(h0.data(), o, h0.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), o2, h1.size() * 2, hipMemcpyDeviceToHost));
        size_t bad = 0, nz = 0;
        for (size_t i = 0; i < h0.size(); ++i) { bad += h0[i] != h1[i]; nz += h0[i] != 0; }
        printf("ring vs baseline output: %zu mismatches of %zu (%zu non-zero)\n", bad, h0.size(), nz);
    }
    for (int r = 0; r < 7; ++r)
        for (auto& c : g_cases) {
            c.launch();
            CK(hipEventRecord(a, 0));
            for (int i = 0; i < 10; ++i) c.launch();
            CK(hipEventRecord(b, 0));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            c.us.push_back(ms * 1e2);
        }
    printf("256x256 / 8 waves / KB4, S=%lld C=%d K=%d bf16\n", (long long)S, Cc, K);
    for (auto& c : g_cases) {
        std::sort(c.us.begin(), c.us.end());
        const double mn = c.us.front(), md = c.us[c.us.size() / 2];
        printf("%-48s min %7.1f us (%6.0f TF)  median %7.1f us (%6.0f TF)\n", c.name.c_str(), mn, c.flops / mn * 1e-6, md, c.flops / md * 1e-6);
    }
    for (auto& c : g_cases) if (c.name.find("clk ") == 0) {
        c.launch();
        CK(hipDeviceSynchronize());
        const int nblk = (int)(S / 256) * (Cc / 256);
        std::vector<unsigned long long> h((size_t)nblk * 8);
        CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_timing), h.size() * 8));
        double tot = 0;
        for (auto v : h) tot += v;
        const double cyc = tot / h.size(), us = c.us[c.us.size() / 2];
        printf("%-24s K loop %9.0f s_memtime ticks per tile; kernel %7.1f us for %d tiles per CU -> >= %.2f ticks/ns\n", c.name.c_str(), cyc, us, nblk / 256, cyc * (nblk / 256) / (us * 1e3));
    }
    {   // per-wave stall breakdown of the TIMING variant (s_memtime ticks = 100 MHz constant clock)
        for (auto& c : g_cases) if (c.name.find("timing") != std::string::npos) { c.launch(); break; }
        CK(hipDeviceSynchronize());
        const int nblk = (int)(S / 256) * (Cc / 256);
        std::vector<unsigned long long> h((size_t)nblk * 8 * 4);
        CK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_timing), h.size() * 8));
        for (int w = 0; w < 8; ++w) {
            double tot = 0, wt = 0, br = 0, dm = 0;
            for (int b = 0; b < nblk; ++b) { const unsigned long long* d = &h[((size_t)b * 8 + w) * 4]; tot += d[0]; wt += d[1]; br += d[2]; dm += d[3]; }
            printf("wave %d: loop %8.1f ticks  vmcnt wait %7.1f  barrier %7.1f  DMA issue %7.1f   (per tile, mean over %d tiles)\n", w, tot / nblk, wt / nblk, br / nblk, dm / nblk, nblk);
        }
    }
    return 0;
}
