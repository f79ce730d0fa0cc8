// EXPERIMENT (not part of libdposer_hip.so; built only by tools/em_l0_pipe_bench.hip).  Round 2.
// Two ways to run post_dense + the Euler-Maruyama update of sampler step i TOGETHER WITH layer 0 (pre_dense + time bias +
// GroupNorm + SiLU, model.py:177-181) of step i + 1, both bit-identical to the shipped two-launch path and both measured
// WITHOUT gain on one MI355X (B = 65536, bf16, H = 1024):
//
//   shipped             post_dense + update (EpiEmStep, 64x128 tiles) 37 us  +  layer 0 (EpiGN, 256x256 tiles, K = 64) 46 us  = 83 us
//   EpiEmStepL0         layer 0 as a second GEMM inside the update kernel's epilogue                                  85 us
//   em_l0_pipe_kernel   producer / consumer pipeline in one persistent workgroup per CU                               89 us
//
// (whole sampler step 668-672 us with EpiEmStepL0 vs 674-676 us shipped in a same-process A/B: inside the noise.)
// Why: the step moves 288 MB through HBM (128 MB hidden activations in, 128 MB layer-0 output out, 32 MB of state) -- >= 52 us at
// the ~5.5 TB/s these access patterns reach -- and layer 0 costs 45 us of VALU chip-wide (64 sub-tiles of 32x32 per SIMD x
// ~0.7 us).  The two launches already overlap little of that, but neither does a fused kernel: in EpiEmStepL0 both resident
// workgroups of a CU run the memory-bound K loop and then the VALU-bound epilogue in lock-step; in the pipeline the first K
// loop, the two state updates (Philox + Box-Muller, 6 us each) and the drain are exposed (14 + 6 + 28 + 6 + 14 us of HBM-paced
// phases on paper, 89 measured).  A deeper pipeline (64-sample chunks, update overlapped with layer 0) could approach
// ~60-65 us = 3 % of the sampler; not pursued.
// The useful by-product is in the library: the bf16 image of a 32 x 32 accumulator tile, regrouped as TileIO<bf16>::store
// regroups it, IS the MFMA B operand of the next layer (no LDS round trip) -- noted in EXPERIMENTS.md (old 4.2).
#pragma once
#include "../../dposer_amd/csrc/gemm_api.h"

template <typename T> struct Mma;     // gemm.h
// The same step with the NEXT step's first layer (pre_dense + time bias + GroupNorm + SiLU, model.py:177-181) computed from
// the registers the new state sits in: the 64-channel state of a wave's 32 samples, converted to bf16 and regrouped exactly as
// TileIO<bf16>::store regroups it for the FT image, IS the MFMA B operand of that layer (4 k-blocks of 16), so layer 0 becomes
// 32 channel tiles x 4 MFMAs per wave with the weights read straight from L2 (128 KiB, shared by every wave) -- no `xin` round
// trip, no separate launch whose 128 MiB output store waits behind its own small K loop.  Per-tile arithmetic is the inference
// EpiGN (same code, same MFMA order: k-blocks 0..3 onto a zero accumulator), so the result equals the two-launch path bit
// for bit.  bf16, H = 1024 (gs = 32), Cp = Dpad = 64, the 64 x 128 tiling only.
struct EmStepL0Params {
    EmStepParams em;
    const void* w0;          // packed layer-0 weights, FT [H][w0_stride_blocks * 16]
    int w0_stride_blocks;
    const float* bias0;      // [H] time-table row of the next step: pre_dense bias + time projection (layer 0)
    const float* gamma0;
    const float* beta0;
    void* h0;                // FT [Spad][H]: output of layer 0 for the next step
    int H;
};
template <typename T> struct EpiEmStepL0 {
    static_assert(sizeof(T) == 2, "bf16 only (the fp32 parity mode keeps the two-launch path)");
    typedef EmStepL0Params Params;
    static constexpr int kH = 1024;
    static constexpr int kRingPerWave = 3 * kH * 4;      // this wave's copy of bias / gamma / beta in the idle K-loop ring
    template <int TC, int TS>
    __device__ static inline void apply_ring(const Params& pp, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, int, int, const float*, int, unsigned char*, unsigned char* ring) {
        static_assert(TC == 2 && TS == 1, "wave tile = the whole 64-channel state of 32 samples");
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        typedef EpiEmStep<T> Em;
        typedef typename Mma<T>::Frag Frag;
        const EmStepParams& p = pp.em;
        const int j = lane & 31, hi = lane >> 5;
        // layer-0 parameters -> LDS (DMA, 1 KiB per instruction); they land while the state update runs
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float* src = a == 0 ? pp.bias0 : (a == 1 ? pp.gamma0 : pp.beta0);
#pragma unroll
            for (int k = 0; k < kH / 256; ++k)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + k * 256 + lane * 4), (lptr_t)(ring + (a * (kH / 256) + k) * 1024), 16, 0, 0);
        }
        const unsigned char* wbase = reinterpret_cast<const unsigned char*>(pp.w0) + lane * 16;
        Frag fa[2][4];
        auto load_w = [&](int ct, int set) __attribute__((always_inline)) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) fa[set][kb] = *reinterpret_cast<const Frag*>(wbase + (((int64_t)ct * pp.w0_stride_blocks + kb) << 10));
        };
        load_w(0, 0);
        // ---- Euler-Maruyama update of the two 32-channel tiles; the bf16 image of the new state stays in registers ----
        const typename Em::Scal sc = Em::scalars(p);
        Frag fb[4];
#pragma unroll
        for (int tc = 0; tc < 2; ++tc) {
            const int c0 = cbase + tc * 32;
            const int64_t s = sbase + j;
            const int64_t tb = ft_tile_base<float>(sbase, c0, p.Cp);
            float x[16], xm[16];
            TileIO<float>::load(p.x_ft + tb, lane, x);
            Em::tile(p, sc, acc[tc][0], c0, s, hi, x, xm);
            TileIO<float>::store(p.x_ft + tb, lane, x);
            if (p.x_mean_ft) TileIO<float>::store(p.x_mean_ft + tb, lane, xm);
#pragma unroll
            for (int h = 0; h < 2; ++h) {               // = TileIO<bf16>::store's regrouping: lane (j, hi) holds chunk kh = hi of k-block h
                unsigned a0 = TileIO<T>::pack2(x[8 * h + 0], x[8 * h + 1]), a1 = TileIO<T>::pack2(x[8 * h + 2], x[8 * h + 3]);
                unsigned b0 = TileIO<T>::pack2(x[8 * h + 4], x[8 * h + 5]), b1 = TileIO<T>::pack2(x[8 * h + 6], x[8 * h + 7]);
                TileIO<T>::swap_halves(a0, b0);
                TileIO<T>::swap_halves(a1, b1);
                u32x4 o = {a0, a1, b0, b1};
                fb[2 * tc + h] = *reinterpret_cast<Frag*>(&o);
            }
        }
        __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));      // parameter DMA (and the first weight fragments) landed
        asm volatile("" ::: "memory");
        // ---- layer 0 of the next step: 32 channel tiles, K = 64 ----
        GNParams gp = {};                               // (bias / gamma / beta are read from the LDS copy; no dropout at inference)
        gp.out = pp.h0;
        gp.H = pp.H;
        const float* lpar = reinterpret_cast<const float*>(ring);
        typename EpiGN<T, false, 0>::Carry cy;
#pragma unroll 2
        for (int ct = 0; ct < kH / 32; ++ct) {
            const int set = ct & 1;
            if (ct + 1 < kH / 32) load_w(ct + 1, set ^ 1);
            f32x16 a;
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = 0.f;
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) Mma<T>::run(fa[set][kb], fb[kb], a);
            EpiGN<T, false, 0>::template sub<1, 1, 0, 0, -1>(gp, cy, a, ct * 32, sbase, lane, 0, 0, lpar + ct * 32, kH, nullptr);
        }
    }
};


struct EmL0PipeArgs {
    const void* wpost;      // packed post_dense weights, FT [64][H]
    const void* h;          // FT [Spad][H]: output of the last hidden layer of this step
    int n_chunks;           // Spad / 128
    EmStepL0Params p;
};

// ---- producer / consumer form --------------------------------------------------------------------------------------------
//   waves 0-3 (producers, one per SIMD): the K loop of chunk j (128 samples, wave = 32 samples x 64 channels; DMA ring of 8
//       stages, the fetch stream runs across chunk boundaries), then the state update in registers; the bf16 image of the new
//       state goes to the consumers through 16 KiB of LDS;
//   waves 4-7 (consumers): layer 0 of chunk j - 1: 32 channel tiles x (4 MFMAs against weights read from L2 + the inference
//       EpiGN code), one tile per K-loop stage -- the stage barrier is the only synchronisation (K = 1024 = 32 stages of 2
//       k-blocks, H = 1024 = 32 channel tiles).
// The last chunk of a workgroup has no K loop beside it: its producers take channel tiles 16-31 of their own samples, the
// consumers tiles 0-15.
namespace em_l0_pipe {
constexpr int H = 1024, NCT = H / 32, KBLK = H / 16;          // channel tiles of layer 0; k-blocks of post_dense
constexpr int NB = 8, KB = 2, NSTAGE = KBLK / KB;             // ring slots, k-blocks per stage, stages per chunk
constexpr int STAGE_BLOCKS = (2 + 4) * KB, STAGE_BYTES = STAGE_BLOCKS * 1024;
constexpr int DMA_PER_WAVE = STAGE_BLOCKS / 4;
constexpr int RING_BYTES = NB * STAGE_BYTES, PAR_BYTES = 3 * H * 4, HAND_BYTES = 4 * 4 * 1024;
constexpr int LDS_BYTES = RING_BYTES + PAR_BYTES + HAND_BYTES;
static_assert(NSTAGE == NCT, "one channel tile of layer 0 per K-loop stage");
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
}   // namespace em_l0_pipe

__global__ void __launch_bounds__(512, 1) em_l0_pipe_kernel(EmL0PipeArgs a) {
    using namespace em_l0_pipe;
    typedef __bf16 T;
    typedef Mma<T>::Frag Frag;
    typedef EpiEmStep<T> Em;
    typedef EpiGN<T, false, 0> Gn;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* ring = smem;
    float* par = reinterpret_cast<float*>(smem + RING_BYTES);            // [bias | gamma | beta][H]
    unsigned char* hand = smem + RING_BYTES + PAR_BYTES;                 // [unit][k-block][64 lanes x 16 B]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef EM_L0_PIPE_ROLE_ODD_EVEN      // (A/B: which wave ids share a SIMD)
    const bool producer = (wave & 1) == 0;
    const int u = wave >> 1;
#else
    const bool producer = wave < 4;
    const int u = wave & 3;                                              // 32-sample unit of the chunk
#endif
    const int j32 = lane & 31, hi = lane >> 5;
    const EmStepL0Params& pp = a.p;
    const EmStepParams& p = pp.em;

    for (int i = threadIdx.x; i < 3 * H / 4; i += 512) {
        const int arr = i / (H / 4), c4 = i % (H / 4);
        const float* src = arr == 0 ? pp.bias0 : (arr == 1 ? pp.gamma0 : pp.beta0);
        reinterpret_cast<f32x4*>(par)[i] = reinterpret_cast<const f32x4*>(src)[c4];
    }
    const int nch = (a.n_chunks - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // chunks b, b + grid, ...
    auto chunk_of = [&](int j) __attribute__((always_inline)) { return (int)blockIdx.x + j * (int)gridDim.x; };
    __syncthreads();

    // ---- layer 0, one channel tile: 4 MFMAs (K = 64) + GroupNorm + SiLU + FT store ----
    GNParams gp = {};
    gp.out = pp.h0;
    gp.H = H;
    const unsigned char* wbase = reinterpret_cast<const unsigned char*>(pp.w0) + lane * 16;
    Frag fw[2][4];
    // (the fragment set is a compile-time index everywhere: a run-time one turns the register array into select chains)
    auto load_w = [&](int ct, auto set_c) __attribute__((always_inline)) {
        constexpr int set = decltype(set_c)::value;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) fw[set][kb] = *reinterpret_cast<const Frag*>(wbase + (((int64_t)ct * pp.w0_stride_blocks + kb) << 10));
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    Gn::Carry cy;
    auto do_ct = [&](int ct, auto set_c, const Frag (&fb)[4], int64_t sbase) __attribute__((always_inline)) {
        constexpr int set = decltype(set_c)::value;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) Mma<T>::run(fw[set][kb], fb[kb], acc);
#ifndef PIPE_DBG_NO_GN
        Gn::template sub<1, 1, 0, 0, -1>(gp, cy, acc, ct * 32, sbase, lane, 0, 0, par + ct * 32, H, nullptr);
#else
        if (acc[0] == 123.f) reinterpret_cast<float*>(gp.out)[lane] = acc[1];
#endif
    };

    if (producer) {
        const int total = nch * NSTAGE;                                   // stages of this workgroup's whole fetch stream
        auto issue = [&](int g) __attribute__((always_inline)) {
            const int chunk = chunk_of(g / NSTAGE), kb0 = (g % NSTAGE) * KB;
            unsigned char* slot = ring + (g % NB) * STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < DMA_PER_WAVE; ++i) {
                const int blk = u + 4 * i;                                // 0..3: weights (tile, kb); 4..11: activations (unit, kb)
                const unsigned char* src;
                if (i == 0) src = reinterpret_cast<const unsigned char*>(a.wpost) + (((int64_t)(blk >> 1) * KBLK + kb0 + (blk & 1)) << 10);
                else src = reinterpret_cast<const unsigned char*>(a.h) + ((((int64_t)chunk * 4 + ((blk - 4) >> 1)) * KBLK + kb0 + ((blk - 4) & 1)) << 10);
                __builtin_amdgcn_global_load_lds((gptr_t)(src + lane * 16), (lptr_t)(slot + (blk << 10)), 16, 0, 0);
            }
        };
        for (int g = 0; g < NB - 1 && g < total; ++g) issue(g);
        const typename Em::Scal sc = Em::scalars(p);
        Frag fb[4];
        int64_t sbase = 0;
        for (int j = 0; j < nch; ++j) {
            f32x16 acc[2];
#pragma unroll
            for (int tc = 0; tc < 2; ++tc)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tc][r] = 0.f;
            for (int t = 0; t < NSTAGE; ++t) {
                const int g = j * NSTAGE + t;
                // stage g has landed for this wave (VMEM returns in order; at most NB - 2 later stages may be in flight)
                if (g + NB - 2 >= total) __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
                else __builtin_amdgcn_s_waitcnt(waitcnt_vm((NB - 2) * DMA_PER_WAVE));
                __syncthreads_lds_only();
                if (g + NB - 1 < total) issue(g + NB - 1);               // into the slot stage g - 1 used: every wave is past its reads
                const unsigned char* slot = ring + (g % NB) * STAGE_BYTES + lane * 16;
                Frag fa[2][KB], fx[KB];
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
                    for (int tc = 0; tc < 2; ++tc) fa[tc][kb] = *reinterpret_cast<const Frag*>(slot + ((tc * KB + kb) << 10));
                    fx[kb] = *reinterpret_cast<const Frag*>(slot + ((2 * KB + u * KB + kb) << 10));
                }
#pragma unroll
                for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                    for (int tc = 0; tc < 2; ++tc) Mma<T>::run(fa[tc][kb], fx[kb], acc[tc]);
            }
            // ---- Euler-Maruyama update of this wave's 32 samples x 64 channels ----
            sbase = ((int64_t)chunk_of(j) * 4 + u) * 32;
#pragma unroll
            for (int tc = 0; tc < 2; ++tc) {
                const int c0 = tc * 32;
                const int64_t tb = ft_tile_base<float>(sbase, c0, p.Cp);
                float x[16], xm[16];
                TileIO<float>::load(p.x_ft + tb, lane, x);
#ifndef PIPE_DBG_NO_EM
                Em::tile(p, sc, acc[tc], c0, sbase + j32, hi, x, xm);
#else
                for (int r = 0; r < 16; ++r) { x[r] += acc[tc][r]; xm[r] = x[r]; }
#endif
                TileIO<float>::store(p.x_ft + tb, lane, x);
                if (p.x_mean_ft) TileIO<float>::store(p.x_mean_ft + tb, lane, xm);
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {          // = TileIO<bf16>::store's regrouping: lane (j, hi) holds chunk kh = hi of k-block hh
                    unsigned a0 = TileIO<T>::pack2(x[8 * hh + 0], x[8 * hh + 1]), a1 = TileIO<T>::pack2(x[8 * hh + 2], x[8 * hh + 3]);
                    unsigned b0 = TileIO<T>::pack2(x[8 * hh + 4], x[8 * hh + 5]), b1 = TileIO<T>::pack2(x[8 * hh + 6], x[8 * hh + 7]);
                    TileIO<T>::swap_halves(a0, b0);
                    TileIO<T>::swap_halves(a1, b1);
                    u32x4 o = {a0, a1, b0, b1};
                    fb[2 * tc + hh] = *reinterpret_cast<Frag*>(&o);
                    *reinterpret_cast<u32x4*>(hand + ((u * 4 + 2 * tc + hh) << 10) + lane * 16) = o;
                }
            }
            // the state stores must not be outstanding when the next chunk counts its DMA pieces (loads and stores share vmcnt)
            __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(0));
            __syncthreads_lds_only();                                     // hand-off visible
        }
        // last chunk: no K loop left to run beside layer 0 -- take its upper half from the registers the operand is still in
#ifdef PIPE_DBG_NO_TAIL
        if (false) {
#else
        if (nch > 0) {
#endif
            load_w(NCT / 2, S0{});
            for (int ct = NCT / 2; ct < NCT; ct += 2) {
                load_w(ct + 1, S1{});
                do_ct(ct, S0{}, fb, sbase);
                if (ct + 2 < NCT) load_w(ct + 2, S0{});
                do_ct(ct + 1, S1{}, fb, sbase);
            }
        }
    } else {
#ifdef PIPE_DBG_NO_CONSUMER
        return;
#endif
        Frag fb[4];
        int64_t sbase = 0;
        for (int j = 0; j <= nch; ++j) {
            const bool have = j >= 1;                                     // layer 0 of chunk j - 1
            const bool beside = j < nch;                                  // a K loop (of chunk j) runs beside it: stage barriers
            const int ct_end = beside ? NCT : NCT / 2;
            if (have) load_w(0, S0{});
            for (int ct = 0; ct < ct_end; ct += 2) {
                if (beside) __syncthreads_lds_only();
                if (have) {
                    load_w(ct + 1, S1{});
                    do_ct(ct, S0{}, fb, sbase);
                }
                if (beside) __syncthreads_lds_only();
                if (have) {
                    if (ct + 2 < ct_end) load_w(ct + 2, S0{});
                    do_ct(ct + 1, S1{}, fb, sbase);
                }
            }
            if (beside) {
                __syncthreads_lds_only();                                 // producers wrote chunk j's operand
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) fb[kb] = *reinterpret_cast<const Frag*>(hand + ((u * 4 + kb) << 10) + lane * 16);
                sbase = ((int64_t)chunk_of(j) * 4 + u) * 32;
            }
        }
    }
}

static inline hipError_t launch_em_l0_pipe(const EmL0PipeArgs& a, hipStream_t stream) {
    static int n_cu[64] = {};
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    dev &= 63;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(em_l0_pipe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, em_l0_pipe::LDS_BYTES);
        if (e != hipSuccess) return e;
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, dev);
        if (e != hipSuccess) return e;
        n_cu[dev] = prop.multiProcessorCount;
        attr_set[dev] = true;
    }
    const int grid = a.n_chunks < n_cu[dev] ? a.n_chunks : n_cu[dev];
    hipLaunchKernelGGL(em_l0_pipe_kernel, dim3(grid), dim3(512), em_l0_pipe::LDS_BYTES, stream, a);
    return hipGetLastError();
}
