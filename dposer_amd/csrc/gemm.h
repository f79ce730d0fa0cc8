// MFMA GEMM core for the score network on gfx950 (CDNA4).
//
// Every layer of ScoreModelFC (reference lib/algorithms/advanced/model.py:141-196) and every
// backward GEMM of its training step is one instance of
//
//        D[c][s] = sum_k  Wp[c][k] * X[s][k]          c = output channel, s = sample
//
// computed in the "swapped" orientation: the MFMA A operand is the (packed) weight matrix, the
// B operand is the activation matrix, so that in the 32x32 accumulator tile
//        lane  = sample (l & 31),   registers = 16 of the 32 channels of ONE GroupNorm group
//        channel(reg, lane) = 32*tile + (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
// GroupNorm statistics (32 contiguous channels per group, model.py:112) therefore reduce over a
// lane's own registers plus ONE cross-lane exchange with lane^32 -- no LDS, no 32-lane shuffles.
// Both operands live in HBM in the fragment-tiled layout of common.h, so staging a tile is a
// stream of coalesced 1-KiB copies and LDS reads are conflict-free ds_read_b128.
//
// T = __bf16 : v_mfma_f32_32x32x16_bf16 (fp32 accumulate)       -- throughput mode
// T = float  : v_mfma_f32_32x32x2_f32   (exact fp32 fma chain)  -- parity mode
#pragma once
#include <type_traits>

#include "common.h"

constexpr int GEMM_MAX_SEG = 8;

struct GemmArgs {
    const void* W;                  // packed weights, FT [Cpad][Ktot]
    const void* src[GEMM_MAX_SEG];  // activation segments, FT [Spad][Kseg]; concatenated along k
    int seg_kblocks[GEMM_MAX_SEG];  // k-blocks (of FT<T>::KBS) per segment, multiples of KB
    int seg_stride_blocks[GEMM_MAX_SEG];   // k-blocks per row-block of the segment's ARRAY when only a K-prefix of it is reduced over
                                    // (0: = seg_kblocks; LBS blend GEMM over the pose-feature columns of the joints that are posed)
    int nseg;
    int ktot_blocks;                // sum of seg_kblocks = k-blocks reduced over
    int w_stride_blocks;            // k-blocks per packed weight row-block (>= ktot_blocks: a K-prefix may be used)
    int n_cblk;                     // channel block tiles in the grid
    int n_sblk;                     // sample block tiles in the grid
    int ksplit;                     // >1: reduction split over blockIdx.y (wgrad); each split gets
                                    //     seg_kblocks[0]/ksplit k-blocks of the (single) segment
    double alg_flops;               // algorithmic FLOPs of this launch (2*M*N*K on the un-padded problem); profiling only
    int panel_order;                // 1: split-K launches whose W panel is the big stream (LBS blend gradient: W = d_offsets, a few sample
                                    //    tiles per channel tile) -- the tiles of one (channel tile, split) run side by side on ONE XCD
    int split_segments;             // 1 (with ksplit == nseg, equal segments): split i reduces over SEGMENT i instead of a slice of segment 0
                                    //    (the time-branch dgrad at small batches: one split per layer's dy)
};

#include "gemm_kloop_asm.h"
#ifndef DPOSER_KLOOP_ASM      // (tuner A/B switch: 0 = the hipcc-scheduled steady-state stage)
#define DPOSER_KLOOP_ASM 1
#endif
#ifndef DPOSER_KLOOP_ASM_MID  // (tuner A/B switch for the 128x128 / 4-wave tiling's asm stage)
#define DPOSER_KLOOP_ASM_MID 1
#endif

template <typename T> struct Mma;
template <> struct Mma<__bf16> {
    typedef bf16x8 Frag;
    __device__ static inline void run(const Frag& a, const Frag& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    typedef f32x4 Frag;
    __device__ static inline void run(const Frag& a, const Frag& b, f32x16& c) {
#pragma unroll
        for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], c, 0, 0, 0);
    }
};

// XCD-aware, bijective block remap (guide T1): hardware places block b on XCD b % 8; give each
// XCD a contiguous run of logical tiles so that tiles sharing a sample panel share an L2.
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

__device__ __forceinline__ void __syncthreads_lds_only() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <typename T, int WC, int WS, int TC, int TS, int KB>
struct GemmCfg {
    static constexpr int NW = WC * WS;
    static constexpr int THREADS = NW * 64;
    static constexpr int CT = WC * TC;   // 32-channel tiles per block
    static constexpr int ST = WS * TS;   // 32-sample tiles per block
    static constexpr int BLOCKS_PER_STAGE = (CT + ST) * KB;
    static constexpr int STAGE_BYTES = BLOCKS_PER_STAGE * 1024;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
    static constexpr int LPW_A = CT * KB / NW;   // 1-KiB weight blocks copied per wave per stage
    static constexpr int LPW_B = ST * KB / NW;   // 1-KiB activation blocks copied per wave per stage
    static constexpr int LPW = LPW_A + LPW_B;
    static_assert((CT * KB) % NW == 0 && (ST * KB) % NW == 0, "stage must split evenly over the waves");
};

// Register budget hint: epilogues may ask for >= kMinWaves waves per SIMD (caps the allocator at 512 / kMinWaves).
template <typename Epi, typename = void> struct EpiMinWaves { static constexpr int value = 1; };
template <typename Epi> struct EpiMinWaves<Epi, decltype((void)Epi::kMinWaves)> { static constexpr int value = Epi::kMinWaves; };

// Epilogues may ask for up to 3 per-channel fp32 arrays (bias / gamma / beta) to be staged into LDS at kernel
// start: global-load latency of those small vectors is then hidden under the whole K loop instead of being paid
// once per channel tile in the epilogue.
template <typename Epi, typename = void> struct EpiParamArrays { static constexpr int value = 0; };
template <typename Epi> struct EpiParamArrays<Epi, decltype((void)Epi::kParamArrays)> { static constexpr int value = Epi::kParamArrays; };

// Per-wave LDS scratch (bytes) an epilogue may ask for (transposed tile stores).
template <typename Epi, typename = void> struct EpiScratch { static constexpr int value = 0; };
template <typename Epi> struct EpiScratch<Epi, decltype((void)Epi::kScratchPerWave)> { static constexpr int value = Epi::kScratchPerWave; };

// Epilogues that prefetch their own operands through the (idle) K-loop ring declare kRingPerWave bytes and apply_ring(...).
// Optional epilogue work IN FRONT of the K loop (round 6: latency-bound small launches, e.g. the sampler's fused post_dense + Euler-Maruyama step, whose
// epilogue -- state tile loads, Philox normals -- was as long as its K loop).  `Epi::Pre<TC, TS>` = register state; `Epi::pre_issue` issues global loads
// into it BEFORE the first DMA (VMEM returns in order and the loop's vmcnt waits count the YOUNGEST operations: older loads do not disturb them);
// `Epi::pre_compute` = arithmetic that needs no accumulator, run behind the prologue's DMA issue, under its latency; the epilogue is then `apply_pre`.
template <typename Epi, int TC, int TS, typename = void> struct EpiPre { struct type {}; static constexpr bool value = false; };
template <typename Epi, int TC, int TS> struct EpiPre<Epi, TC, TS, decltype((void)sizeof(typename Epi::template Pre<TC, TS>))> {
    typedef typename Epi::template Pre<TC, TS> type;
    static constexpr bool value = true;
};
template <typename Epi, typename = void> struct EpiRing { static constexpr int value = 0; };
#ifndef DPOSER_NO_EPI_RING   // (A/B switch for the tuner)
template <typename Epi> struct EpiRing<Epi, decltype((void)Epi::kRingPerWave)> { static constexpr int value = Epi::kRingPerWave; };
#endif

// The kernel.  One workgroup = one output tile (x one k-split).
// Main loop: double-buffered LDS ring filled with global_load_lds_dwordx4 (the fragment-tiled HBM image IS the LDS
// image: one 1-KiB block per wave instruction, LDS address = wave-uniform base + lane*16, no VGPR round trip) and
// double-buffered operand fragments: the ds_reads of k-block kb+1 are issued BEFORE the MFMAs of k-block kb, so LDS
// latency hides under the matrix pipe (left to itself hipcc reuses one fragment register set and serialises
// read -> wait -> MFMA per k-block).  The stage barrier sits in front of the LAST MFMA group of a stage, followed by
// the DMA issue of stage t+2 and the first fragment reads of stage t+1.
// NB = 2 is that loop.  NB > 2 selects the RING pipeline the large tilings use: NB slots of KB k-blocks; the DMA of
// stage t+NB-1 goes to the slot stage t-1 used (freed by the barrier that ended it) and is issued in single pieces
// BETWEEN the MFMAs of stage t (weights in the groups before the barrier, activations after it), together with the
// ds_reads of the next k-block, under an explicit sched_group_barrier pattern.  Why: `tools/gemm_ablate.hip` shows the
// 2-slot loop loses ~25 % to the DMA issue burst -- after each barrier every wave sits ~450-970 cycles in VMEM issue
// (64 KiB per CU through the 64 B/clk vector-memory path) with no MFMA behind it, while DMA latency itself costs
// ~50 cycles per stage; spreading the pieces removes the burst (+9-11 % on the 256x256 loop, bit-identical results).
// Variants that were built and measured without gain (DMA issue interleaved between MFMA GROUPS under s_setprio,
// delayed DMA issue for the second wave of each SIMD, register staging, persistent workgroups) are in docs/experiments_rounds_1-4.md (4.1).
// Epi::apply(params, acc, channel_base, sample_base, lane, wave row id, split, staged params, stride, scratch).
// One output tile (cblk, sblk) [x one k-split] by the C::THREADS threads whose workgroup-local id is `tid` (the whole workgroup in
// gemm_ft_kernel; a persistent kernel -- gemm_sampler.hip -- calls it tile after tile, or with two half-workgroups side by side on
// disjoint LDS regions: the barriers inside are workgroup-wide, so both halves must run the same number of stages).
template <typename T, int WC, int WS, int TC, int TS, int KB, typename Epi, int NB>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const typename Epi::Params& ep, int cblk, int sblk, int split, unsigned char* smem, int tid) {
#ifdef DPOSER_PHASE_STAMPS     // tools/tile_phase_probe.hip: where a tile's time goes (prologue / K loop / epilogue) and when it ran
    const uint64_t ps_rt0 = __builtin_amdgcn_s_memrealtime(), ps_t0 = __builtin_amdgcn_s_memtime();
    uint64_t ps_t1 = 0, ps_t2 = 0;
#endif
    static_assert(KB % 2 == 0, "fragment double buffering assumes an even number of k-blocks per stage");
    static_assert(NB >= 2 && NB <= 4, "ring depth");
    typedef GemmCfg<T, WC, WS, TC, TS, KB> C;
    typedef typename Mma<T>::Frag Frag;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WS, ws = wave % WS;

    constexpr int NPAR = EpiParamArrays<Epi>::value;
    float* lds_par = reinterpret_cast<float*>(smem + NB * C::STAGE_BYTES);   // [NPAR][CT*32]
    if constexpr (NPAR > 0) {
        for (int i = tid; i < NPAR * C::CT * 32; i += C::THREADS) {
            const int a = i / (C::CT * 32), c = i % (C::CT * 32);
            lds_par[i] = Epi::param_array(ep, a)[cblk * C::CT * 32 + c];
        }
    }

    f32x16 acc[TC][TS];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TS; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- stage bookkeeping -------------------------------------------------------------------
    // (segment table is only ever indexed through unrolled selects: dynamic indexing of a
    //  kernel-argument array would push the whole struct to scratch)
    auto seg_ptr = [&](int i) __attribute__((always_inline)) {
        const void* p = g.src[0];
#pragma unroll
        for (int k = 1; k < GEMM_MAX_SEG; ++k) p = (i == k) ? g.src[k] : p;
        return reinterpret_cast<const unsigned char*>(p);
    };
    auto seg_blocks = [&](int i) __attribute__((always_inline)) {
        int n = g.seg_kblocks[0];
#pragma unroll
        for (int k = 1; k < GEMM_MAX_SEG; ++k) n = (i == k) ? g.seg_kblocks[k] : n;
        return n;
    };
    auto seg_stride_of = [&](int i) __attribute__((always_inline)) {
        int n = g.seg_stride_blocks[0], m = g.seg_kblocks[0];
#pragma unroll
        for (int k = 1; k < GEMM_MAX_SEG; ++k) { n = (i == k) ? g.seg_stride_blocks[k] : n; m = (i == k) ? g.seg_kblocks[k] : m; }
        return n > 0 ? n : m;
    };
    int seg = 0, seg_kb = 0;   // position of the NEXT stage to fetch
    int w_kb = 0;              // weight k-block of the next stage
    int seg_total = g.seg_kblocks[0];          // k-blocks of the current segment
    int seg_stride = seg_stride_of(0);         // k-blocks per row-block of the current segment's array
    int seg_end = seg_total;                   // first k-block NOT to fetch from this segment
    int nstages = g.ktot_blocks / KB;
    const unsigned char* sbase = seg_ptr(0);
    if (g.ksplit > 1 && g.split_segments) {
        seg = split;
        sbase = seg_ptr(seg);
        seg_stride = seg_stride_of(seg);
        nstages = seg_total / KB;               // (equal segments)
        w_kb = split * seg_total;
    } else if (g.ksplit > 1) {
        const int per = seg_total / g.ksplit;
        nstages = per / KB;
        seg_kb = split * per;
        w_kb = seg_kb;
        seg_end = seg_kb + per;
    }

    auto fetch_w = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < C::LPW_A; ++i) {
            const int blk = wave + i * C::NW;
            const int rb = blk / KB, kb = blk % KB;
            const unsigned char* p = reinterpret_cast<const unsigned char*>(g.W) +
                                     (((int64_t)(cblk * C::CT + rb) * g.w_stride_blocks + w_kb + kb) << 10);
            __builtin_amdgcn_global_load_lds((gptr_t)(p + lane * 16), (lptr_t)(smem + buf * C::STAGE_BYTES + (blk << 10)), 16, 0, 0);
        }
    };
    auto fetch_x = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < C::LPW_B; ++i) {
            const int blk = wave + i * C::NW;
            const int rb = blk / KB, kb = blk % KB;
            const unsigned char* p = sbase + (((int64_t)(sblk * C::ST + rb) * seg_stride + seg_kb + kb) << 10);
            __builtin_amdgcn_global_load_lds((gptr_t)(p + lane * 16), (lptr_t)(smem + buf * C::STAGE_BYTES + ((C::CT * KB + blk) << 10)), 16, 0, 0);
        }
    };
    auto fetch_advance = [&]() __attribute__((always_inline)) {
        seg_kb += KB;
        w_kb += KB;
        if (seg_kb >= seg_end && seg + 1 < g.nseg) {
            ++seg;
            seg_kb = 0;
            seg_total = seg_blocks(seg);
            seg_stride = seg_stride_of(seg);
            seg_end = seg_total;
            sbase = seg_ptr(seg);
        }
    };
    auto fetch_glds = [&](int buf) __attribute__((always_inline)) {
        fetch_w(buf);
        fetch_x(buf);
        fetch_advance();
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

#ifdef DPOSER_KLOOP_PRIO      // (tuner A/B: static priority for the second-dispatched half of the workgroup, MI355X_MICROARCH.md "two waves per SIMD" item 4)
    if (DPOSER_KLOOP_PRIO == 1 && wave >= C::NW / 2) __builtin_amdgcn_s_setprio(1);
    if (DPOSER_KLOOP_PRIO == 2 && wave < C::NW / 2) __builtin_amdgcn_s_setprio(1);
#endif
    typedef EpiPre<Epi, TC, TS> PreT;
    typename PreT::type pre_state;
    const int e_cbase = (cblk * C::CT + wc * TC) * 32;
    const int64_t e_sbase = ((int64_t)sblk * C::ST + ws * TS) * 32;
    if constexpr (PreT::value) Epi::template pre_issue<TC, TS>(ep, pre_state, e_cbase, e_sbase, lane);
    if constexpr (NB == 2) {
        fetch_glds(0);
        if (nstages > 1) fetch_glds(1);
        if constexpr (PreT::value) Epi::template pre_compute<TC, TS>(ep, pre_state, e_cbase, e_sbase, lane);
        if (nstages > 1) __builtin_amdgcn_s_waitcnt(waitcnt_vm(C::LPW)); else __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
        __syncthreads_lds_only();
        load_frags(0, 0, 0);
        for (int t = 0; t < nstages; ++t) {
            const int buf = t & 1;
    #pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                if (kb + 1 < KB) {
                    load_frags(buf, kb + 1, (kb + 1) & 1);
                } else if (t + 1 < nstages) {
                    // stage t+1 must have landed (its DMA was issued one full stage ago); every wave is done reading buf
                    __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(0));
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    if (t + 2 < nstages) fetch_glds(buf);            // refill the buffer we just finished reading
                    load_frags(buf ^ 1, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                mma(kb & 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        constexpr int PRE = NB - 1;                             // stages in flight ahead of the one being computed
        constexpr int NM = TC * TS * (sizeof(T) == 4 ? 4 : 1);  // MFMA instructions per k-block
        // one MFMA group's issue order: MFMA, ds_read (while they last), and NV DMA pieces spread evenly
        auto pattern = [&](auto nv) __attribute__((always_inline)) {
            constexpr int NV = decltype(nv)::value;
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i < TC + TS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                if (NV > 0 && (i * NV) / NM != ((i + 1) * NV) / NM) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
            }
        };
        const int npre = nstages < PRE ? nstages : PRE;
        // Hand-placed steady-state stage (gemm_kloop_asm.h) for the shipped 256x256 bf16 tiling; every DMA of such a kernel is issued
        // from asm (saddr form: SGPR base per piece + one VGPR offset per operand), the compiler tracks none of them.
        constexpr bool ASM_BIG = DPOSER_KLOOP_ASM && sizeof(T) == 2 && NB == 4 && KB == 2 && TC == 4 && TS == 2 && WC == 2 && WS == 4;
        constexpr bool ASM_MID = DPOSER_KLOOP_ASM && DPOSER_KLOOP_ASM_MID && sizeof(T) == 2 && NB == 4 && KB == 2 && TC == 2 && TS == 2 && WC == 2 && WS == 2;
        constexpr bool ASM = ASM_BIG || ASM_MID;
        const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
        const uint32_t s_m0 = __builtin_amdgcn_readfirstlane(lds0 + (wave << 10));
        uint32_t v_wofs = lane * 16 + (w_kb << 10), v_xofs = lane * 16 + (seg_kb << 10);
        uint64_t sW[2] = {0, 0}, sX[2] = {0, 0};
        auto x_bases = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int blk = wave + i * C::NW;
                sX[i] = sgpr_u64((uint64_t)(uintptr_t)sbase + ((uint64_t)((int64_t)(sblk * C::ST + blk / KB) * seg_stride + blk % KB) << 10));
            }
        };
        auto advance_asm = [&]() __attribute__((always_inline)) {       // fetch_advance() for the asm address state (offsets advance in the asm)
            seg_kb += KB;
            if (seg_kb >= seg_end && seg + 1 < g.nseg) {
                ++seg;
                seg_kb = 0;
                seg_total = seg_blocks(seg);
                seg_stride = seg_stride_of(seg);
                seg_end = seg_total;
                sbase = seg_ptr(seg);
                x_bases();
                v_xofs = lane * 16;
            }
        };
        if constexpr (ASM) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int blk = wave + i * C::NW;
                sW[i] = sgpr_u64((uint64_t)(uintptr_t)g.W + ((uint64_t)((int64_t)(cblk * C::CT + blk / KB) * g.w_stride_blocks + blk % KB) << 10));
            }
            x_bases();
        }
        auto fetch_w_asm = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) ring_dma_piece(v_wofs, sW[i], s_m0 + buf * C::STAGE_BYTES + ((i * C::NW) << 10));
            v_wofs += KB << 10;
        };
        auto fetch_x_asm = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) ring_dma_piece(v_xofs, sX[i], s_m0 + buf * C::STAGE_BYTES + ((C::CT * KB + i * C::NW) << 10));
            v_xofs += KB << 10;
            advance_asm();
        };
        // The asm stages take their ring slot as a template parameter and run in groups of four: the ring starts at the slot that
        // puts the first group on slot 0 (the `rem` stages in front of it run the C++-scheduled stage with asm-issued DMA).
        const int n_dma = nstages > PRE ? nstages - PRE : 0;          // stages that fetch stage t + PRE
        const int slot0 = ASM ? ((4 - (n_dma & 3)) & 3) : 0;
        if constexpr (ASM) {
            for (int s0 = 0; s0 < npre; ++s0) { fetch_w_asm((slot0 + s0) & 3); fetch_x_asm((slot0 + s0) & 3); }
        } else {
            for (int s0 = 0; s0 < npre; ++s0) fetch_glds(s0);
        }
        if constexpr (PreT::value) Epi::template pre_compute<TC, TS>(ep, pre_state, e_cbase, e_sbase, lane);
        // stage 0 landed?  VMEM returns in order: all but the last npre-1 stages
        if (npre >= 3) __builtin_amdgcn_s_waitcnt(waitcnt_vm(2 * C::LPW));
        else if (npre == 2) __builtin_amdgcn_s_waitcnt(waitcnt_vm(C::LPW));
        else __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
        __syncthreads_lds_only();
        load_frags(slot0, 0, 0);
        int slot = slot0;                  // slot of the stage being computed
        int fill = (slot0 + PRE) % NB;     // slot the stage fetched now goes to (= the slot of stage t-1)
        // DMA: fetch stage t+PRE while computing; ALLOW: DMA pieces that may still be in flight at the barrier
        // (everything issued after stage t+1); LAST: no following stage
        auto stage = [&](auto dma, auto allow, auto last) __attribute__((always_inline)) {
            constexpr bool DMA = decltype(dma)::value, LAST = decltype(last)::value;
            constexpr int ALLOW = decltype(allow)::value;
            const int nslot = (slot + 1 == NB) ? 0 : slot + 1;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                if (kb + 1 < KB) {
                    load_frags(slot, kb + 1, (kb + 1) & 1);
                    if constexpr (DMA && ASM) { if (kb == 0) fetch_w_asm(fill); }
                    else if (DMA && kb == 0) fetch_w(fill);
                    mma(kb & 1);
                    if (kb == 0) pattern(std::integral_constant<int, DMA ? C::LPW_A : 0>{});
                    else pattern(std::integral_constant<int, 0>{});
                } else if constexpr (!LAST) {
                    // stage t+1 has landed for this wave, and this wave is done reading `slot`
                    __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(ALLOW));
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    load_frags(nslot, 0, 0);
                    if constexpr (DMA && ASM) fetch_x_asm(fill);
                    else if constexpr (DMA) { fetch_x(fill); fetch_advance(); }
                    mma(kb & 1);
                    pattern(std::integral_constant<int, DMA ? C::LPW_B : 0>{});
                } else {
                    mma(kb & 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            slot = nslot;
            fill = (fill + 1 == NB) ? 0 : fill + 1;
        };
        typedef std::true_type Y;
        typedef std::false_type N;
        int t = 0;
#ifdef DPOSER_PHASE_STAMPS
        ps_t1 = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (ASM) {
            const uint32_t vA_lo = lds0 + ((wc * TC * KB) << 10) + lane * 16, vB_lo = lds0 + ((C::CT * KB + ws * TS * KB) << 10) + lane * 16;
            const uint32_t vA_hi = vA_lo + 65536, vB_hi = vB_lo + 65536;
#ifdef DPOSER_KLOOP_FILL_INC
            // issue-slot probe (tools/kloop_fill_probe.hip): filler stores go to the scratch g.src[7] -- 4 KiB per wave and stage, a fresh
            // range every stage (HBM traffic like an epilogue's, no line written twice); g.src[6] receives this wave's K-loop cycles
#define DP_RS_FILL_ARGS , fill_off, fill_base, fill_stride
            uint32_t fill_off = (uint32_t)((blockIdx.x * C::NW + wave) * 4096 + lane * 16);
            const uint64_t fill_base = sgpr_u64((uint64_t)(uintptr_t)g.src[7]);
            const uint32_t fill_stride = __builtin_amdgcn_readfirstlane((uint32_t)(gridDim.x * C::NW * 4096));
            const uint64_t stamp0 = __builtin_amdgcn_s_memtime();
#else
#define DP_RS_FILL_ARGS
#endif
#define DP_RING_STAGE(S, MODE)                                                                                                                     \
    if constexpr (ASM_BIG) ring_stage_asm<S, MODE>(acc, fa[0], fb[0], fa[1], fb[1], vA_lo, vB_lo, vA_hi, vB_hi, v_wofs, v_xofs, sW[0], sW[1], sX[0], sX[1], s_m0 DP_RS_FILL_ARGS); \
    else ring_stage_asm_mid<S, MODE>(acc, fa[0], fb[0], fa[1], fb[1], vA_lo, vB_lo, v_wofs, v_xofs, sW[0], sW[1], sX[0], sX[1], s_m0);             \
    ++t
            // the ring starts at slot0 = 4 - rem: the `rem` stages in front of the groups of four run on slots 4 - rem ... 3
            const int rem = n_dma & 3;
            if (rem >= 3) { DP_RING_STAGE(1, 0); advance_asm(); }
            if (rem >= 2) { DP_RING_STAGE(2, 0); advance_asm(); }
            if (rem >= 1) { DP_RING_STAGE(3, 0); advance_asm(); }
            for (int grp = n_dma >> 2; grp > 0; --grp) {        // one loop, one exit (a loop with several exits made hipcc spill the accumulators)
                DP_RING_STAGE(0, 0); advance_asm();
                DP_RING_STAGE(1, 0); advance_asm();
                DP_RING_STAGE(2, 0); advance_asm();
                DP_RING_STAGE(3, 0); advance_asm();
            }
            if (nstages >= 3) {                                 // the three stages without DMA (slot == 0 here)
                DP_RING_STAGE(0, 1);
                DP_RING_STAGE(1, 2);
                DP_RING_STAGE(2, 3);
            }
#undef DP_RING_STAGE
#ifdef DPOSER_KLOOP_FILL_INC
            if (lane == 0) reinterpret_cast<uint64_t*>(const_cast<void*>(g.src[6]))[blockIdx.x * C::NW + wave] = __builtin_amdgcn_s_memtime() - stamp0;
#endif
#undef DP_RS_FILL_ARGS
        } else {
            for (; t + PRE < nstages; ++t) stage(Y{}, std::integral_constant<int, (PRE - 2) * C::LPW + C::LPW_A>{}, N{});
        }
        if (!(ASM && nstages >= 3)) {
            if constexpr (PRE >= 3) {
                if (nstages - t >= 3) { stage(N{}, std::integral_constant<int, C::LPW>{}, N{}); ++t; }
            }
            if (nstages - t >= 2) { stage(N{}, std::integral_constant<int, 0>{}, N{}); ++t; }
            stage(N{}, std::integral_constant<int, 0>{}, Y{});
        }
    }

#ifdef DPOSER_PHASE_STAMPS
    ps_t2 = __builtin_amdgcn_s_memtime();
#endif
#ifdef DPOSER_KLOOP_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    unsigned char* wave_scratch = reinterpret_cast<unsigned char*>(lds_par + NPAR * C::CT * 32) + wave * EpiScratch<Epi>::value;
    if constexpr (EpiRing<Epi>::value > 0 && NB * C::STAGE_BYTES >= C::NW * EpiRing<Epi>::value) {
        __syncthreads_lds_only();      // every wave is done with the ring (no DMA is in flight after the last stage)
        Epi::template apply_ring<TC, TS>(ep, acc, (cblk * C::CT + wc * TC) * 32, ((int64_t)sblk * C::ST + ws * TS) * 32, lane,
                                         sblk * WS + ws, split, lds_par + wc * TC * 32, C::CT * 32, wave_scratch, smem + wave * EpiRing<Epi>::value);
    } else if constexpr (PreT::value) {
        Epi::template apply_pre<TC, TS>(ep, acc, e_cbase, e_sbase, lane, pre_state);
    } else {
        Epi::template apply<TC, TS>(ep, acc, (cblk * C::CT + wc * TC) * 32, ((int64_t)sblk * C::ST + ws * TS) * 32, lane,
                                    sblk * WS + ws, split, lds_par + wc * TC * 32, C::CT * 32, wave_scratch);
    }
#ifdef DPOSER_PHASE_STAMPS
    __builtin_amdgcn_s_waitcnt(0);      // (the epilogue's stores have left the wave)
    if ((tid & 63) == 0) {
        uint64_t* o = reinterpret_cast<uint64_t*>(const_cast<void*>(g.src[6])) + ((size_t)(blockIdx.x + blockIdx.y * gridDim.x) * (WC * WS) + (tid >> 6)) * 6;
        const uint64_t t3 = __builtin_amdgcn_s_memtime();
        o[0] = ps_rt0; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = ps_t1 - ps_t0; o[3] = ps_t2 - ps_t1; o[4] = t3 - ps_t2;
        o[5] = ((uint64_t)__builtin_amdgcn_s_getreg(20 | (3 << 11)) << 32) | __builtin_amdgcn_s_getreg(4 | (31 << 11));      // XCC_ID, HW_ID
    }
#endif
}

template <typename T, int WC, int WS, int TC, int TS, int KB, typename Epi, int NB = 2>
__global__ void __launch_bounds__(WC* WS * 64, (EpiMinWaves<Epi>::value * 256 + WC * WS * 64 - 1) / (WC * WS * 64)) gemm_ft_kernel(GemmArgs g, typename Epi::Params ep) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int ntiles = g.n_cblk * g.n_sblk;
    int L, split;
    const int nblocks = ntiles * (g.ksplit > 1 ? g.ksplit : 1);
    if (g.panel_order == 1 && (nblocks & 7) == 0) {
        // hardware XCD = linear block id % 8, blocks are placed in id order: XCD x takes the logical ids [x n/8, (x+1) n/8), and the
        // sample tiles of one (channel tile, split) are consecutive ids -- resident together, they stream the same W panel, so that
        // panel crosses HBM once and is shared through that XCD's L2 (a pair 30 tiles apart landed on two XCDs: two HBM reads)
        const int lin = blockIdx.x + blockIdx.y * gridDim.x;
        const int q = (lin & 7) * (nblocks >> 3) + (lin >> 3);
        const int r = q / g.n_sblk;
        split = r / g.n_cblk;
        L = (r % g.n_cblk) + (q % g.n_sblk) * g.n_cblk;
    } else if (g.ksplit >= 8 && (g.ksplit & 7) == 0) {
        // split-K (wgrad): hardware XCD = linear block id % 8.  Put a k-range on ONE XCD with all output tiles, so each
        // slice of the two operands is fetched from HBM once and shared through that XCD's L2 by every tile.
        const int lin = blockIdx.x + blockIdx.y * gridDim.x;
        split = lin % g.ksplit;
        L = lin / g.ksplit;
    } else {
        L = xcd_remap(blockIdx.x, ntiles);
        split = blockIdx.y;
    }
    gemm_tile<T, WC, WS, TC, TS, KB, Epi, NB>(g, ep, L % g.n_cblk, L / g.n_cblk, split, smem, (int)threadIdx.x);
}

template <typename T, int WC, int WS, int TC, int TS, int KB, typename Epi, int NB = 2>
static inline hipError_t launch_gemm(const GemmArgs& g, const typename Epi::Params& ep, hipStream_t stream) {
    typedef GemmCfg<T, WC, WS, TC, TS, KB> C;
    auto kern = gemm_ft_kernel<T, WC, WS, TC, TS, KB, Epi, NB>;
    constexpr int lds_bytes = NB * C::STAGE_BYTES + EpiParamArrays<Epi>::value * C::CT * 32 * 4 + EpiScratch<Epi>::value * C::NW;
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    // (function attributes are per device: a process that drives several GPUs sets it once on each)
    static bool attr_set[64] = {};
    int dev = 0;
    if (lds_bytes > 64 * 1024 && hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (lds_bytes > 64 * 1024 && !attr_set[dev & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set[dev & 63] = true;
    }
    dim3 grid(g.n_cblk * g.n_sblk, g.ksplit > 1 ? g.ksplit : 1, 1);
    hipLaunchKernelGGL(kern, grid, dim3(C::THREADS), lds_bytes, stream, g, ep);
    return hipGetLastError();
}
