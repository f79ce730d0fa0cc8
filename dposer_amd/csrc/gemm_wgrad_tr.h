// Weight-gradient GEMM straight from the UN-transposed activations (bf16):
//
//        dW[n][k] = sum_s dY[s][n] * H[s][k]        dY: FT [Spad][N],  H: FT [Spad][Kc]   (both "sample-major")
//
// The plain kernel (gemm.h) wants both operands with the reduction index (here: the sample) contiguous per lane, i.e. the
// transposed copies dY^T [N][Spad] and H^T [Kc][Spad] that the training epilogues used to write next to every activation
// (10 x 128 MB per step at 65536 samples).  This kernel reads the sample-major tiles instead and transposes on the way from
// LDS to the MFMA operand registers with ds_read_b64_tr_b16 (semantics pinned by tools/tr_read_probe.hip: per 16-lane group,
// lane i supplies the address of row i>>2, columns 4(i&3)..+3 of a 4 x 16 block of b16 and receives column i of it).
//
// One stage of the K loop = one 32-sample block row: CT*2 + ST*2 FT blocks of (32 samples x 16 channels).  A 32-channel
// operand fragment needs, per 16-lane group, 4 samples x 16 channels = eight 16-byte chunks of ONE block -- four from each
// 8-channel half, which sit 512 B apart in the FT block, i.e. on the same LDS banks.  The global_load_lds DMA therefore
// PERMUTES the 64 chunks of a block on the way in (the per-lane global address is free): chunk (sample r, half h) goes to
// position ((r>>2)*8 + h*4 + (r&3) + 8*(block&1)) & 63, so that the eight chunks of a group are 128 contiguous bytes and
// the two groups of a 32-lane pass (neighbouring channel blocks) are 32 banks apart: conflict-free, two 2-cycle reads per
// fragment = the cost of the ds_read_b128 of the plain kernel.
// K pipeline, split-K / XCD mapping and the EpiWgrad slab store are the ones of gemm.h (ring of 4 slots).
#pragma once
#include "epilogues.h"
#include "gemm.h"
#include "wgrad_batch.h"

struct WgradTrArgs {
    const void* dY;      // FT [Spad][N]   (rows of dW)
    const void* H;       // FT [Spad][Kc]  (columns of dW)
    int N, Kc;           // padded channel counts (multiples of the tile)
    int n_cblk, n_sblk;  // tiles along N / Kc
    int sblocks;         // Spad / 32
    int ksplit;          // multiple of 8 (or 1); sblocks % ksplit == 0
    double alg_flops;
};

// One output tile over the sample-block rows [sb, sb + nstages): K loop + EpiWgrad store at tile-relative coordinates (en0, ek0) of `ep`.
// dY_ / H_: operand bases, nA / nB: their 16-channel blocks per sample-block row, span: bytes the 32-bit DMA offsets must cover.
template <int WC, int WS, int TC, int TS, int NB>
__device__ __forceinline__ void wgrad_tr_tile(const void* dY_, const void* H_, const int nA, const int nB, const int cblk, const int sblk, int sb,
                                              const int nstages, const int64_t span, const WgradParams& ep, const int split, const int en0,
                                              const int64_t ek0, unsigned char* smem) {
    typedef __bf16 T;
    constexpr int KB = 2;
    typedef GemmCfg<T, WC, WS, TC, TS, KB> C;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    typedef __attribute__((ext_vector_type(4))) short s16x4;
    static_assert(NB >= 3 && NB <= 4, "ring depth");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wc = wave / WS, ws = wave % WS;

    f32x16 acc[TC][TS];
#pragma unroll
    for (int i = 0; i < TC; ++i)
#pragma unroll
        for (int j = 0; j < TS; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // DMA: LDS position `lane` of a block receives the chunk whose un-rotated position is (lane - rot) & 63
    auto src_off = [&](int rot) __attribute__((always_inline)) {
        const int pn = (lane - rot) & 63;
        const int r = (pn >> 3) * 4 + (pn & 3), h = (pn >> 2) & 1;
        return (r + 32 * h) * 16;
    };
    const int off0 = src_off(0), off1 = src_off(8);
    // The DMA is issued from inline asm: hipcc orders every LDS read through an intrinsic (the transposing reads below)
    // behind ALL pending global_load_lds it knows of with an s_waitcnt vmcnt(0), which would serialise the ring; the
    // counted vmcnt waits of the pipeline are explicit anyway.
    auto dma_1k = [&](const unsigned char* src, unsigned char* dst) __attribute__((always_inline)) {
        const unsigned lds = (unsigned)(size_t)(lptr_t)dst;
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds) : "memory");      // (no "m0" clobber: not honoured -- see common.h, tools/check_isa.py)
    };
    auto fetch_a = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < C::LPW_A; ++i) {
            const int blk = wave + i * C::NW;                   // 16-channel block inside the tile
            const unsigned char* p = reinterpret_cast<const unsigned char*>(dY_) + (((int64_t)sb * nA + cblk * C::CT * 2 + blk) << 10);
            dma_1k(p + ((blk & 1) ? off1 : off0), smem + slot * C::STAGE_BYTES + (blk << 10));
        }
    };
    auto fetch_b = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < C::LPW_B; ++i) {
            const int blk = wave + i * C::NW;
            const unsigned char* p = reinterpret_cast<const unsigned char*>(H_) + (((int64_t)sb * nB + sblk * C::ST * 2 + blk) << 10);
            dma_1k(p + ((blk & 1) ? off1 : off0), smem + slot * C::STAGE_BYTES + ((C::CT * 2 + blk) << 10));
        }
    };
    // transposing fragment reads
    const int gq = (lane >> 4) & 1, kh = lane >> 5, i16 = lane & 15;
    const int lane_off = ((((i16 & 3) >> 1) * 4 + (i16 >> 2)) * 16) + (i16 & 1) * 8 + gq * 128;
    typedef typename Mma<T>::Frag Frag;
    Frag fa[2][TC], fb[2][TS];
    auto read_frag = [&](const unsigned char* blocks, int tile, int kb) __attribute__((always_inline)) {
        const unsigned char* b = blocks + ((2 * tile + gq) << 10);
        const int base = (4 * kb + 2 * kh) * 128 + lane_off;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b + (base & 1023)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b + ((base + 128) & 1023)));
        union { Frag f; s16x4 h[2]; } u;
        u.h[0] = lo;
        u.h[1] = hi;
        return u.f;
    };
    auto load_frags = [&](int slot, int kb, int set) __attribute__((always_inline)) {
        const unsigned char* a_blocks = smem + slot * C::STAGE_BYTES;
        const unsigned char* b_blocks = a_blocks + ((C::CT * 2) << 10);
#pragma unroll
        for (int i = 0; i < TC; ++i) fa[set][i] = read_frag(a_blocks, wc * TC + i, kb);
#pragma unroll
        for (int j = 0; j < TS; ++j) fb[set][j] = read_frag(b_blocks, ws * TS + j, kb);
    };
    auto mma = [&](int set) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TS; ++j) Mma<T>::run(fa[set][i], fb[set][j], acc[i][j]);
    };
    auto pattern = [&](auto nv) __attribute__((always_inline)) {
        constexpr int NV = decltype(nv)::value, NM = TC * TS, ND = 2 * (TC + TS);   // two transposing reads per fragment
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
            for (int k = (i * ND) / NM; k < ((i + 1) * ND) / NM; ++k) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            if (NV > 0 && (i * NV) / NM != ((i + 1) * NV) / NM) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        }
    };

    constexpr int PRE = NB - 1;
    // Hand-placed K loop (gemm_wgrad_tr_asm.inc, generated by tools/gen_wgrad_asm.py) for the shipped 256x256 tiling: one asm statement
    // (fragments in fixed registers v208 ... v255), DMA addresses = SGPR base per piece + one 32-bit VGPR offset per operand.
    constexpr bool ASM = DPOSER_KLOOP_ASM && WC == 2 && WS == 4 && TC == 4 && TS == 2 && NB == 4;
    if constexpr (ASM) {
        if (nstages >= 3 && span < (int64_t)0xfff00000) {
            const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
            const uint32_t s_m0 = __builtin_amdgcn_readfirstlane(lds0 + (wave << 10));
            const uint32_t strA = __builtin_amdgcn_readfirstlane(nA << 10), strB = __builtin_amdgcn_readfirstlane(nB << 10);
            uint32_t v_aofs = ((wave & 1) ? off1 : off0) + (uint32_t)sb * strA, v_bofs = ((wave & 1) ? off1 : off0) + (uint32_t)sb * strB;
            uint64_t sA[2], sB[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int blk = wave + i * C::NW;
                sA[i] = sgpr_u64((uint64_t)(uintptr_t)dY_ + ((uint64_t)(cblk * C::CT * 2 + blk) << 10));
                sB[i] = sgpr_u64((uint64_t)(uintptr_t)H_ + ((uint64_t)(sblk * C::ST * 2 + blk) << 10));
            }
            const int n_dma = nstages - PRE;
            const uint32_t rem = __builtin_amdgcn_readfirstlane(n_dma & 3);
            uint32_t grp = __builtin_amdgcn_readfirstlane(n_dma >> 2);
            const int slot0 = (4 - (n_dma & 3)) & 3;                      // the groups of four start on slot 0
            for (int s0 = 0; s0 < PRE; ++s0) {
                const uint32_t m = s_m0 + ((slot0 + s0) & 3) * C::STAGE_BYTES;
#pragma unroll
                for (int i = 0; i < 2; ++i) ring_dma_piece(v_aofs, sA[i], m + ((i * C::NW) << 10));
#pragma unroll
                for (int i = 0; i < 2; ++i) ring_dma_piece(v_bofs, sB[i], m + ((C::CT * 2 + i * C::NW) << 10));
                v_aofs += strA;
                v_bofs += strB;
            }
            __builtin_amdgcn_s_waitcnt(waitcnt_vm(2 * C::LPW));
            __syncthreads_lds_only();
            // per-lane addresses of the transposing reads (read_frag below, restated): k-block 0 lo / hi (+128) and k-block 1 lo (+512)
            // from one base; the hi read of k-block 1 wraps inside its 1-KiB block for the lanes with kh = gq = 1
            const int gq_ = (lane >> 4) & 1, kh_ = lane >> 5, i16_ = lane & 15;
            const int base0 = 2 * kh_ * 128 + ((((i16_ & 3) >> 1) * 4 + (i16_ >> 2)) * 16) + (i16_ & 1) * 8 + gq_ * 128;
            const uint32_t vA0_lo = lds0 + wc * TC * 2048 + gq_ * 1024 + base0, vA1_lo = lds0 + wc * TC * 2048 + gq_ * 1024 + ((base0 + 640) & 1023);
            const uint32_t vB0_lo = lds0 + ((C::CT * 2) << 10) + ws * TS * 2048 + gq_ * 1024 + base0;
            const uint32_t vB1_lo = lds0 + ((C::CT * 2) << 10) + ws * TS * 2048 + gq_ * 1024 + ((base0 + 640) & 1023);
            const uint32_t vA0_hi = vA0_lo + 65536, vA1_hi = vA1_lo + 65536, vB0_hi = vB0_lo + 65536, vB1_hi = vB1_lo + 65536;
#include "gemm_wgrad_tr_asm.inc"
            EpiWgrad<T>::template apply<TC, TS>(ep, acc, en0 + wc * TC * 32, ek0 + ws * TS * 32, lane, 0, split, nullptr, 0, nullptr);
            return;
        }
    }
    const int npre = nstages < PRE ? nstages : PRE;
    for (int s0 = 0; s0 < npre; ++s0) { fetch_a(s0); fetch_b(s0); ++sb; }
    if (npre >= 3) __builtin_amdgcn_s_waitcnt(waitcnt_vm(2 * C::LPW));
    else if (npre == 2) __builtin_amdgcn_s_waitcnt(waitcnt_vm(C::LPW));
    else __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
    __syncthreads_lds_only();
    load_frags(0, 0, 0);
    int slot = 0, fill = PRE;
    auto stage = [&](auto dma, auto allow, auto last) __attribute__((always_inline)) {
        constexpr bool DMA = decltype(dma)::value, LAST = decltype(last)::value;
        constexpr int ALLOW = decltype(allow)::value;
        const int nslot = (slot + 1 == NB) ? 0 : slot + 1;
        load_frags(slot, 1, 1);
        if constexpr (DMA) fetch_a(fill);
        mma(0);
        pattern(std::integral_constant<int, DMA ? C::LPW_A : 0>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!LAST) {
            __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(ALLOW));
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            load_frags(nslot, 0, 0);
            if constexpr (DMA) { fetch_b(fill); ++sb; }
            mma(1);
            pattern(std::integral_constant<int, DMA ? C::LPW_B : 0>{});
        } else {
            mma(1);
        }
        __builtin_amdgcn_sched_barrier(0);
        slot = nslot;
        fill = (fill + 1 == NB) ? 0 : fill + 1;
    };
    typedef std::true_type Y;
    typedef std::false_type N_;
    int t = 0;
    for (; t + PRE < nstages; ++t) stage(Y{}, std::integral_constant<int, (PRE - 2) * C::LPW + C::LPW_A>{}, N_{});
    if constexpr (PRE >= 3) {
        if (nstages - t >= 3) { stage(N_{}, std::integral_constant<int, C::LPW>{}, N_{}); ++t; }
    }
    if (nstages - t >= 2) { stage(N_{}, std::integral_constant<int, 0>{}, N_{}); ++t; }
    stage(N_{}, std::integral_constant<int, 0>{}, Y{});

    EpiWgrad<T>::template apply<TC, TS>(ep, acc, en0 + wc * TC * 32, ek0 + ws * TS * 32, lane, 0, split, nullptr, 0, nullptr);
}

template <int WC, int WS, int TC, int TS, int NB>
__global__ void __launch_bounds__(WC* WS * 64, (2 * 256 + WC * WS * 64 - 1) / (WC * WS * 64)) gemm_wgrad_tr_kernel(WgradTrArgs g, WgradParams ep) {
    typedef GemmCfg<__bf16, WC, WS, TC, TS, 2> C;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int L, split;
    if (g.ksplit >= 8 && (g.ksplit & 7) == 0) {       // one k-range per XCD with all of its tiles (see gemm.h)
        const int lin = blockIdx.x + blockIdx.y * gridDim.x;
        split = lin % g.ksplit;
        L = lin / g.ksplit;
    } else {
        L = xcd_remap(blockIdx.x, g.n_cblk * g.n_sblk);
        split = blockIdx.y;
    }
    const int cblk = L % g.n_cblk, sblk = L / g.n_cblk;
    const int nstages = g.sblocks / (g.ksplit > 1 ? g.ksplit : 1);
    const int nA = g.N >> 4, nB = g.Kc >> 4;                    // 16-channel blocks per sample-block row
    const int64_t span = ((int64_t)g.sblocks * (nA > nB ? nA : nB)) << 10;                // the 32-bit DMA offsets cover the operand
    wgrad_tr_tile<WC, WS, TC, TS, NB>(g.dY, g.H, nA, nB, cblk, sblk, split * nstages, nstages, span, ep, split, cblk * C::CT * 32,
                                      (int64_t)sblk * C::ST * 32, smem);
}

// ---- all 256 x 256 weight-gradient tiles of a training step as ONE launch (wgrad_batch.h) ---------------------------------------
template <int WC, int WS, int TC, int TS, int NB>
__global__ void __launch_bounds__(WC* WS * 64, (2 * 256 + WC * WS * 64 - 1) / (WC * WS * 64)) gemm_wgrad_tr_batch_kernel(WgradBatchArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x;
    const int lane_id = wgb_lane(b), slot = wgb_slot(b);
    const int lo = lane_id * a.q, hi = lo + a.q;
    int start = 0, ord = 0;
    for (int p = 0; p < a.nprob; ++p) {
        const WgradLaneProblem& pr = a.prob[p];
        const int end = start + pr.len;
        const int s0 = lo > start ? lo : start, s1 = hi < end ? hi : end;
        if (s0 < s1) {
            WgradParams ep;
            ep.slab = a.partials + ((int64_t)(b * WGB_MAX_SEG + ord) << 16); ep.slab_stride = 0; ep.ld = 256; ep.N_valid = 256; ep.K_valid = 256;
            int half, cblk, sblk, sboff;
            if (pr.mode == 1) { half = 0; cblk = slot & 1; sblk = (slot >> 1) & 1; sboff = (slot >> 2) * pr.len; }
            else { half = slot >> 3; cblk = slot & 3; sblk = pr.sblk0[half] + ((slot & 7) >> 2); sboff = pr.sb_off[half]; }
            wgrad_tr_tile<WC, WS, TC, TS, NB>(pr.dY[half], pr.H[half], pr.nA[half], pr.nB[half], cblk, sblk, s0 - start + sboff, s1 - s0, a.span, ep,
                                              0, 0, 0, smem);
            ++ord;
            __syncthreads();                                       // the ring is reused by the next segment
        }
        start = end;
    }
}
template <int WC, int WS, int TC, int TS, int NB>
static inline hipError_t launch_wgrad_tr_batch(const WgradBatchArgs& a, hipStream_t stream) {
    typedef GemmCfg<__bf16, WC, WS, TC, TS, 2> C;
    auto kern = gemm_wgrad_tr_batch_kernel<WC, WS, TC, TS, NB>;
    constexpr int lds_bytes = NB * C::STAGE_BYTES;
    static bool attr_set[64] = {};
    int dev = 0;
    if (lds_bytes > 64 * 1024 && hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (lds_bytes > 64 * 1024 && !attr_set[dev & 63]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set[dev & 63] = true;
    }
    hipLaunchKernelGGL(kern, dim3(WGB_BLOCKS), dim3(C::THREADS), lds_bytes, stream, a);
    return hipGetLastError();
}

template <int WC, int WS, int TC, int TS, int NB>
static inline hipError_t launch_wgrad_tr(const WgradTrArgs& g, const WgradParams& ep, hipStream_t stream) {
    typedef GemmCfg<__bf16, WC, WS, TC, TS, 2> C;
    auto kern = gemm_wgrad_tr_kernel<WC, WS, TC, TS, NB>;
    constexpr int lds_bytes = NB * C::STAGE_BYTES;
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
