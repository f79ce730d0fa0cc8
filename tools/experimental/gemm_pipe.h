// EXPERIMENT (not part of libdposer_hip.so; built only by tools/tune_gemm.hip with TUNE_PIPE=1).
// Software-pipelined variant of the MFMA GEMM (gemm.h): the epilogue of tile i-1 runs INSIDE the k-loop of tile i,
// in the same wave.
//
// Status (round 1, B = 65536, K = 1024, one MI355X): results agree with the shipped kernel (3 of 6.7e7 outputs differ by one
// bf16 ulp), hipcc does emit the requested "1 MFMA : N VALU" interleave once the epilogue is branch-free, but it is SLOWER
// than the shipped kernels: plain 141 us (shipped 256x256 / 8 waves: 121), GroupNorm forward 168 us (145), GroupNorm forward
// training 256 us (183) with 4-wave workgroups; with 8-wave workgroups of 2x2 wave tiles (two waves per SIMD, 256 registers each)
// plain 136 us, GroupNorm forward 152 us, training 229 us.  With one wave per SIMD every s_waitcnt of the interleaved epilogue (LDS parameter reads, the
// lane^32 exchange, global loads) also stalls that wave's MFMA issue, and the 1-wave main loop is 17 % slower to begin with.
// With two waves per SIMD the other wave's MFMAs queue at the shared issue port and block this wave's interleaved VALU again
// (the cross-wave exclusion of overlap_probe.hip), so the interleave only pays at one wave per SIMD -- where nothing hides the
// remaining waits.  The training epilogue (~7 VALU per MFMA averaged over a tile) is also simply too heavy to hide completely.
// Kept as the starting point for a hand-scheduled version (explicit prefetch of epilogue operands one stage ahead,
// v_permlane32_swap instead of LDS shuffles).
//
// Why (measured on MI355X, tools/overlap_probe*.hip): VALU work of one wave does not overlap MFMAs of ANOTHER wave on
// the same SIMD (an MFMA waiting for the matrix pipe blocks the vector issue port: MFMA-only wave + VALU-only wave =
// sum of their times), but VALU instructions placed between the MFMAs of the SAME wave are almost free (~6 per
// 32x32x16 MFMA).  A second resident workgroup therefore cannot hide a fused GroupNorm / SiLU / dropout epilogue --
// same-wave interleaving can.
//
// Structure: persistent workgroups (4 waves, one per SIMD, up to 512 registers per lane), each walking a strided list of
// output tiles.  Per tile: zero acc; k-loop (identical to gemm.h: global_load_lds double buffer + fragment double
// buffer); stage t of the k-loop additionally carries sub-tile t of the PREVIOUS tile's epilogue (Epi::sub), and a
// sched_group_barrier pipeline asks the scheduler for "1 MFMA, N VALU" groups.  After the k-loop the accumulators move
// to the "previous" set.  The last tile's epilogue runs un-overlapped.  The first two k-stages of the next tile are
// issued before the last MFMA group of the current one, so the DMA prologue is hidden too.
#pragma once
#include "gemm.h"

#ifndef PIPE_VPM
#define PIPE_VPM 5      // VALU instructions requested behind each MFMA of an epilogue-carrying stage
#endif

template <typename T, int WC, int WS, int TC, int TS, int KB, typename Epi>
__global__ void __launch_bounds__(WC* WS * 64, 1) gemm_ft_pipe_kernel(GemmArgs g, typename Epi::Params ep) {
    static_assert(KB % 2 == 0, "fragment double buffering assumes an even number of k-blocks per stage");
    typedef GemmCfg<T, WC, WS, TC, TS, KB> C;
    typedef typename Mma<T>::Frag Frag;
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int NPH = Epi::kPhases;          // epilogue phases per sub-tile
    constexpr int NSUB = TC * TS * NPH;        // epilogue slices per tile = k-loop stages that carry one
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wc = wave / WS, ws = wave % WS;

    // this workgroup's tiles: hardware XCD x = block id % 8 owns the contiguous logical tiles [base, base + count) (xcd_remap);
    // the workgroup takes i = (block id / 8) + k * (grid / 8), k = 0, 1, ...  (g.ksplit == 1: persistent; 0: one tile only)
    const int ntiles = g.n_cblk * g.n_sblk;
    const int xq = ntiles >> 3, xr = ntiles & 7, xx = blockIdx.x & 7;
    const int base = xx < xr ? xx * (xq + 1) : xr * (xq + 1) + (xx - xr) * xq;
    const int count = xq + (xx < xr ? 1 : 0);
    const int istep = g.ksplit == 1 ? (int)(gridDim.x >> 3) : ntiles;
    int ti = blockIdx.x >> 3;

    constexpr int NPAR = EpiParamArrays<Epi>::value;
    float* lds_par = reinterpret_cast<float*>(smem + 2 * C::STAGE_BYTES);   // [NPAR][CT*32]
    unsigned char* wave_scratch = reinterpret_cast<unsigned char*>(lds_par + NPAR * C::CT * 32) + wave * EpiScratch<Epi>::value;

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
    const int nstages = g.ktot_blocks / KB;

    // ---- DMA state: position of the NEXT stage to fetch, for the tile (f_cblk, f_sblk) -------------
    int f_cblk = 0, f_sblk = 0;
    int seg = 0, seg_kb = 0, w_kb = 0, seg_total = 0;
    const unsigned char* sbase = nullptr;
    auto begin_fetch = [&](int i) __attribute__((always_inline)) {
        const int L = base + i;
        f_cblk = L % g.n_cblk;
        f_sblk = L / g.n_cblk;
        seg = 0; seg_kb = 0; w_kb = 0;
        seg_total = g.seg_kblocks[0];
        sbase = seg_ptr(0);
    };
    auto fetch_glds = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < C::LPW_A; ++i) {
            const int blk = wave + i * C::NW;
            const int rb = blk / KB, kb = blk % KB;
            const unsigned char* p = reinterpret_cast<const unsigned char*>(g.W) +
                                     (((int64_t)(f_cblk * C::CT + rb) * g.w_stride_blocks + w_kb + kb) << 10);
            __builtin_amdgcn_global_load_lds((gptr_t)(p + lane * 16), (lptr_t)(smem + buf * C::STAGE_BYTES + (blk << 10)), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < C::LPW_B; ++i) {
            const int blk = wave + i * C::NW;
            const int rb = blk / KB, kb = blk % KB;
            const unsigned char* p = sbase + (((int64_t)(f_sblk * C::ST + rb) * seg_total + seg_kb + kb) << 10);
            __builtin_amdgcn_global_load_lds((gptr_t)(p + lane * 16), (lptr_t)(smem + buf * C::STAGE_BYTES + ((C::CT * KB + blk) << 10)), 16, 0, 0);
        }
        seg_kb += KB;
        w_kb += KB;
        if (seg_kb >= seg_total && seg + 1 < g.nseg) {
            ++seg;
            seg_kb = 0;
            seg_total = seg_blocks(seg);
            sbase = seg_ptr(seg);
        }
    };

    Frag fa[2][TC], fb[2][TS];
    f32x16 acc[TC][TS], pacc[TC][TS];
    typename Epi::Carry carry;
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

    if (ti >= count) return;
    begin_fetch(ti);
    int cblk = f_cblk, sblk = f_sblk;          // tile being accumulated
    int p_cblk = 0, p_sblk = 0;                // tile whose epilogue is in flight (pacc)
    if constexpr (NPAR > 0) {                  // every tile of a workgroup has the same channel block (host guarantees it)
        for (int i = threadIdx.x; i < NPAR * C::CT * 32; i += C::THREADS) {
            const int a = i / (C::CT * 32), c = i % (C::CT * 32);
            lds_par[i] = Epi::param_array(ep, a)[cblk * C::CT * 32 + c];
        }
    }
    fetch_glds(0);
    if (nstages > 1) fetch_glds(1);
    if (nstages > 1) __builtin_amdgcn_s_waitcnt(waitcnt_vm(C::LPW)); else __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));

    // one k-loop stage of the current tile; SUB >= 0: carries sub-tile SUB of the previous tile's epilogue.  The first KB-1
    // k-blocks form ONE scheduling region together with the epilogue code, with a "1 MFMA : VPM VALU" pipeline request; the last
    // k-block (stage barrier, DMA issue) stays a region of its own.
    auto stage = [&](auto sub_tag, int t, bool has_next, int ti_next) __attribute__((always_inline)) {
        constexpr int SUB = decltype(sub_tag)::value;
        constexpr int VPM = PIPE_VPM;
        const int buf = t & 1;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SUB >= 0) {
            constexpr int ST_ = SUB / NPH, PH_ = NPH == 1 ? -1 : SUB % NPH;
            Epi::template sub<TC, TS, ST_ / TS, ST_ % TS, PH_>(ep, carry, pacc[ST_ / TS][ST_ % TS], (p_cblk * C::CT + wc * TC) * 32,
                                                              ((int64_t)p_sblk * C::ST + ws * TS) * 32, lane, p_sblk * WS + ws, 0,
                                                              lds_par + wc * TC * 32, C::CT * 32, wave_scratch);
        }
#pragma unroll
        for (int kb = 0; kb + 1 < KB; ++kb) {
            load_frags(buf, kb + 1, (kb + 1) & 1);
            if constexpr (SUB < 0) __builtin_amdgcn_sched_barrier(0);
            mma(kb & 1);
            if constexpr (SUB < 0) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (SUB >= 0) {
#pragma unroll
            for (int m = 0; m < (KB - 1) * TC * TS; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (t + 1 < nstages) {
            // stage t+1 must have landed (its DMA was issued one full stage ago); every wave is done reading buf
            __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(0));
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (t + 2 < nstages) fetch_glds(buf);            // refill the buffer we just finished reading
            load_frags(buf ^ 1, 0, 0);
        } else if (has_next) {
            // last k-block of the tile: every fragment is in registers; once all waves are here both LDS buffers are free
            __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(0));
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            begin_fetch(ti_next);
            fetch_glds(0);
            if (nstages > 1) fetch_glds(1);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma((KB - 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
    };
    // run the stages [T0, T1) with compile-time stage index and epilogue sub-tile = stage index
    auto peeled = [&](auto self, auto idx_tag, bool has_next, int ti_next) __attribute__((always_inline)) -> void {
        constexpr int I = decltype(idx_tag)::value;
        if constexpr (I < NSUB) {
            stage(std::integral_constant<int, I>{}, I, has_next, ti_next);
            self(self, std::integral_constant<int, I + 1>{}, has_next, ti_next);
        }
    };
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TS; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };

    // ---- first tile: nothing to overlap with ---------------------------------------------------------
    int ti_next = ti + istep;
    bool has_next = ti_next < count;
    zero_acc();
    __syncthreads_lds_only();
    load_frags(0, 0, 0);
    for (int t = 0; t < nstages; ++t) stage(std::integral_constant<int, -1>{}, t, has_next, ti_next);
    for (;;) {
#pragma unroll
        for (int i = 0; i < TC; ++i)
#pragma unroll
            for (int j = 0; j < TS; ++j) pacc[i][j] = acc[i][j];
        p_cblk = cblk; p_sblk = sblk;
        if (!has_next) break;
        ti = ti_next;
        cblk = f_cblk; sblk = f_sblk;
        ti_next = ti + istep;
        has_next = ti_next < count;
        // the tile's first stages were issued before the previous epilogue's stores: wait for all of them (stores need the L2 ack)
        __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(0));
        zero_acc();
        __syncthreads_lds_only();
        load_frags(0, 0, 0);
        // ---- stages 0 .. NSUB-1 carry the NSUB sub-tiles of the previous tile's epilogue (host guarantees nstages >= NSUB) ----
        peeled(peeled, std::integral_constant<int, 0>{}, has_next, ti_next);
        for (int t = NSUB; t < nstages; ++t) stage(std::integral_constant<int, -1>{}, t, has_next, ti_next);
    }
    Epi::template apply<TC, TS>(ep, pacc, (p_cblk * C::CT + wc * TC) * 32, ((int64_t)p_sblk * C::ST + ws * TS) * 32, lane,
                                p_sblk * WS + ws, 0, lds_par + wc * TC * 32, C::CT * 32, wave_scratch);
}

// Persistent launch when every tile of a workgroup keeps its channel block (so the staged per-channel parameters stay
// valid while the previous tile's epilogue is still in flight); otherwise one tile per workgroup (epilogue not overlapped).
template <typename T, int WC, int WS, int TC, int TS, int KB, typename Epi>
static inline hipError_t launch_gemm_pipe(const GemmArgs& g_in, const typename Epi::Params& ep, hipStream_t stream) {
    typedef GemmCfg<T, WC, WS, TC, TS, KB> C;
    auto kern = gemm_ft_pipe_kernel<T, WC, WS, TC, TS, KB, Epi>;
    constexpr int lds_bytes = 2 * C::STAGE_BYTES + EpiParamArrays<Epi>::value * C::CT * 32 * 4 + EpiScratch<Epi>::value * C::NW;
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    static int resident = 0;
    if (resident == 0) {
        if (lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
            if (e != hipSuccess) return e;
        }
        int per_cu = 0, dev = 0, cus = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, C::THREADS, lds_bytes);
        if (e != hipSuccess) return e;
        if ((e = hipGetDevice(&dev)) != hipSuccess) return e;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
        resident = (per_cu < 1 ? 1 : per_cu) * cus;
        resident -= resident % 8;
        if (resident < 8) resident = 8;
    }
    GemmArgs g = g_in;
    const int ntiles = g.n_cblk * g.n_sblk;
    int grid = ntiles;
    g.ksplit = 0;                                   // one tile per workgroup
    const int nstages = g.ktot_blocks / KB;
    if (ntiles > resident && ((resident >> 3) % g.n_cblk) == 0 && nstages >= TC * TS * Epi::kPhases) {
        grid = resident;
        g.ksplit = 1;                               // persistent
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::THREADS), lds_bytes, stream, g, ep);
    return hipGetLastError();
}
