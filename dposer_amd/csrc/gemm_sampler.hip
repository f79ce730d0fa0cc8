// Persistent Euler-Maruyama sampler: one workgroup owns a block of 256 samples and walks ALL layers of ALL reverse steps.
//
// The reference's sampler (lib/algorithms/advanced/sampling.py:449-468) is N sequential steps of one network evaluation
// + an elementwise update; samples never interact (corrector "none", no observation).  dposer_em_sampler's fast path runs a
// step as 6 dependent launches of 1024 tiles on 256 CUs: every launch ends with all CUs waiting for the slowest tile, starts
// with an empty pipeline on every CU, and sends each layer's activations through HBM.  But layer l + 1 of a sample block needs
// nothing except layer l of the SAME block -- so here a workgroup keeps its block: per step it computes the 4 channel tiles of
// each GroupNorm layer (gemm_tile, the same K loop and epilogue as the launches: results are bit-identical), then post_dense
// + the EM update (EpiEmStep) as two 64 x 128 tiles side by side, and goes on to the next step.  No workgroup ever waits for
// another one, there is no grid-wide join and no cross-workgroup visibility to establish: a workgroup reads only what it wrote
// itself (same CU: an s_waitcnt vmcnt(0) + barrier at each layer boundary is all the ordering needed) and read-only weights /
// bias rows.  Its activations (3 x 512 KB) cycle through L2 / MALL instead of HBM.  Any number of sample blocks may be launched:
// blocks beyond the resident 256 simply start when a CU frees up.
#include "gemm_api.h"
#include "kernels_api.h"

template <typename T>
__global__ void __launch_bounds__(512, 1) k_sampler_persistent(SamplerArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int sblk = blockIdx.x;
    const int tid = threadIdx.x;
    constexpr int KB = 2, NB = 4;
    typedef GemmCfg<T, 1, 4, 2, 1, KB> CF;                       // post_dense tiling: 64 channels x 128 samples, 4 waves
    constexpr int FINAL_GROUP_BYTES = NB * CF::STAGE_BYTES;
    for (int i = 0; i < a.n_steps; ++i) {
        const float* trow = a.table + (int64_t)i * a.L * a.H;
        for (int l = 0; l < a.L; ++l) {
            const SamplerLayer ly = a.layers[l];                  // (device table: wave-uniform scalar loads)
            GemmArgs g;
            g.W = ly.W; g.nseg = 1; g.src[0] = ly.in; g.seg_kblocks[0] = ly.kblocks; g.ktot_blocks = ly.kblocks;
            g.w_stride_blocks = ly.w_stride_blocks; g.n_cblk = a.H / 256; g.n_sblk = gridDim.x; g.ksplit = 1; g.alg_flops = 0.0;
#pragma unroll
            for (int k = 1; k < GEMM_MAX_SEG; ++k) { g.src[k] = nullptr; g.seg_kblocks[k] = 0; }
#pragma unroll
            for (int k = 0; k < GEMM_MAX_SEG; ++k) g.seg_stride_blocks[k] = 0;
            GNParams p;
            p.bias = trow + l * a.H; p.gamma = ly.gamma; p.beta = ly.beta; p.out = ly.out; p.resid = ly.resid; p.xhat = nullptr; p.aux = nullptr;
            p.H = a.H; p.outT = nullptr; p.Spad = a.Spad; p.act = DP_ACT_SWISH; p.out_hi = nullptr; p.out_lo = nullptr;
            p.drop.p = 0.f; p.drop.scale = 1.f; p.drop.thr = 65536; p.drop.site = 0; p.drop.offset = 0; p.drop.seed = 0; p.drop.groups_x4 = a.H / 8; p.drop.ext_keep = nullptr; p.drop.ext_rows = 0;
            // layer boundary: this workgroup's stores of the previous layer (or of the state update) have reached L2 before any of
            // its waves fetches them as the next operand
            __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
            __syncthreads();
            for (int ct = 0; ct < a.H / 256; ++ct) {
                if (ct) __syncthreads();                          // the ring and the staged parameters of the previous tile are free
                gemm_tile<T, 2, 4, 4, 2, KB, EpiGN<T, false>, NB>(g, p, ct, sblk, 0, smem, tid);
            }
        }
        __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
        __syncthreads();
        {
            // post_dense + Euler-Maruyama update: the two half-workgroups take samples [0, 128) and [128, 256) of the block
            GemmArgs g;
            g.W = a.Wpost; g.nseg = 1; g.src[0] = a.last; g.seg_kblocks[0] = a.post_kblocks; g.ktot_blocks = a.post_kblocks;
            g.w_stride_blocks = a.post_kblocks; g.n_cblk = 1; g.n_sblk = 2 * gridDim.x; g.ksplit = 1; g.alg_flops = 0.0;
#pragma unroll
            for (int k = 1; k < GEMM_MAX_SEG; ++k) { g.src[k] = nullptr; g.seg_kblocks[k] = 0; }
#pragma unroll
            for (int k = 0; k < GEMM_MAX_SEG; ++k) g.seg_stride_blocks[k] = 0;
            EmStepParams p = a.em;
            p.t = a.tsteps[i];
            p.step = a.step0 + (uint32_t)i;
            p.x_mean_ft = (i + 1 == a.n_steps) ? a.x_mean_ft : nullptr;
            const int group = tid >> 8;
            gemm_tile<T, 1, 4, 2, 1, KB, EpiEmStep<T>, NB>(g, p, 0, 2 * sblk + group, 0, smem + group * FINAL_GROUP_BYTES, tid & 255);
        }
    }
}

hipError_t launch_sampler_persistent(int prec, const SamplerArgs& a, int64_t n_sample_blocks, hipStream_t st) {
    typedef GemmCfg<__bf16, 2, 4, 4, 2, 2> CM;     // (stage bytes do not depend on the element type: blocks are 1 KiB)
    typedef GemmCfg<__bf16, 1, 4, 2, 1, 2> CF;
    constexpr int lds_main = 4 * CM::STAGE_BYTES + 3 * CM::CT * 32 * 4;
    constexpr int lds_final = 2 * 4 * CF::STAGE_BYTES;
    constexpr int lds_bytes = lds_main > lds_final ? lds_main : lds_final;
    static_assert(lds_bytes <= 160 * 1024, "LDS budget");
    static bool attr_set[2][64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    const void* fn = prec == PREC_FP32 ? reinterpret_cast<const void*>(k_sampler_persistent<float>) : reinterpret_cast<const void*>(k_sampler_persistent<__bf16>);
    if (!attr_set[prec == PREC_FP32][dev & 63]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set[prec == PREC_FP32][dev & 63] = true;
    }
    if (prec == PREC_FP32) hipLaunchKernelGGL(k_sampler_persistent<float>, dim3((unsigned)n_sample_blocks), dim3(512), lds_bytes, st, a);
    else hipLaunchKernelGGL(k_sampler_persistent<__bf16>, dim3((unsigned)n_sample_blocks), dim3(512), lds_bytes, st, a);
    return hipGetLastError();
}

// ---- cluster form ------------------------------------------------------------------------------------------------------------
// What the one-workgroup form pays for is the input panel: its workgroup reads the block's 512 KB of activations once per channel
// tile, four times per layer.  Here FOUR workgroups of ONE XCD (one L2) form a cluster and take one channel tile each of the same
// sample block, so the panel is fetched from that L2 the way the launches' four neighbouring tiles fetch it.  The price is an
// ordering between workgroups: layer l + 1 of a block may start when all four tiles of its layer l are written.  A counter per
// sample block carries that (4 increments per phase; a step = L layer phases + the update phase, which one member runs and
// signs with 4): a member waits for `progress[b] >= 4 * phase`.  A cluster owns every n_clusters-th block and walks its blocks
// layer by layer, so with k blocks per cluster the tile a member waits for was finished k - 1 tiles ago; with one block per
// cluster (<= 16384 samples) the counter is a real 4-way barrier per layer.
//   release: every wave drains its stores (s_waitcnt vmcnt(0): write-through L1, acknowledged by the XCD's L2), barrier, one
//            relaxed agent-scope add.  No L2 write-back: writer and reader share the L2 -- which is why membership is decided by
//            the XCC id a workgroup reads from its own hardware register, not by an assumption about blockIdx.
//   acquire: one lane polls the counter past L1, barrier, every wave invalidates the CU's L1 (agent-scope acquire fence =
//            buffer_inv sc1) before its DMA touches the block.
// Needs every workgroup resident (grid <= 256, one per CU): a poll budget turns a scheduling surprise into an error code, not a hang.
__device__ __forceinline__ int xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15;
}

constexpr uint32_t SAMPLER_POLL_BUDGET = 1u << 21;

template <bool SYNC>
__device__ __forceinline__ bool cluster_wait(const SamplerArgs& a, int b, uint32_t target, int tid, int* s_flag) {
    if constexpr (SYNC) {
        if (tid == 0) {
            uint32_t polls = 0;
            int bad = 0;
            while (__hip_atomic_load(&a.progress[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++polls > SAMPLER_POLL_BUDGET) { bad = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            if (bad) __hip_atomic_store(&a.ctrl[8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (polls > 64) __hip_atomic_fetch_max(&a.ctrl[9], polls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *s_flag = bad;
        }
        __syncthreads();
        if (*s_flag) return false;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    } else {
        __syncthreads();
    }
    return true;
}

template <bool SYNC>
__device__ __forceinline__ void cluster_signal(const SamplerArgs& a, int b, uint32_t inc, int tid) {
    if constexpr (SYNC) {
        __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(&a.progress[b], inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <typename T, bool SYNC>
__global__ void __launch_bounds__(512, 1) k_sampler_cluster(SamplerArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ int s_word[2];
    const int tid = threadIdx.x;
    constexpr int KB = 2, NB = 4;
    typedef GemmCfg<T, 1, 4, 2, 1, KB> CF;
    constexpr int FINAL_GROUP_BYTES = NB * CF::STAGE_BYTES;
    if (tid == 0) {
        const int xcc = xcc_id();
        const int slot = (int)__hip_atomic_fetch_add(&a.ctrl[xcc & 7], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_word[0] = (xcc & 7) | (slot << 8);
    }
    __syncthreads();
    const int word = __builtin_amdgcn_readfirstlane(s_word[0]);
    const int n_clusters = (int)gridDim.x >> 2;
    const int cluster = (word & 7) + 8 * ((word >> 8) >> 2), member = (word >> 8) & 3;
    if (cluster >= n_clusters) {                                  // an XCD got more than its share of the grid: its partner cluster is short
        if (tid == 0) __hip_atomic_store(&a.ctrl[8], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    uint32_t phase = 0;
    for (int i = 0; i < a.n_steps; ++i) {
        const float* trow = a.table + (int64_t)i * a.L * a.H;
        for (int l = 0; l < a.L; ++l, ++phase) {
            const SamplerLayer ly = a.layers[l];
            GemmArgs g;
            g.W = ly.W; g.nseg = 1; g.src[0] = ly.in; g.seg_kblocks[0] = ly.kblocks; g.ktot_blocks = ly.kblocks;
            g.w_stride_blocks = ly.w_stride_blocks; g.n_cblk = a.H / 256; g.n_sblk = a.n_sblk; g.ksplit = 1; g.alg_flops = 0.0;
#pragma unroll
            for (int k = 1; k < GEMM_MAX_SEG; ++k) { g.src[k] = nullptr; g.seg_kblocks[k] = 0; }
#pragma unroll
            for (int k = 0; k < GEMM_MAX_SEG; ++k) g.seg_stride_blocks[k] = 0;
            GNParams p;
            p.bias = trow + l * a.H; p.gamma = ly.gamma; p.beta = ly.beta; p.out = ly.out; p.resid = ly.resid; p.xhat = nullptr; p.aux = nullptr;
            p.H = a.H; p.outT = nullptr; p.Spad = a.Spad; p.act = DP_ACT_SWISH; p.out_hi = nullptr; p.out_lo = nullptr;
            p.drop.p = 0.f; p.drop.scale = 1.f; p.drop.thr = 65536; p.drop.site = 0; p.drop.offset = 0; p.drop.seed = 0; p.drop.groups_x4 = a.H / 8; p.drop.ext_keep = nullptr; p.drop.ext_rows = 0;
            for (int b = cluster; b < a.n_sblk; b += n_clusters) {
                if (!cluster_wait<SYNC>(a, b, 4u * phase, tid, &s_word[1])) return;
                gemm_tile<T, 2, 4, 4, 2, KB, EpiGN<T, false>, NB>(g, p, member, b, 0, smem, tid);
                cluster_signal<SYNC>(a, b, 1u, tid);
            }
        }
        {
            GemmArgs g;
            g.W = a.Wpost; g.nseg = 1; g.src[0] = a.last; g.seg_kblocks[0] = a.post_kblocks; g.ktot_blocks = a.post_kblocks;
            g.w_stride_blocks = a.post_kblocks; g.n_cblk = 1; g.n_sblk = 2 * a.n_sblk; g.ksplit = 1; g.alg_flops = 0.0;
#pragma unroll
            for (int k = 1; k < GEMM_MAX_SEG; ++k) { g.src[k] = nullptr; g.seg_kblocks[k] = 0; }
#pragma unroll
            for (int k = 0; k < GEMM_MAX_SEG; ++k) g.seg_stride_blocks[k] = 0;
            EmStepParams p = a.em;
            p.t = a.tsteps[i];
            p.step = a.step0 + (uint32_t)i;
            p.x_mean_ft = (i + 1 == a.n_steps) ? a.x_mean_ft : nullptr;
            const int group = tid >> 8;
            int bi = 0;
            for (int b = cluster; b < a.n_sblk; b += n_clusters, ++bi) {
                if ((bi & 3) != member) continue;                 // the update of a block: one member, in turn
                if (!cluster_wait<SYNC>(a, b, 4u * phase, tid, &s_word[1])) return;
                gemm_tile<T, 1, 4, 2, 1, KB, EpiEmStep<T>, NB>(g, p, 0, 2 * b + group, 0, smem + group * FINAL_GROUP_BYTES, tid & 255);
                cluster_signal<SYNC>(a, b, 4u, tid);
            }
            ++phase;
        }
    }
}

hipError_t launch_sampler_cluster(int prec, const SamplerArgs& a, int sync, hipStream_t st) {
    typedef GemmCfg<__bf16, 2, 4, 4, 2, 2> CM;
    typedef GemmCfg<__bf16, 1, 4, 2, 1, 2> CF;
    constexpr int lds_main = 4 * CM::STAGE_BYTES + 3 * CM::CT * 32 * 4;
    constexpr int lds_final = 2 * 4 * CF::STAGE_BYTES;
    constexpr int lds_bytes = lds_main > lds_final ? lds_main : lds_final;
    static_assert(lds_bytes + 64 <= 160 * 1024, "LDS budget");
    static bool attr_set[4][64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    const int f32 = prec == PREC_FP32, which = f32 * 2 + (sync ? 1 : 0);
    const void* fns[4] = {reinterpret_cast<const void*>(k_sampler_cluster<__bf16, false>), reinterpret_cast<const void*>(k_sampler_cluster<__bf16, true>),
                          reinterpret_cast<const void*>(k_sampler_cluster<float, false>), reinterpret_cast<const void*>(k_sampler_cluster<float, true>)};
    if (!attr_set[which][dev & 63]) {
        hipError_t e = hipFuncSetAttribute(fns[which], hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
        attr_set[which][dev & 63] = true;
    }
    // clusters: a multiple of 8 (the dispatcher deals workgroups to the 8 XCDs in turn), at most one workgroup per CU
    int n_clusters = (a.n_sblk + 7) / 8 * 8;
    if (n_clusters > 64) n_clusters = 64;
    const dim3 grid((unsigned)n_clusters * 4), block(512);
    if (f32) { if (sync) hipLaunchKernelGGL((k_sampler_cluster<float, true>), grid, block, lds_bytes, st, a); else hipLaunchKernelGGL((k_sampler_cluster<float, false>), grid, block, lds_bytes, st, a); }
    else { if (sync) hipLaunchKernelGGL((k_sampler_cluster<__bf16, true>), grid, block, lds_bytes, st, a); else hipLaunchKernelGGL((k_sampler_cluster<__bf16, false>), grid, block, lds_bytes, st, a); }
    return hipGetLastError();
}
