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
            p.H = a.H; p.outT = nullptr; p.Spad = a.Spad; p.act = DP_ACT_SWISH;
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
