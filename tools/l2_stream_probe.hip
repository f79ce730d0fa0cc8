// Upper bound of the "sample-stationary" training form for <= 8192 poses per rank (EXPERIMENTS.md item 9): a workgroup
// that owns 32 (or 64) samples x ALL 1024 channels of a layer has to stream the WHOLE weight matrix of that layer through its CU --
// 3 MB in the training forward (K = 1024 + 512), 2 MB in the dgrad -- while every other CU streams the same bytes.  Whatever the K loop
// looks like, a layer cannot finish before that stream has: this probe measures it, with nothing else in the kernel.
//   variant "dma": global_load_lds_dwordx4 into a 2 x 64 KB LDS ring (the GEMM kernels' staging path), 256 threads per workgroup;
//   variant "vgpr": plain 16-byte global loads folded into a register (no LDS).
// One workgroup per CU (256) or two (512: the 64-sample / 32-sample forms at 8192 poses... 8192 / 32 = 256 workgroups); every workgroup
// reads the SAME buffer front to back -- the L2 of each XCD serves 32 CUs the same lines.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/l2_stream_probe.hip -o tools/bin/l2_stream_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 64 KB stage = 256 threads x 16 x 16 bytes
__global__ void __launch_bounds__(256) k_stream_dma(const u32x4* __restrict__ w, int stages, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x;
    uint32_t acc = 0;
    for (int s = 0; s < stages; ++s) {
        unsigned char* slot = lds + (s & 1) * 65536;
        const u32x4* src = w + (size_t)s * 4096;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            // one wave writes 64 x 16 contiguous bytes: LDS address = M0 base + lane * 16 (the instruction's own layout)
            const int chunk = i * 4 + (tid >> 6);                     // 1-KiB chunk of the stage this wave fills
            __builtin_amdgcn_global_load_lds((const void*)(src + chunk * 64 + (tid & 63)), (__attribute__((address_space(3))) void*)(slot + chunk * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += *reinterpret_cast<const uint32_t*>(slot + tid * 4);      // (touch the stage: one LDS read per thread)
    }
    if (acc == 0x12345678u) sink[blockIdx.x] = acc;
}
__global__ void __launch_bounds__(256) k_stream_vgpr(const u32x4* __restrict__ w, int stages, uint32_t* sink) {
    const int tid = threadIdx.x;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int s = 0; s < stages; ++s) {
        const u32x4* src = w + (size_t)s * 4096;
        u32x4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = src[i * 256 + tid];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc ^= v[i];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[blockIdx.x] = acc[0];
}

int main() {
    const size_t max_bytes = 4u << 20;
    u32x4* w;
    uint32_t* sink;
    CK(hipMalloc(&w, max_bytes));
    CK(hipMalloc(&sink, 4096 * 4));
    std::vector<uint32_t> h(max_bytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u) | 1u;
    CK(hipMemcpy(w, h.data(), max_bytes, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)k_stream_dma, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("| variant | workgroups | weights streamed per workgroup | us (median of 20) | GB/s per CU | TB/s aggregate L2 -> CU |\n|---|---:|---:|---:|---:|---:|\n");
    for (int variant = 0; variant < 2; ++variant)
        for (int wgs : {256, 512, 128})
            for (int mb : {2, 3}) {
                const int stages = mb * 16;       // 64 KB stages
                std::vector<float> t;
                for (int rep = 0; rep < 25; ++rep) {
                    CK(hipEventRecord(e0));
                    if (variant == 0) hipLaunchKernelGGL(k_stream_dma, dim3(wgs), dim3(256), 131072, 0, w, stages, sink);
                    else hipLaunchKernelGGL(k_stream_vgpr, dim3(wgs), dim3(256), 0, 0, w, stages, sink);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep >= 5) t.push_back(ms * 1e3f);
                }
                std::sort(t.begin(), t.end());
                const double us = t[t.size() / 2];
                const double per_wg = (double)mb * 1048576.0;
                const int per_cu = wgs > 256 ? 2 : 1;
                printf("| %s | %d | %d MB | %.1f | %.0f | %.1f |\n", variant == 0 ? "dma -> LDS" : "vgpr", wgs, mb, us, per_wg * per_cu / us * 1e-3,
                       per_wg * wgs / us * 1e-6);
            }
    return 0;
}
