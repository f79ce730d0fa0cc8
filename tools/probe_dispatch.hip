// Where does the hardware place workgroup b?  (run on the GPU box)
//   hipcc --offload-arch=gfx950 -O2 tools/probe_dispatch.hip -o tools/bin/probe_dispatch && tools/bin/probe_dispatch
// Each 256-thread workgroup (64 KiB LDS => two per CU) records XCC / SE / CU ids and its start tick.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(256) probe(unsigned* out, int spin) {
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID, 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));    // HW_REG_XCC_ID
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        out[blockIdx.x * 4 + 0] = hw;
        out[blockIdx.x * 4 + 1] = xcc;
        out[blockIdx.x * 4 + 2] = (unsigned)(t0 & 0xffffffffu);
        out[blockIdx.x * 4 + 3] = (unsigned)(t0 >> 32);
    }
    smem[threadIdx.x] = 1;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
}

int main() {
    const int nb = 1024;
    unsigned* d;
    hipMalloc(&d, nb * 16);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    probe<<<nb, 256, 65536>>>(d, 20);
    hipDeviceSynchronize();
    probe<<<nb, 256, 65536>>>(d, 20);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nb * 4);
    hipMemcpy(h.data(), d, nb * 16, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull;
    for (int b = 0; b < nb; ++b) { unsigned long long t = ((unsigned long long)h[b * 4 + 3] << 32) | h[b * 4 + 2]; if (t < tmin) tmin = t; }
    for (int b = 0; b < nb; ++b) {
        const unsigned hw = h[b * 4], xcc = h[b * 4 + 1] & 0xf;
        unsigned long long t = ((unsigned long long)h[b * 4 + 3] << 32) | h[b * 4 + 2];
        if (b < 96 || (b >= 256 && b < 300) || (b >= 512 && b < 540))
            printf("b %4d  xcc %u  se %u sh %u cu %2u simd %u wave %u  hw %08x  t %llu\n", b, xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15, hw, t - tmin);
    }
    return 0;
}
