// Achievable dense bf16 MFMA rate on this device (sustained clock under matrix load), no memory traffic:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak && tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC> void run(int blocks, int iters, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * NACC * 2.0 * 32 * 32 * 16;
    printf("acc/wave %d  blocks %5d (waves/SIMD %d)  iters %d : %8.3f ms  %7.1f TFLOP/s\n", NACC, blocks, blocks / 256, iters, ms, flops / ms * 1e-9);
}

int main() {
    float* d;
    hipMalloc(&d, 4096 * 256 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<4>(256, 20000, d);
        run<8>(256, 10000, d);
        run<4>(512, 20000, d);
        run<8>(512, 10000, d);
        run<8>(512, 100000, d);   // ~0.35 s: sustained clocks
    }
    return 0;
}
