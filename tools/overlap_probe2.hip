// How many VALU instructions hide behind MFMAs issued by the SAME wave?  (companion of overlap_probe.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NV>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
    f32x16 acc[8];
    for (int n = 0; n < 8; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) v[(n * NV + i) & 15] = __builtin_fmaf(v[(n * NV + i) & 15], 1.0001f, 0.5f);
        }
    }
    float s = 0.f;
    for (int n = 0; n < 8; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV> void run(float* d, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV><<<blocks, 256>>>(d, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NV><<<blocks, 256>>>(d, 20000);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("waves/SIMD %d  VALU per MFMA %2d : %8.3f ms   (MFMA-only floor %.3f ms at 32 cyc/MFMA, 2.0 GHz)\n", blocks / 256, NV, ms,
           blocks / 256 * 20000.0 * 8 * 32 / 2.0e9 * 1e3);
}

int main() {
    float* d;
    hipMalloc(&d, 512 * 256 * 4);
    for (int blocks : {256, 512}) {
        run<0>(d, blocks); run<1>(d, blocks); run<2>(d, blocks); run<3>(d, blocks); run<4>(d, blocks); run<5>(d, blocks);
        run<6>(d, blocks); run<7>(d, blocks); run<8>(d, blocks); run<12>(d, blocks);
    }
    return 0;
}
