// Does the MFMA SHAPE change the power-limited matrix rate?  Register-only loops of v_mfma_f32_32x32x16_bf16 (16 accumulator registers per
// instruction) against v_mfma_f32_16x16x32_bf16 (4 accumulator registers, half the FLOPs per instruction: half the accumulator register
// traffic per FLOP) on random full-mantissa operands, ~25 ms each (sustained clocks), interleaved:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o tools/bin/mfma_shape_probe && tools/bin/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__device__ inline float rnd(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return ((x >> 8) * (1.f / 16777216.f) - 0.5f) * 0.2f; }
template <int SHAPE> __global__ void __launch_bounds__(256) k(float* out, int iters) {
    bf16x8 a, b, a2, b2;
    for (int i = 0; i < 8; ++i) {
        const unsigned id = (blockIdx.x * 256 + threadIdx.x) * 32 + i;
        a[i] = (__bf16)rnd(id); b[i] = (__bf16)rnd(id + 8); a2[i] = (__bf16)rnd(id + 16); b2[i] = (__bf16)rnd(id + 24);
    }
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[8];
        for (int n = 0; n < 8; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[n] = (n & 1) ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[n], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
        for (int n = 0; n < 8; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    } else {
        f32x4 acc[32];                      // the same 128 accumulator registers
        for (int n = 0; n < 32; ++n) for (int r = 0; r < 4; ++r) acc[n][r] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int n = 0; n < 32; ++n) acc[n] = (n & 1) ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b2, acc[n], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[n], 0, 0, 0);
        for (int n = 0; n < 32; ++n) for (int r = 0; r < 4; ++r) s += acc[n][r];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int SHAPE> void run(float* d, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<SHAPE><<<512, 256>>>(d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<SHAPE><<<512, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_it = SHAPE == 32 ? 8 * 2.0 * 32 * 32 * 16 : 32 * 2.0 * 16 * 16 * 32;       // equal FLOPs per iteration
    printf("v_mfma_f32_%s_bf16, 2 waves / SIMD, %6d iterations: %8.3f ms  %7.1f TFLOP/s\n", SHAPE == 32 ? "32x32x16" : "16x16x32", iters, ms, 512.0 * 4 * iters * per_it / ms * 1e-9);
}
int main() {
    float* d;
    hipMalloc(&d, 512 * 256 * 4);
    for (int rep = 0; rep < 3; ++rep) { run<32>(d, 100000); run<16>(d, 100000); }
    return 0;
}
