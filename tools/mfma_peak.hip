// Achievable dense bf16 MFMA rate on this device (sustained clock under matrix load), no memory traffic:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/bin/mfma_peak && tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// MODE 0: small-integer operands (few mantissa bits set); 1: random full-mantissa operands in (-0.1, 0.1), fixed;
// 2: two random operand sets alternating from MFMA to MFMA (what a GEMM's K loop presents to the pipe).
// Power, and with it the sustained clock, depends on the operand data: the rate a real GEMM can reach is mode 2's.
__device__ unsigned long long g_ticks[4096 * 4];
__device__ inline float rnd(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return ((x >> 8) * (1.f / 16777216.f) - 0.5f) * 0.2f; }
template <int NACC, int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    bf16x8 a, b, a2, b2;
    for (int i = 0; i < 8; ++i) {
        if (MODE == 0) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); a2[i] = a[i]; b2[i] = b[i]; }
        else {
            const unsigned id = (blockIdx.x * 256 + threadIdx.x) * 32 + i;
            a[i] = (__bf16)rnd(id); b[i] = (__bf16)rnd(id + 8); a2[i] = (__bf16)rnd(id + 16); b2[i] = (__bf16)rnd(id + 24);
        }
    }
    f32x16 acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int n = 0; n < NACC; ++n) {
            if (MODE == 2 && (n & 1)) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[n], 0, 0, 0);
            else acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
        }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) g_ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float s = 0.f;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int MODE> void run(int blocks, int iters, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, MODE><<<blocks, 256>>>(d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, MODE><<<blocks, 256>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * NACC * 2.0 * 32 * 32 * 16;
    static unsigned long long h[4096 * 4];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ticks), sizeof(unsigned long long) * blocks * 4);
    double tk = 0;
    for (int i = 0; i < blocks * 4; ++i) tk += h[i];
    tk /= blocks * 4;
    const char* names[] = {"integer operands", "random operands", "random, alternating"};
    printf("%-20s acc/wave %d  blocks %5d (waves/SIMD %d)  iters %6d : %8.3f ms  %7.1f TFLOP/s   s_memtime %.3f ticks/ns, %.1f ticks per MFMA per SIMD\n", names[MODE], NACC, blocks,
           blocks / 256, iters, ms, flops / ms * 1e-9, tk / (ms * 1e6), tk / ((double)iters * NACC * (blocks / 256)));
}

int main() {
    float* d;
    hipMalloc(&d, 4096 * 256 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<4, 0>(256, 20000, d);
        run<8, 0>(256, 10000, d);
        run<8, 0>(512, 10000, d);
        run<8, 0>(512, 100000, d);   // ~25 ms: sustained clocks
        run<8, 1>(512, 10000, d);
        run<8, 1>(512, 100000, d);
        run<8, 2>(512, 10000, d);
        run<8, 2>(512, 100000, d);
    }
    return 0;
}
