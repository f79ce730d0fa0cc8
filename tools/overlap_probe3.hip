// Does padding the MFMA stream of one wave with s_nop (so that its NEXT MFMA does not sit in the vector issue port waiting for
// the matrix pipe) let a VALU-only wave on the same SIMD run underneath?   (run on the GPU box)
//   hipcc --offload-arch=gfx950 -O3 tools/overlap_probe3.hip -o tools/bin/overlap_probe3 && tools/bin/overlap_probe3
// 512 workgroups x 256 threads = 2 waves per SIMD (workgroups b and b + 256 share a CU, tools/probe_dispatch.hip).
// mode 0: MFMA half only; 1: VALU half only; 2: both.  NOP = s_nop cycles after every MFMA (0, 8, 16, 24, 28).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NOP>
__global__ void __launch_bounds__(256) k(float* out, int iters_m, int iters_v, int mode) {
    const bool first = (blockIdx.x & 256) == 0;
    const bool do_m = first && (mode == 0 || mode == 2);
    const bool do_v = !first && (mode == 1 || mode == 2);
    float s = 0.f;
    if (do_m) {
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
        f32x16 acc[8];
        for (int n = 0; n < 8; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        for (int it = 0; it < iters_m; ++it)
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
                if constexpr (NOP >= 16) asm volatile("s_nop 15");
                if constexpr (NOP % 16 >= 8) asm volatile("s_nop 7");
                if constexpr (NOP % 8 >= 4) asm volatile("s_nop 3");
                __builtin_amdgcn_sched_barrier(0);
            }
        for (int n = 0; n < 8; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    }
    if (do_v) {
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
        for (int it = 0; it < iters_v; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
        for (int i = 0; i < 16; ++i) s += v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NOP> void run(float* d) {
    const int im = 20000;
    for (int iv : {40000, 80000}) {
        float t[3];
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            k<NOP><<<512, 256>>>(d, im, iv, mode);
            hipEventRecord(e0);
            k<NOP><<<512, 256>>>(d, im, iv, mode);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&t[mode], e0, e1);
        }
        printf("s_nop %2d cycles / MFMA, VALU iters %6d:  MFMA alone %7.3f ms   VALU alone %7.3f ms   both %7.3f ms   (sum %7.3f, max %7.3f)\n",
               NOP, iv, t[0], t[1], t[2], t[0] + t[1], t[0] > t[1] ? t[0] : t[1]);
    }
}
int main() {
    float* d;
    hipMalloc(&d, 512 * 256 * 4);
    run<0>(d); run<8>(d); run<16>(d); run<20>(d); run<24>(d); run<28>(d);
    return 0;
}
