// Do VALU work and MFMA work from DIFFERENT waves of a SIMD overlap?  (run on the GPU box)
//   hipcc --offload-arch=gfx950 -O3 tools/overlap_probe.hip -o tools/bin/overlap_probe && tools/bin/overlap_probe
// 512 workgroups x 256 threads = 2 waves per SIMD.  mode 0: every wave runs MFMAs; 1: every wave runs VALU fmas;
// 2: workgroups alternate (even = MFMA, odd = VALU) so each SIMD holds one wave of each kind; 3/4: only the even / odd half.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void __launch_bounds__(256) k(float* out, int iters_m, int iters_v, int mode, int prio) {
    const bool do_m = mode == 0 || ((mode == 2 || mode == 3) && (blockIdx.x & 256) == 0);
    const bool do_v = mode == 1 || ((mode == 2 || mode == 4) && (blockIdx.x & 256) != 0);
    float s = 0.f;
    if (do_m) {
        if (prio == 2) __builtin_amdgcn_s_setprio(3);
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(threadIdx.x * 3 + i); }
        f32x16 acc[8];
        for (int n = 0; n < 8; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
        for (int it = 0; it < iters_m; ++it)
#pragma unroll
            for (int n = 0; n < 8; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
        for (int n = 0; n < 8; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    }
    if (do_v) {
        if (prio == 1) __builtin_amdgcn_s_setprio(3);
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
        for (int it = 0; it < iters_v; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
        for (int i = 0; i < 16; ++i) s += v[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* d;
    hipMalloc(&d, 512 * 256 * 4);
    const int im = 20000, iv = 80000;       // 160k MFMAs (32 cyc each) vs 1.28M fmas (4 cyc each) per wave: ~5.1M cycles each
    const char* names[] = {"all MFMA", "all VALU", "one MFMA wave + one VALU wave per SIMD", "MFMA half only", "VALU half only"};
    for (int prio = 0; prio < 3; ++prio)
        for (int mode = 0; mode < 5; ++mode) {
            if (prio > 0 && mode != 2) continue;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            k<<<512, 256>>>(d, im, iv, mode, prio);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%-44s prio-mode %d %8.3f ms\n", names[mode], prio, ms);
        }
    return 0;
}
