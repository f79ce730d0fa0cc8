// VALU micro-benchmarks for the epilogue arithmetic (run on the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Idposer_amd/csrc -Iinclude tools/valu_bench.hip -o tools/bin/valu_bench
// What does a dropout draw (Philox4x32 + decision decode) cost per lane, and which multiply form is cheaper on gfx950:
// v_mul_hi_u32 + v_mul_lo_u32 (what hipcc emits for __umulhi / *) or one v_mad_u64_u32 (what it emits for a 64-bit product)?
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct P4 { uint32_t v[4]; };
template <int ROUNDS, bool MAD64>
__device__ __forceinline__ P4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        if (MAD64) {
            const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
            hi0 = (uint32_t)(p0 >> 32); lo0 = (uint32_t)p0; hi1 = (uint32_t)(p1 >> 32); lo1 = (uint32_t)p1;
        } else {
            hi0 = __umulhi(M0, c0); lo0 = M0 * c0; hi1 = __umulhi(M1, c2); lo1 = M1 * c2;
        }
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    P4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// MODE 0: Philox only (xor of the words); 1: + decode into 16 keep floats and a bit set (epilogues.h dropout_mask16_bits);
// 2: + decode into the bit set only
template <int ROUNDS, bool MAD64, int MODE>
__global__ void __launch_bounds__(512) k_drop(uint32_t* out, uint32_t seed, uint32_t thr, float scale, int iters) {
    const uint32_t tid = blockIdx.x * 512 + threadIdx.x;
    uint32_t accb = 0;
    float accf = 0.f;
    for (int it = 0; it < iters; ++it) {
        uint32_t bits = 0;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const P4 r = philox<ROUNDS, MAD64>(tid * 2 + m, (uint32_t)it, 16, 5, seed, 7);
            if (MODE == 0) { bits ^= r.v[0] ^ r.v[1] ^ r.v[2] ^ r.v[3]; continue; }
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int i0 = 4 * (2 * m + (w >> 1)) + (w & 1) * 2;
                const bool k0 = (r.v[w] & 0xffffu) < thr, k1 = (r.v[w] >> 16) < thr;
                if (MODE == 1) { accf += k0 ? scale : 0.f; accf += k1 ? scale * 0.5f : 0.f; }
                bits |= (k0 ? 1u : 0u) << i0;
                bits |= (k1 ? 1u : 0u) << (i0 + 1);
            }
        }
        accb += bits;
    }
    out[tid & 4095] = accb + __float_as_uint(accf);
}

// SiLU forms: 0 = x * rcp(1 + exp(-x)) (shipped), 1 = scale folded into the reciprocal: x * rcp(c + c * exp(-x))
template <int FORM>
__global__ void __launch_bounds__(512) k_silu(float* out, float c, int iters) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = (threadIdx.x * 16 + i) * 1e-3f - 4.f;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float a = x[i] + it * 1e-4f;
            float y;
            if (FORM == 0) y = a * __builtin_amdgcn_rcpf(1.0f + __expf(-a)) * c;
            else y = a * __builtin_amdgcn_rcpf(__builtin_fmaf(__expf(-a), c, c));
            acc += y;
        }
    }
    out[(blockIdx.x * 512 + threadIdx.x) & 4095] = acc;
}

template <typename F> static double time_us(F launch, int reps = 5) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    launch(); CK(hipDeviceSynchronize());
    std::vector<double> t;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a, 0)); launch(); CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms * 1e3);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    uint32_t* d; CK(hipMalloc(&d, 4096 * 4));
    const int G = 1024, IT = 128;     // 1024 x 8 waves: 4 workgroups per CU, 2 resident at a time (512 threads, 8 waves per SIMD max)
    const uint32_t thr = (uint32_t)(0.9 * 65536.0);
    const double lanes_draws = (double)G * 512 * IT;     // (lane, 16 decisions) draws per launch
#define RUN(NAME, ...) { double us = time_us([&] { hipLaunchKernelGGL((__VA_ARGS__), dim3(G), dim3(512), 0, 0, d, 1234u, thr, 1.111f, IT); }); \
        printf("%-44s %8.1f us   %6.2f ns per wave-draw of 16 decisions (per CU: %.3f us per 8 waves x 8 sub-tiles)\n", NAME, us, us * 1e3 / (lanes_draws / 64) * 256, us / (lanes_draws / 64) * 256 * 64); }
    RUN("philox10 mul_hi/lo, no decode", k_drop<10, false, 0>);
    RUN("philox10 mad_u64,   no decode", k_drop<10, true, 0>);
    RUN("philox7  mad_u64,   no decode", k_drop<7, true, 0>);
    RUN("philox10 mul_hi/lo + keep floats + bits", k_drop<10, false, 1>);
    RUN("philox10 mad_u64   + keep floats + bits", k_drop<10, true, 1>);
    RUN("philox7  mad_u64   + keep floats + bits", k_drop<7, true, 1>);
    RUN("philox10 mad_u64   + bits only", k_drop<10, true, 2>);
    RUN("philox7  mad_u64   + bits only", k_drop<7, true, 2>);
    float* f = reinterpret_cast<float*>(d);
    {
        double u0 = time_us([&] { hipLaunchKernelGGL((k_silu<0>), dim3(G), dim3(512), 0, 0, f, 0.9f, IT); });
        double u1 = time_us([&] { hipLaunchKernelGGL((k_silu<1>), dim3(G), dim3(512), 0, 0, f, 0.9f, IT); });
        printf("silu x16: shipped form * scale %8.1f us, scale folded into rcp %8.1f us\n", u0, u1);
    }
    return 0;
}
