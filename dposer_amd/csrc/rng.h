// Counter-based RNG for the DPoser kernels: Philox4x32-10 (Salmon et al. 2011; -7 for the dropout decisions) + bit->float maps.
// The contract (counter/key layout, stream ids, float maps) is restated on the CPU in
// oracle/philox.py so tests can inject identical numbers into the oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum : uint32_t {
    STREAM_TRAIN_T = 1,
    STREAM_TRAIN_Z = 2,
    STREAM_EM_NOISE = 3,
    STREAM_IMPUTE_A = 4,
    STREAM_IMPUTE_B = 5,
    STREAM_LANGEVIN = 6,
    STREAM_PRIOR = 7,
    STREAM_DROPOUT0 = 16,
};

struct Philox4 {
    uint32_t v[4];
};

// ROUNDS = 10: the standard generator (t, z, sampler / prior noise).  ROUNDS = 7: the dropout decisions -- Salmon et al. (SC'11,
// "Parallel random numbers: as easy as 1, 2, 3", table 2) find Philox4x32 Crush-resistant from 7 rounds on; 10 is their default with a
// safety margin.  Two calls per lane and 32 x 32 sub-tile make the draw 35-45 % of the training-forward epilogue's VALU time
// (tools/valu_bench.hip: 47 ns per wave and call at 10 rounds, 29 ns at 7), and a keep / drop decision at p = 0.1 asks far less of
// its bits than a Monte-Carlo integrand does.
template <int ROUNDS = 10>
__device__ __forceinline__ Philox4 philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        // one 64-bit product per multiplier: hipcc emits a single v_mad_u64_u32 for it, where __umulhi() + the 32-bit product
        // are a v_mul_hi_u32 and a v_mul_lo_u32 (all three quarter rate): 21 % off a Philox call (tools/valu_bench.hip)
        const uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += W0; k1 += W1;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

__device__ __forceinline__ Philox4 philox_at(uint64_t index, uint32_t stream, uint32_t offset, uint64_t seed) {
    return philox4x32<10>((uint32_t)index, (uint32_t)(index >> 32), stream, offset, (uint32_t)seed, (uint32_t)(seed >> 32));
}
// the dropout streams (STREAM_DROPOUT0 + site): 7 rounds, see above
__device__ __forceinline__ Philox4 philox_at_dropout(uint64_t index, uint32_t stream, uint32_t offset, uint64_t seed) {
    return philox4x32<7>((uint32_t)index, (uint32_t)(index >> 32), stream, offset, (uint32_t)seed, (uint32_t)(seed >> 32));
}

__device__ __forceinline__ float u01_open_low(uint32_t b) { return ((float)(b >> 8) + 1.0f) * 5.9604644775390625e-08f; }  // (0,1]
__device__ __forceinline__ float u01(uint32_t b) { return (float)(b >> 8) * 5.9604644775390625e-08f; }                    // [0,1)

// four N(0,1) from one counter (Box-Muller on (r0,r1) and (r2,r3))
__device__ __forceinline__ void normals4(uint64_t index, uint32_t stream, uint32_t offset, uint64_t seed, float out[4]) {
    Philox4 r = philox_at(index, stream, offset, seed);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float rad = sqrtf(-2.0f * logf(u01_open_low(r.v[2 * h])));
        float ang = 6.283185307179586f * u01(r.v[2 * h + 1]);
        float s, c;
        sincosf(ang, &s, &c);
        out[2 * h] = rad * c;
        out[2 * h + 1] = rad * s;
    }
}
