// Shared device/host helpers for the DPoser MI355X (gfx950) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// ----------------------------------------------------------------------------------------------
// error plumbing for the C ABI (thread-local last error string)
// ----------------------------------------------------------------------------------------------
#include "../../include/dposer_hip.h"   // DPOSER_OK / DPOSER_ERR_* status codes

int dposer_set_error(int code, const std::string& msg);

#define DP_CHECK_ARG(cond, msg)                                                \
    do {                                                                       \
        if (!(cond)) return dposer_set_error(DPOSER_ERR_BAD_ARG, std::string(__func__) + ": " + (msg)); \
    } while (0)

#define DP_CHECK_HIP(expr)                                                     \
    do {                                                                       \
        hipError_t _e = (expr);                                                \
        if (_e != hipSuccess)                                                  \
            return dposer_set_error(DPOSER_ERR_HIP, std::string(__func__) + ": " + #expr + ": " + hipGetErrorString(_e)); \
    } while (0)

#define DP_CHECK_LAUNCH() DP_CHECK_HIP(hipGetLastError())

#define DP_TRY(expr)                   \
    do {                               \
        int _rc = (expr);              \
        if (_rc != DPOSER_OK) return _rc; \
    } while (0)

// roctx range around a library call (SURVEY 5 tracing row): with DPOSER_ROCTX=1 in the environment every compute entry point of the C ABI
// pushes / pops a range named after itself, so a rocprofv3 --marker-trace --kernel-trace run folds the kernel trace by CALL instead of "by
// position in the step".  libroctx64.so is dlopen'ed at the first use (no link-time dependency); off (one predictable branch) otherwise.
void dposer_range_push(const char* name);
void dposer_range_pop();
struct DpRange {
    explicit DpRange(const char* name) { dposer_range_push(name); }
    ~DpRange() { dposer_range_pop(); }
    DpRange(const DpRange&) = delete;
    DpRange& operator=(const DpRange&) = delete;
};
#define DP_RANGE() DpRange _dp_range(__func__)

static inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
static inline int64_t ceil_div(int64_t x, int64_t m) { return (x + m - 1) / m; }

// ----------------------------------------------------------------------------------------------
// Fragment-tiled ("FT") matrix layout -- the HBM layout of every intermediate of the score net.
//
// A logical matrix T[R][K] (R = samples or output channels, K = the reduction dimension of the
// GEMM that CONSUMES it) is stored as 1-KiB blocks, block (rb, kb) at ((rb * K/KBS) + kb) KiB:
//     block = 64 lanes x 16 B, lane l = (kh = l >> 5, r = l & 31)
//     bf16: KBS = 16, lane holds T[32 rb + r][16 kb + 8 kh + 0..7]   (= one v_mfma_f32_32x32x16_bf16 operand)
//     fp32: KBS = 8,  lane holds T[32 rb + r][ 8 kb + 4 kh + 0..3]   (= four v_mfma_f32_32x32x2_f32 operands)
// so a wave fetches an MFMA operand with ONE perfectly coalesced 1-KiB load, the LDS image of a
// tile is the HBM image (global_load_lds friendly, ds_read_b128 conflict-free without swizzle),
// and the epilogue of the producing GEMM (lane = sample, registers = channels in quads of 4)
// stores straight into it.
// ----------------------------------------------------------------------------------------------
template <typename T> struct FT {
    static constexpr int EPL = 16 / sizeof(T);   // elements per lane chunk (8 bf16 / 4 fp32)
    static constexpr int KBS = 2 * EPL;          // reduction elements per block (16 / 8)
    static constexpr int BLOCK_ELEMS = 64 * EPL;
    // element index of T[r][k] for a matrix with K (multiple of KBS) columns
    __host__ __device__ static inline int64_t index(int64_t r, int k, int K) {
        return ((r >> 5) * (int64_t)(K / KBS) + (k / KBS)) * BLOCK_ELEMS + ((((k % KBS) / EPL) << 5) + (r & 31)) * EPL + (k % EPL);
    }
};

__device__ __forceinline__ float bf16_to_f32(__bf16 v) { return (float)v; }

__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    // round-to-nearest-even, NaN preserved
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

// 4 consecutive elements (a "quad") load/store in the storage type
template <typename T> struct Quad;
template <> struct Quad<float> {
    __device__ static inline f32x4 load(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    __device__ static inline void store(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
    __device__ static inline f32x4 round_trip(f32x4 v) { return v; }      // the values as they read back after store()
};
template <> struct Quad<__bf16> {
    __device__ static inline f32x4 load(const __bf16* p) {
        uint2 u = *reinterpret_cast<const uint2*>(p);
        f32x4 v;
        v[0] = __uint_as_float(u.x << 16);
        v[1] = __uint_as_float(u.x & 0xffff0000u);
        v[2] = __uint_as_float(u.y << 16);
        v[3] = __uint_as_float(u.y & 0xffff0000u);
        return v;
    }
    __device__ static inline void store(__bf16* p, f32x4 v) {
        // native casts lower to v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN preserved): 2 instructions per quad
        bf16x4 b = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        *reinterpret_cast<bf16x4*>(p) = b;
    }
    __device__ static inline f32x4 round_trip(f32x4 v) {
        return f32x4{(float)(__bf16)v[0], (float)(__bf16)v[1], (float)(__bf16)v[2], (float)(__bf16)v[3]};
    }
};

// ----------------------------------------------------------------------------------------------
// Whole-tile I/O between a 32x32 accumulator tile (lane = sample, v[4q + r] = channel 8q + 4hi + r of a
// 32-channel group) and the FT layout.  `tile` points at element 0 of FT block (sample block, k-block of
// the group's first channel).  Every access is 16 bytes per lane, lane-linear (1 KiB per instruction):
//   fp32: quad q of lane (j, hi) IS the lane's chunk of k-block q                       -> 4 x b128
//   bf16: a lane owns 8 bytes of each chunk; one v_permlane32_swap per dword regroups the two half-waves so
//         that lane (j, hi) holds the full 16-byte chunk (kh = hi) of k-blocks 0 and 1    -> 2 x b128
// ----------------------------------------------------------------------------------------------
// s_waitcnt immediate for "vmcnt(N) only" on gfx9-family encodings: vmcnt = [15:14|3:0], expcnt [6:4], lgkmcnt [11:8]
constexpr int waitcnt_vm(int n) { return (n & 0xF) | ((n >> 4) << 14) | (0x7 << 4) | (0xF << 8); }
constexpr int waitcnt_vm_lgkm0(int n) { return (n & 0xF) | ((n >> 4) << 14) | (0x7 << 4) | (0x0 << 8); }

template <typename T> struct TileIO;
template <> struct TileIO<float> {
    __device__ static inline void store(float* tile, int lane, const float (&v)[16]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            *reinterpret_cast<f32x4*>(tile + q * 256 + lane * 4) = o;
        }
    }
    __device__ static inline void load(const float* tile, int lane, float (&v)[16]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 o = *reinterpret_cast<const f32x4*>(tile + q * 256 + lane * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * q + r] = o[r];
        }
    }
    // split form: issue the loads early (Raw stays in registers), unpack where the values are needed
    struct Raw { f32x4 q[4]; };
    __device__ static inline void load_raw(const float* tile, int lane, Raw& w) {
#pragma unroll
        for (int q = 0; q < 4; ++q) w.q[q] = *reinterpret_cast<const f32x4*>(tile + q * 256 + lane * 4);
    }
    __device__ static inline void unpack(const Raw& w, float (&v)[16]) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[4 * q + r] = w.q[q][r];
    }
};
template <> struct TileIO<__bf16> {
    __device__ static inline unsigned pack2(float a, float b) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
        bf16x2 t = {(__bf16)a, (__bf16)b};
        return *reinterpret_cast<unsigned*>(&t);
    }
    __device__ static inline void swap_halves(unsigned& a, unsigned& b) {
        // lanes 32-63 of a <-> lanes 0-31 of b   (v_permlane32_swap_b32)
        auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        a = r[0];
        b = r[1];
    }
    __device__ static inline void store(__bf16* tile, int lane, const float (&v)[16]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {               // k-block h of the group: quads 2h, 2h+1
            unsigned a0 = pack2(v[8 * h + 0], v[8 * h + 1]), a1 = pack2(v[8 * h + 2], v[8 * h + 3]);
            unsigned b0 = pack2(v[8 * h + 4], v[8 * h + 5]), b1 = pack2(v[8 * h + 6], v[8 * h + 7]);
            swap_halves(a0, b0);
            swap_halves(a1, b1);
            u32x4 o = {a0, a1, b0, b1};
            *reinterpret_cast<u32x4*>(tile + h * 512 + lane * 8) = o;
        }
    }
    struct Raw { u32x4 h[2]; };
    __device__ static inline void load_raw(const __bf16* tile, int lane, Raw& w) {
#pragma unroll
        for (int h = 0; h < 2; ++h) w.h[h] = *reinterpret_cast<const u32x4*>(tile + h * 512 + lane * 8);
    }
    __device__ static inline void unpack(const Raw& w, float (&v)[16]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned a0 = w.h[h][0], a1 = w.h[h][1], b0 = w.h[h][2], b1 = w.h[h][3];
            swap_halves(a0, b0);
            swap_halves(a1, b1);
            v[8 * h + 0] = __uint_as_float(a0 << 16); v[8 * h + 1] = __uint_as_float(a0 & 0xffff0000u);
            v[8 * h + 2] = __uint_as_float(a1 << 16); v[8 * h + 3] = __uint_as_float(a1 & 0xffff0000u);
            v[8 * h + 4] = __uint_as_float(b0 << 16); v[8 * h + 5] = __uint_as_float(b0 & 0xffff0000u);
            v[8 * h + 6] = __uint_as_float(b1 << 16); v[8 * h + 7] = __uint_as_float(b1 & 0xffff0000u);
        }
    }
    __device__ static inline void load(const __bf16* tile, int lane, float (&v)[16]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4 o = *reinterpret_cast<const u32x4*>(tile + h * 512 + lane * 8);
            unsigned a0 = o[0], a1 = o[1], b0 = o[2], b1 = o[3];
            swap_halves(a0, b0);                    // the regrouping is an involution
            swap_halves(a1, b1);
            v[8 * h + 0] = __uint_as_float(a0 << 16); v[8 * h + 1] = __uint_as_float(a0 & 0xffff0000u);
            v[8 * h + 2] = __uint_as_float(a1 << 16); v[8 * h + 3] = __uint_as_float(a1 & 0xffff0000u);
            v[8 * h + 4] = __uint_as_float(b0 << 16); v[8 * h + 5] = __uint_as_float(b0 & 0xffff0000u);
            v[8 * h + 6] = __uint_as_float(b1 << 16); v[8 * h + 7] = __uint_as_float(b1 & 0xffff0000u);
        }
    }
};
// LDS-DMA issued from inline asm (1 KiB / 256 B per wave instruction; LDS destination = wave-uniform base + lane * size).
// Why asm: beside a global_load_lds it knows of, hipcc drains the whole VMEM queue (s_waitcnt vmcnt(0)) in front of every
// ordinary load result and every LDS read that may alias the DMA target -- which turns a prefetch into a synchronous load.
// Issued from asm the DMA is invisible to that pass; the caller then owns the ordering: a counted s_waitcnt vmcnt(N) before
// the ds_reads of the landed data (VMEM operations of one wave retire in issue order on gfx9-family parts, stores included),
// and an lgkmcnt(0) between the last ds_read of a buffer and the DMA that refills it.
// M0: the statement writes it and reads it itself.  hipcc reserves M0 and does NOT honour an "m0" clobber (it only warns), so none
// is listed; what keeps this safe is that every compiler-issued M0 reader re-materialises M0 first -- audited on the emitted ISA of
// every kernel by tools/check_isa.py (CPU test test_emitted_isa_passes_the_asm_audits).
__device__ __forceinline__ void glds_asm_b128(const void* src_lane, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src_lane), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void glds_asm_b32(const void* src_lane, unsigned lds_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src_lane), "s"(lds_base) : "memory");
}
// s_waitcnt vmcnt(n) for an n that folds to a constant after unrolling (the builtin wants a literal)
__device__ __forceinline__ void wait_vmcnt_n(int n) {
    switch (n) {
#define DP_WVM(N) case N: __builtin_amdgcn_s_waitcnt(waitcnt_vm(N)); break;
        DP_WVM(0) DP_WVM(1) DP_WVM(2) DP_WVM(3) DP_WVM(4) DP_WVM(5) DP_WVM(6) DP_WVM(7) DP_WVM(8) DP_WVM(9) DP_WVM(10) DP_WVM(11)
        DP_WVM(12) DP_WVM(13) DP_WVM(14) DP_WVM(15) DP_WVM(16) DP_WVM(17) DP_WVM(18) DP_WVM(19) DP_WVM(20)
#undef DP_WVM
        default: __builtin_amdgcn_s_waitcnt(waitcnt_vm(0)); break;
    }
}
__device__ __forceinline__ void wait_lgkmcnt0() { __builtin_amdgcn_s_waitcnt(waitcnt_vm_lgkm0(63)); }

// bit i of `bits` as an all-ones / all-zeros word: ONE v_bfe_i32 (sign-extending 1-bit field extract).  Written with the builtin,
// hipcc canonicalises "extract and AND" into v_and (bit test) + v_cmp_ne + v_cndmask -- three VALU instructions per element in the
// dropout epilogues instead of two (v_bfe_i32 + v_and).
template <int I> __device__ __forceinline__ uint32_t bit_mask(uint32_t bits) {
    uint32_t m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(bits), "n"(I));
    return m;
}

// (index that folds to a constant after unrolling)
__device__ __forceinline__ uint32_t bit_mask_rt(uint32_t bits, int i) {
    switch (i) {
#define DP_BM(I) case I: return bit_mask<I>(bits);
        DP_BM(0) DP_BM(1) DP_BM(2) DP_BM(3) DP_BM(4) DP_BM(5) DP_BM(6) DP_BM(7) DP_BM(8) DP_BM(9) DP_BM(10) DP_BM(11) DP_BM(12) DP_BM(13) DP_BM(14) DP_BM(15)
#undef DP_BM
        default: return 0u - ((bits >> i) & 1u);
    }
}

// x + (x of lane ^ 32) with one v_permlane32_swap: the two accumulator halves of a GroupNorm group live in lanes l and l ^ 32.
// (__shfl_xor(x, 32) is a ds_bpermute: an LDS round trip plus an s_waitcnt that stalls the wave -- and its MFMAs -- in an epilogue.)
__device__ __forceinline__ float sum_xor32(float x) {
    const unsigned u = __float_as_uint(x);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);     // r[0]: upper half <- lower half of u; r[1]: lower half <- upper half
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);               // = x + partner in both halves (addition commutes: same bits)
}

// Transposed store of a 32x32 tile: the same values written as FT[channel rows][sample k] (the operand layout of the
// wgrad GEMMs, which reduce over samples).  The tile is transposed through a per-wave LDS scratch (row = channel, row
// stride 80 B / 144 B => conflict-free b128 reads) and leaves as lane-linear 16-byte chunks, 1 KiB per instruction.
//   tileT points at element 0 of FT block (row block c0/32, k-block of sample s0) of the [C][Spad] matrix.
template <typename T> struct TileT;
template <> struct TileT<__bf16> {
    static constexpr int SCRATCH_BYTES = 32 * 80;
    __device__ static inline void store(__bf16* tileT, unsigned char* scratch, int lane, const float (&v)[16]) {
        const int j = lane & 31, hi = lane >> 5;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *reinterpret_cast<__bf16*>(scratch + (8 * q + 4 * hi + r) * 80 + j * 2) = (__bf16)v[4 * q + r];
        // no s_waitcnt: the scratch is wave-private and a wave's LDS instructions execute in issue order, so the reads below see
        // the writes above; the compiler barrier only pins that order
        asm volatile("" ::: "memory");
#pragma unroll
        for (int h = 0; h < 2; ++h) {                             // k-block h = samples 16h .. 16h+15
            u32x4 o = *reinterpret_cast<const u32x4*>(scratch + j * 80 + (16 * h + 8 * hi) * 2);
            *reinterpret_cast<u32x4*>(tileT + h * 512 + lane * 8) = o;
        }
        asm volatile("" ::: "memory");
    }
};
template <> struct TileT<float> {
    static constexpr int SCRATCH_BYTES = 16 * 144;                // two passes of 16 channels keep the BIG fp32 tile inside 160 KB
    __device__ static inline void store(float* tileT, unsigned char* scratch, int lane, const float (&v)[16]) {
        const int j = lane & 31, hi = lane >> 5;
        const int r16 = lane & 15, c4 = lane >> 4;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {                    // channels 16*pass .. 16*pass+15
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    *reinterpret_cast<float*>(scratch + (8 * q2 + 4 * hi + r) * 144 + j * 4) = v[4 * (2 * pass + q2) + r];
            asm volatile("" ::: "memory");                        // (in-order LDS per wave, see the bf16 variant)
#pragma unroll
            for (int half = 0; half < 2; ++half) {                // sample chunk sc = 4 samples; k-block = sc>>1, kh = sc&1
                const int sc = 4 * half + c4;
                f32x4 o = *reinterpret_cast<const f32x4*>(scratch + r16 * 144 + sc * 16);
                *reinterpret_cast<f32x4*>(tileT + (sc >> 1) * 256 + ((sc & 1) * 32 + 16 * pass + r16) * 4) = o;
            }
            asm volatile("" ::: "memory");                        // next pass's writes are issued after these reads: in order
        }
    }
};
// element offset of FT block (row block of channel c0, k-block of sample s0) in the transposed [C][Spad] matrix
template <typename T> __device__ __forceinline__ int64_t ft_tileT_base(int64_t s0, int c0, int64_t Spad) {
    return ((int64_t)(c0 >> 5) * (Spad / FT<T>::KBS) + s0 / FT<T>::KBS) * FT<T>::BLOCK_ELEMS;
}

// element offset of the FT block holding (sample block of s0, k-block of channel c0); s0 % 32 == 0, c0 % 32 == 0
template <typename T> __device__ __forceinline__ int64_t ft_tile_base(int64_t s0, int c0, int K) {
    return ((s0 >> 5) * (int64_t)(K / FT<T>::KBS) + c0 / FT<T>::KBS) * FT<T>::BLOCK_ELEMS;
}

template <typename T> __device__ __forceinline__ T from_f32(float f);
template <> __device__ __forceinline__ float from_f32<float>(float f) { return f; }
template <> __device__ __forceinline__ __bf16 from_f32<__bf16>(float f) { return (__bf16)f; }

// Activations of lib/algorithms/advanced/model.py:54-66 (get_act): the shipped configuration is 'swish' (= SiLU), which the hot
// tilings compile in; 'elu' / 'relu' / 'lrelu' (negative slope 0.2) run through the ACTRT epilogue instantiations (128-wide
// tilings), which pick the function from a wave-uniform runtime code.
enum : int { DP_ACT_SWISH = 0, DP_ACT_ELU = 1, DP_ACT_RELU = 2, DP_ACT_LRELU = 3 };

// precise / fast scalar math selected by the storage type (fp32 mode = parity mode)
template <bool PRECISE> __device__ __forceinline__ float silu_f(float a) {
    if (PRECISE) return a / (1.0f + expf(-a));
    return a * __builtin_amdgcn_rcpf(1.0f + __expf(-a));      // v_exp + v_rcp (1 ulp each): 5 VALU ops
}
// d silu / da
template <bool PRECISE> __device__ __forceinline__ float dsilu_f(float a) {
    float s = PRECISE ? 1.0f / (1.0f + expf(-a)) : __builtin_amdgcn_rcpf(1.0f + __expf(-a));
    return s * (1.0f + a * (1.0f - s));
}

// runtime-selected activation and its derivative (act is wave-uniform)
template <bool PRECISE> __device__ __forceinline__ float act_rt(float a, int act) {
    switch (act) {
        case DP_ACT_ELU: return a > 0.f ? a : (PRECISE ? expm1f(a) : __expf(a) - 1.0f);     // nn.ELU(alpha = 1)
        case DP_ACT_RELU: return fmaxf(a, 0.f);
        case DP_ACT_LRELU: return a > 0.f ? a : 0.2f * a;                                   // nn.LeakyReLU(0.2)
        default: return silu_f<PRECISE>(a);
    }
}
template <bool PRECISE> __device__ __forceinline__ float dact_rt(float a, int act) {
    switch (act) {
        case DP_ACT_ELU: return a > 0.f ? 1.0f : (PRECISE ? expf(a) : __expf(a));
        case DP_ACT_RELU: return a > 0.f ? 1.0f : 0.f;
        case DP_ACT_LRELU: return a > 0.f ? 1.0f : 0.2f;
        default: return dsilu_f<PRECISE>(a);
    }
}
