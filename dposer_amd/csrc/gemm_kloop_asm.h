// Hand-placed steady-state stage of the 256x256 / 8-wave bf16 ring K loop (gemm.h: TC = 4, TS = 2, KB = 2, NB = 4).
//
// Why: hipcc's schedule of the same stage (tools/tune_gemm, ISA of gemm_ft_kernel<bf16,2,4,4,2,2,EpiPlainFT,4>) re-derives the four
// DMA addresses of a stage with s_mul / 64-bit add chains (~45 SALU + 8 VALU per 16 MFMAs) and puts ~40 non-MFMA instructions
// (address arithmetic, s_waitcnt, s_barrier, six ds_reads, two DMAs) between the last MFMA in front of the stage barrier and the
// first one behind it: both waves of a SIMD reach that stretch together (the barrier keeps them in phase), so the matrix pipe idles
// there.  rocprofv3 counters on the 65536 x 1024 x 4096 problem: MFMA busy 0.66 at 1.81 GHz for that loop against 0.79 at 1.76 GHz for
// the vendor library's hand-scheduled kernel (profiles/r03_vendor_pmc_K4096.md) -- the gap is cycles, not clock.
//
// This stage: one asm statement, 16 MFMAs with at most three other instructions between two of them;
//   * DMA addresses = loop-invariant 64-bit SGPR base per piece + one 32-bit VGPR offset per operand that advances 2 KiB per stage
//     (global_load_lds saddr form): 2 VALU + 4 SALU per stage for all addressing;
//   * the ring slot is a template parameter, so LDS read offsets and the M0 values are immediates;
//   * s_waitcnt / s_barrier sit between two MFMAs, the next MFMA follows the barrier at once (its operands were read from LDS
//     during the first half of the stage).
// Contract (identical to the C++ `stage(Y, 6, N)` it replaces; same MFMA order per accumulator => bit-identical results):
//   in : f0a / f0b = fragments of k-block 0 of this stage (slot S), landed;
//   out: f0a / f0b = fragments of k-block 0 of the next stage (slot S+1), landed (lgkmcnt(0) inside); f1a / f1b scratch;
//   DMA: the two weight pieces (before the barrier) and the two activation pieces (after it) of stage t+3 into slot S+3;
//   wait: vmcnt(6) in front of the barrier = everything issued before {stage t+2's four pieces, this stage's two weight pieces}.
#pragma once
#include "common.h"
#ifndef DPOSER_KLOOP_VARIANT
#define DPOSER_KLOOP_VARIANT 0
#endif

// LDS byte offsets relative to the wave's A / B fragment base of slots {0,1} (lo) or {2,3} (hi): the ds_read offset field is 16 bits
template <int S> struct RingSlot {
    static constexpr int kRel = (S & 1) * 32768;       // relative to the lo / hi base
    static constexpr bool kHi = S >= 2;
};

// Stage text.  W0 / W1 / WADD / X0 / X1 / XADD: the DMA lines of a fetching stage ("" in the tail stages); WAIT: the s_waitcnt in front of
// the barrier; the LAST stage has no barrier and reads no following k-block.
// Probe hooks (tools/kloop_fill_probe.hip: how many issue slots per MFMA gap does THIS stage leave?).  DP_RS_Gi = text behind MFMA i of the
// stage; empty in the library -- the shipped statement is the text below and nothing else.
#ifdef DPOSER_KLOOP_FILL_INC
#include DPOSER_KLOOP_FILL_INC
#define DP_RS_FILL_PARAMS , uint32_t& fill_off, uint64_t fill_base, uint32_t fill_stride
#else
#define DP_RS_G0 ""
#define DP_RS_G1 ""
#define DP_RS_G2 ""
#define DP_RS_G3 ""
#define DP_RS_G4 ""
#define DP_RS_G5 ""
#define DP_RS_G6 ""
#define DP_RS_G7 ""
#define DP_RS_G8 ""
#define DP_RS_G9 ""
#define DP_RS_G10 ""
#define DP_RS_G11 ""
#define DP_RS_G12 ""
#define DP_RS_G13 ""
#define DP_RS_G14 ""
#define DP_RS_G15 ""
#define DP_RS_VMCNT_MAIN "6"
#define DP_RS_FILL_CLOBBER
#define DP_RS_FILL_INOUT
#define DP_RS_FILL_IN
#define DP_RS_FILL_PARAMS
#endif
#define DP_RS_MFMA(c, a, b) "v_mfma_f32_32x32x16_bf16 %[" #c "], %[" #a "], %[" #b "], %[" #c "]\n"
#define DP_RS_READ(dst, base, off) "ds_read_b128 %[" #dst "], %[" #base "] offset:%[" #off "]\n"
#define DP_RS_HALF1(W0S, W0L, W1S, W1L, WADD, WAIT, BARRIER)                                                      \
    DP_RS_MFMA(c00, a00, b00) DP_RS_READ(a10, vA1, r1a0) DP_RS_READ(b10, vB1, r1b0) DP_RS_G0                       \
    DP_RS_MFMA(c01, a00, b01) DP_RS_READ(b11, vB1, r1b1) W0S DP_RS_G1                                              \
    DP_RS_MFMA(c10, a01, b00) DP_RS_READ(a11, vA1, r1a1) W0L DP_RS_G2                                              \
    DP_RS_MFMA(c11, a01, b01) DP_RS_READ(a12, vA1, r1a2) W1S DP_RS_G3                                              \
    DP_RS_MFMA(c20, a02, b00) DP_RS_READ(a13, vA1, r1a3) W1L DP_RS_G4                                              \
    DP_RS_MFMA(c21, a02, b01) WADD DP_RS_G5                                                                        \
    DP_RS_MFMA(c30, a03, b00) DP_RS_G6 WAIT                                                                        \
    DP_RS_MFMA(c31, a03, b01) DP_RS_G7 BARRIER
#define DP_RS_HALF2(X0S, X0L, X1S, X1L, XADD)                                                                      \
    DP_RS_MFMA(c00, a10, b10) DP_RS_READ(a00, vA0, r0a0) DP_RS_READ(b00, vB0, r0b0) DP_RS_G8                       \
    DP_RS_MFMA(c01, a10, b11) DP_RS_READ(b01, vB0, r0b1) DP_RS_G9 X0S                                              \
    DP_RS_MFMA(c10, a11, b10) DP_RS_READ(a01, vA0, r0a1) X0L DP_RS_G10                                             \
    DP_RS_MFMA(c11, a11, b11) DP_RS_READ(a02, vA0, r0a2) X1S DP_RS_G11                                             \
    DP_RS_MFMA(c20, a12, b10) DP_RS_READ(a03, vA0, r0a3) X1L DP_RS_G12                                             \
    DP_RS_MFMA(c21, a12, b11) XADD DP_RS_G13                                                                       \
    DP_RS_MFMA(c30, a13, b10) DP_RS_G14                                                                            \
    DP_RS_MFMA(c31, a13, b11) DP_RS_G15 "s_waitcnt lgkmcnt(0)\n"
// The compiler does not know that the statement ends on an MFMA: the wait states between an 8-pass XDL write and a VALU read of its result
// (11 on gfx940-family parts; the hazard recognizer inserts them for MFMAs it sees) are spent inside the LAST stage's statement.
#define DP_RS_XDL_DRAIN "s_nop 7\n" "s_nop 7\n"
#define DP_RS_HALF2_LAST                                                                                           \
    DP_RS_MFMA(c00, a10, b10) DP_RS_MFMA(c01, a10, b11) DP_RS_MFMA(c10, a11, b10) DP_RS_MFMA(c11, a11, b11)        \
    DP_RS_MFMA(c20, a12, b10) DP_RS_MFMA(c21, a12, b11) DP_RS_MFMA(c30, a13, b10) DP_RS_MFMA(c31, a13, b11) DP_RS_XDL_DRAIN
#define DP_RS_M0(m) "s_add_i32 m0, %[sm0], %[" #m "]\n"       // (an MFMA and a ds_read sit between the M0 write and the DMA that uses it)
#define DP_RS_DMA(v, s) "global_load_lds_dwordx4 %[" #v "], %[" #s "]\n"
#define DP_RS_OPERANDS                                                                                                                 \
        : [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][1]), [c10] "+v"(acc[1][0]), [c11] "+v"(acc[1][1]), [c20] "+v"(acc[2][0]),           \
          [c21] "+v"(acc[2][1]), [c30] "+v"(acc[3][0]), [c31] "+v"(acc[3][1]),                                                         \
          [a00] "+v"(f0a[0]), [a01] "+v"(f0a[1]), [a02] "+v"(f0a[2]), [a03] "+v"(f0a[3]), [b00] "+v"(f0b[0]), [b01] "+v"(f0b[1]),       \
          [a10] "=&v"(f1a[0]), [a11] "=&v"(f1a[1]), [a12] "=&v"(f1a[2]), [a13] "=&v"(f1a[3]), [b10] "=&v"(f1b[0]), [b11] "=&v"(f1b[1]), \
          [vw] "+v"(v_wofs), [vx] "+v"(v_xofs) DP_RS_FILL_INOUT                                                                        \
        : [vA1] "v"(vA1), [vB1] "v"(vB1), [vA0] "v"(vA0), [vB0] "v"(vB0), [sW0] "s"(sW0), [sW1] "s"(sW1), [sX0] "s"(sX0), [sX1] "s"(sX1), \
          [sm0] "s"(s_m0),                                                                                                             \
          [r1a0] "n"(R1), [r1a1] "n"(R1 + 2048), [r1a2] "n"(R1 + 4096), [r1a3] "n"(R1 + 6144), [r1b0] "n"(R1), [r1b1] "n"(R1 + 2048),  \
          [r0a0] "n"(R0), [r0a1] "n"(R0 + 2048), [r0a2] "n"(R0 + 4096), [r0a3] "n"(R0 + 6144), [r0b0] "n"(R0), [r0b1] "n"(R0 + 2048),  \
          [mw0] "n"(M), [mw1] "n"(M + 8192), [mx0] "n"(M + 16384), [mx1] "n"(M + 24576) DP_RS_FILL_IN                                  \
        : "memory", "scc" DP_RS_FILL_CLOBBER

// MODE 0: fetching stage (the DMA lands in slot S+3; vmcnt(6) in front of the barrier); 1 / 2: third-last / second-last stage (no DMA;
// vmcnt(4) / vmcnt(0): the pieces of the stages behind t+1 may still fly); 3: last stage (no barrier, no following k-block).
template <int S, int MODE = 0>
__device__ __forceinline__ void ring_stage_asm(f32x16 (&acc)[4][2], bf16x8 (&f0a)[4], bf16x8 (&f0b)[2], bf16x8 (&f1a)[4], bf16x8 (&f1b)[2],
                                               uint32_t vA_lo, uint32_t vB_lo, uint32_t vA_hi, uint32_t vB_hi, uint32_t& v_wofs,
                                               uint32_t& v_xofs, uint64_t sW0, uint64_t sW1, uint64_t sX0, uint64_t sX1, uint32_t s_m0 DP_RS_FILL_PARAMS) {
    constexpr int S1 = (S + 1) & 3, D = (S + 3) & 3;
    // k-block 1 of slot S (first half), k-block 0 of slot S1 (second half)
    const uint32_t vA1 = RingSlot<S>::kHi ? vA_hi : vA_lo, vB1 = RingSlot<S>::kHi ? vB_hi : vB_lo;
    const uint32_t vA0 = RingSlot<S1>::kHi ? vA_hi : vA_lo, vB0 = RingSlot<S1>::kHi ? vB_hi : vB_lo;
    constexpr int R1 = RingSlot<S>::kRel + 1024, R0 = RingSlot<S1>::kRel;        // fragment i of k-block kb sits at (i * 2 + kb) KiB
    constexpr int M = D * 32768;                                                  // DMA target slot; pieces: W0, W1 = +0, +8 KiB; X0, X1 = +16, +24 KiB
    if constexpr (MODE == 0) {
#if DPOSER_KLOOP_VARIANT == 1      // activations (the HBM-sourced operand) first: three full stages of lead instead of two and a half
        asm volatile(DP_RS_HALF1(DP_RS_M0(mx0), DP_RS_DMA(vx, sX0), DP_RS_M0(mx1), DP_RS_DMA(vx, sX1), "v_add_u32 %[vx], 0x800, %[vx]\n", "s_waitcnt vmcnt(" DP_RS_VMCNT_MAIN ") lgkmcnt(0)\n", "s_barrier\n")
                     DP_RS_HALF2(DP_RS_M0(mw0), DP_RS_DMA(vw, sW0), DP_RS_M0(mw1), DP_RS_DMA(vw, sW1), "v_add_u32 %[vw], 0x800, %[vw]\n") DP_RS_OPERANDS);
#else
        asm volatile(DP_RS_HALF1(DP_RS_M0(mw0), DP_RS_DMA(vw, sW0), DP_RS_M0(mw1), DP_RS_DMA(vw, sW1), "v_add_u32 %[vw], 0x800, %[vw]\n", "s_waitcnt vmcnt(" DP_RS_VMCNT_MAIN ") lgkmcnt(0)\n", "s_barrier\n")
                     DP_RS_HALF2(DP_RS_M0(mx0), DP_RS_DMA(vx, sX0), DP_RS_M0(mx1), DP_RS_DMA(vx, sX1), "v_add_u32 %[vx], 0x800, %[vx]\n") DP_RS_OPERANDS);
#endif
    } else if constexpr (MODE == 1) {
        asm volatile(DP_RS_HALF1("", "", "", "", "", "s_waitcnt vmcnt(4) lgkmcnt(0)\n", "s_barrier\n") DP_RS_HALF2("", "", "", "", "") DP_RS_OPERANDS);
    } else if constexpr (MODE == 2) {
        asm volatile(DP_RS_HALF1("", "", "", "", "", "s_waitcnt vmcnt(0) lgkmcnt(0)\n", "s_barrier\n") DP_RS_HALF2("", "", "", "", "") DP_RS_OPERANDS);
    } else {
        asm volatile(DP_RS_HALF1("", "", "", "", "", "s_waitcnt lgkmcnt(0)\n", "") DP_RS_HALF2_LAST DP_RS_OPERANDS);
    }
}

// ---- the 128x128 / 4-wave tiling (TC = 2, TS = 2, KB = 2, NB = 4; two workgroups per CU): same stage contract, 8 MFMAs, 8 fragment reads and
// 4 DMA pieces per stage; slots of 16 KiB, so the whole 64-KiB ring is within the 16-bit ds_read offset of ONE base per operand.
#define DP_RM_HALF1(W0S, W0L, W1S, W1L, WADD, WAIT, BARRIER)                                                      \
    DP_RS_MFMA(c00, a00, b00) DP_RS_READ(a10, vA, r1a0) DP_RS_READ(b10, vB, r1b0) W0S                              \
    DP_RS_MFMA(c01, a00, b01) DP_RS_READ(b11, vB, r1b1) DP_RS_READ(a11, vA, r1a1) W0L W1S                          \
    DP_RS_MFMA(c10, a01, b00) W1L WADD WAIT                                                                        \
    DP_RS_MFMA(c11, a01, b01) BARRIER
#define DP_RM_HALF2(X0S, X0L, X1S, X1L, XADD)                                                                      \
    DP_RS_MFMA(c00, a10, b10) DP_RS_READ(a00, vA, r0a0) DP_RS_READ(b00, vB, r0b0) X0S                              \
    DP_RS_MFMA(c01, a10, b11) DP_RS_READ(b01, vB, r0b1) DP_RS_READ(a01, vA, r0a1) X0L X1S                          \
    DP_RS_MFMA(c10, a11, b10) X1L XADD                                                                             \
    DP_RS_MFMA(c11, a11, b11) "s_waitcnt lgkmcnt(0)\n"
#define DP_RM_HALF2_LAST DP_RS_MFMA(c00, a10, b10) DP_RS_MFMA(c01, a10, b11) DP_RS_MFMA(c10, a11, b10) DP_RS_MFMA(c11, a11, b11) DP_RS_XDL_DRAIN
#define DP_RM_OPERANDS                                                                                                                 \
        : [c00] "+v"(acc[0][0]), [c01] "+v"(acc[0][1]), [c10] "+v"(acc[1][0]), [c11] "+v"(acc[1][1]),                                  \
          [a00] "+v"(f0a[0]), [a01] "+v"(f0a[1]), [b00] "+v"(f0b[0]), [b01] "+v"(f0b[1]),                                               \
          [a10] "=&v"(f1a[0]), [a11] "=&v"(f1a[1]), [b10] "=&v"(f1b[0]), [b11] "=&v"(f1b[1]), [vw] "+v"(v_wofs), [vx] "+v"(v_xofs)     \
        : [vA] "v"(vA), [vB] "v"(vB), [sW0] "s"(sW0), [sW1] "s"(sW1), [sX0] "s"(sX0), [sX1] "s"(sX1), [sm0] "s"(s_m0),                  \
          [r1a0] "n"(R1), [r1a1] "n"(R1 + 2048), [r1b0] "n"(R1), [r1b1] "n"(R1 + 2048),                                                \
          [r0a0] "n"(R0), [r0a1] "n"(R0 + 2048), [r0b0] "n"(R0), [r0b1] "n"(R0 + 2048),                                                \
          [mw0] "n"(M), [mw1] "n"(M + 4096), [mx0] "n"(M + 8192), [mx1] "n"(M + 12288)                                                 \
        : "memory", "scc"
template <int S, int MODE = 0>
__device__ __forceinline__ void ring_stage_asm_mid(f32x16 (&acc)[2][2], bf16x8 (&f0a)[2], bf16x8 (&f0b)[2], bf16x8 (&f1a)[2], bf16x8 (&f1b)[2],
                                                   uint32_t vA, uint32_t vB, uint32_t& v_wofs, uint32_t& v_xofs, uint64_t sW0, uint64_t sW1,
                                                   uint64_t sX0, uint64_t sX1, uint32_t s_m0) {
    constexpr int S1 = (S + 1) & 3, D = (S + 3) & 3;
    constexpr int R1 = S * 16384 + 1024, R0 = S1 * 16384, M = D * 16384;
    if constexpr (MODE == 0) {
        asm volatile(DP_RM_HALF1(DP_RS_M0(mw0), DP_RS_DMA(vw, sW0), DP_RS_M0(mw1), DP_RS_DMA(vw, sW1), "v_add_u32 %[vw], 0x800, %[vw]\n", "s_waitcnt vmcnt(6) lgkmcnt(0)\n", "s_barrier\n")
                     DP_RM_HALF2(DP_RS_M0(mx0), DP_RS_DMA(vx, sX0), DP_RS_M0(mx1), DP_RS_DMA(vx, sX1), "v_add_u32 %[vx], 0x800, %[vx]\n") DP_RM_OPERANDS);
    } else if constexpr (MODE == 1) {
        asm volatile(DP_RM_HALF1("", "", "", "", "", "s_waitcnt vmcnt(4) lgkmcnt(0)\n", "s_barrier\n") DP_RM_HALF2("", "", "", "", "") DP_RM_OPERANDS);
    } else if constexpr (MODE == 2) {
        asm volatile(DP_RM_HALF1("", "", "", "", "", "s_waitcnt vmcnt(0) lgkmcnt(0)\n", "s_barrier\n") DP_RM_HALF2("", "", "", "", "") DP_RM_OPERANDS);
    } else {
        asm volatile(DP_RM_HALF1("", "", "", "", "", "s_waitcnt lgkmcnt(0)\n", "") DP_RM_HALF2_LAST DP_RM_OPERANDS);
    }
}

// One 1-KiB DMA piece (prologue; not hot): M0 = LDS byte address of the piece, source = SGPR base + per-lane VGPR offset.
__device__ __forceinline__ void ring_dma_piece(uint32_t voff, uint64_t sbase, uint32_t m0val) {
    asm volatile("s_mov_b32 m0, %2\n s_nop 0\n global_load_lds_dwordx4 %0, %1\n" ::"v"(voff), "s"(sbase), "s"(__builtin_amdgcn_readfirstlane(m0val)) : "memory");
}

__device__ __forceinline__ uint64_t sgpr_u64(uint64_t x) {       // both halves through v_readfirstlane: an "s" operand must live in SGPRs
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}
