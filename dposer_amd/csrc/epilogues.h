// Fused GEMM epilogues for the score network (forward and backward).
// Register layout on entry (see gemm.h): acc[tc][ts][4q + r] = D[channel = cbase + 32 tc + 8 q + 4 hi + r]
//                                                             [sample  = sbase + 32 ts + (lane & 31)],  hi = lane >> 5.
#pragma once
#include "common.h"
#include "rng.h"
#include "sde_dev.h"

// dropout keep decisions for one lane's 16 channels of GroupNorm group g (contract: oracle/philox.py
// dropout_keep_mask).  Two Philox4x32-7 calls (rng.h); call m covers quads q = 2m, 2m+1; 16-bit lanes; lane_in_call = (q%2)*4 + r.
struct DropoutCfg {
    float p;           // drop probability; 0 => disabled
    float scale;       // 1/(1-p)
    uint32_t thr;      // keep <=> lane16 < thr, thr = floor((1-p) * 65536)
    uint32_t site;     // dropout site id 0..4
    uint32_t offset;   // optimisation step
    uint64_t seed;
    int groups_x4;     // H/8 = number of counters per sample
    long long ext_rows;              // rows (samples) of ext_keep: padded rows beyond them keep everything
    const unsigned char* ext_keep;   // TEST HOOK (dposer_scorefc_debug_set_dropout_masks): keep decisions [samples][H] of this site, one byte
                                     // each, used instead of the Philox draw -- feeds a recorded torch mask through the fused step.  Read
                                     // only by the test-hook build (-DDPOSER_TEST_HOOKS, libdposer_hip_testhooks.so); the fields stay in
                                     // the struct so every object file agrees on its layout
};

// keep[4q + r] = 1/(1-p) or 0 for channel 8q + 4hi + r.  Storing the decisions (one bit each) in the forward pass for the
// backward pass was measured (tools/tune_gemm.hip, TUNE_GNBWD): -3 % on the GN-backward GEMM, +3 % on the forward one -- the
// masks are therefore regenerated from Philox(seed, step, site) on both sides and never stored.
__device__ __forceinline__ void dropout_mask16(const DropoutCfg& d, int64_t s, int g, int hi, float keep[16]) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        Philox4 r = philox_at_dropout((uint64_t)s * d.groups_x4 + g * 4 + hi * 2 + m, STREAM_DROPOUT0 + d.site, d.offset, d.seed);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int q = 2 * m + (w >> 1), r0 = (w & 1) * 2;
            keep[4 * q + r0] = ((r.v[w] & 0xffffu) < d.thr) ? d.scale : 0.f;
            keep[4 * q + r0 + 1] = ((r.v[w] >> 16) < d.thr) ? d.scale : 0.f;
        }
    }
}

// Run Epi::sub<TC, TS, tc, ts> over every sub-tile of a wave tile in (tc, ts) order (compile-time indices).
template <typename Epi, int TC, int TS, int I = 0, typename P, typename C>
__device__ __forceinline__ void epi_for_each_sub(const P& pp, C& carry, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, int wrow,
                                                 int split, const float* lpar, int lstride, unsigned char* scr) {
    if constexpr (I < TC * TS) {
        Epi::template sub<TC, TS, I / TS, I % TS>(pp, carry, acc[I / TS][I % TS], cbase, sbase, lane, wrow, split, lpar, lstride, scr);
        epi_for_each_sub<Epi, TC, TS, I + 1>(pp, carry, acc, cbase, sbase, lane, wrow, split, lpar, lstride, scr);
    }
}

// One Philox call = the 8 decisions of quads 2m and 2m+1 (same numbers as dropout_mask16, for register-lean callers).
__device__ __forceinline__ void dropout_mask8(const DropoutCfg& d, int64_t s, int g, int hi, int m, float keep[8]) {
    Philox4 r = philox_at_dropout((uint64_t)s * d.groups_x4 + g * 4 + hi * 2 + m, STREAM_DROPOUT0 + d.site, d.offset, d.seed);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        keep[2 * w] = ((r.v[w] & 0xffffu) < d.thr) ? d.scale : 0.f;
        keep[2 * w + 1] = ((r.v[w] >> 16) < d.thr) ? d.scale : 0.f;
    }
}

// What the training forward pass leaves for the GroupNorm-backward epilogue, per lane of a 32x32 sub-tile (tile-major:
// record ((sample block * H/32 + group) * 64 + lane), so a wave reads / writes one coalesced 512-byte run per sub-tile):
//   rstd  the sample's 1/sqrt(var + eps) of this group (both lane halves hold it)
//   keep  bit 4q + r = dropout keep decision of channel 8q + 4hi + r (all ones when dropout is off)
// The decisions are the Philox draws of dropout_mask16 -- drawn ONCE, in the forward epilogue; re-drawing them in the backward
// epilogue cost ~10 VALU instructions per element there (20 integer multiplies per 8 decisions).
struct GnAux { float rstd; uint32_t keep; };
__device__ __forceinline__ int64_t gn_aux_index(int64_t s0, int group, int H, int lane) { return (((s0 >> 5) * (int64_t)(H >> 5) + group) << 6) + lane; }

// dropout_mask16 that also returns the 16 decisions as a bit set (bit 4q + r)
__device__ __forceinline__ uint32_t dropout_mask16_bits(const DropoutCfg& d, int64_t s, int g, int hi, float keep[16]) {
    uint32_t bits = 0;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        Philox4 r = philox_at_dropout((uint64_t)s * d.groups_x4 + g * 4 + hi * 2 + m, STREAM_DROPOUT0 + d.site, d.offset, d.seed);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int i0 = 4 * (2 * m + (w >> 1)) + (w & 1) * 2;
            const bool k0 = (r.v[w] & 0xffffu) < d.thr, k1 = (r.v[w] >> 16) < d.thr;
            keep[i0] = k0 ? d.scale : 0.f;
            keep[i0 + 1] = k1 ? d.scale : 0.f;
            bits |= (k0 ? 1u : 0u) << i0;
            bits |= (k1 ? 1u : 0u) << (i0 + 1);
        }
    }
    return bits;
}

// the 16 decisions of dropout_mask16 as a bit set only (bit 4q + r); the training epilogue applies them as AND masks
__device__ __forceinline__ uint32_t dropout_bits16(const DropoutCfg& d, int64_t s, int g, int hi) {
    uint32_t bits = 0;
#ifdef DPOSER_TEST_HOOKS                               // libdposer_hip_testhooks.so only: the shipped epilogue has no such branch
    if (d.ext_keep) {                                  // injected decisions (tests): channel 32 g + 8 q + 4 hi + r -> bit 4 q + r
        if (s >= d.ext_rows) return 0xffffu;
        const unsigned char* row = d.ext_keep + s * (int64_t)(d.groups_x4 * 8) + 32 * g + 4 * hi;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) bits |= (row[8 * q + r] ? 1u : 0u) << (4 * q + r);
        return bits;
    }
#endif
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        Philox4 r = philox_at_dropout((uint64_t)s * d.groups_x4 + g * 4 + hi * 2 + m, STREAM_DROPOUT0 + d.site, d.offset, d.seed);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int i0 = 4 * (2 * m + (w >> 1)) + (w & 1) * 2;
            bits |= ((r.v[w] & 0xffffu) < d.thr ? 1u : 0u) << i0;
            bits |= ((r.v[w] >> 16) < d.thr ? 1u : 0u) << (i0 + 1);
        }
    }
    return bits;
}

// ----------------------------------------------------------------------------------------------
// forward:  out = [resid +] Drop(SiLU(GroupNorm32(acc + bias)))         model.py:166-187
// ----------------------------------------------------------------------------------------------
struct GNParams {
    const float* bias;     // [H] dense bias (+ time bias row in shared-t mode)
    const float* gamma;    // [H]
    const float* beta;     // [H]
    void* out;             // FT [Spad][H]
    const void* resid;     // FT [Spad][H] or null
    void* xhat;            // TRAIN: FT [Spad][H]
    GnAux* aux;            // TRAIN: [Spad/32][H/32][64] records for the backward epilogue (rstd + dropout decisions)
    int H;
    DropoutCfg drop;
    void* outT;            // TRAIN, optional: the output again as FT [H][Spad] (operand of the wgrad GEMMs)
    int64_t Spad;
    int act;               // DP_ACT_* (read by the ACTRT instantiations only; 0 = swish)
    void* out_hi;          // fp32-storage instantiations only (bf16x3 mode), optional: the output again as two bf16 FT planes [Spad][H],
    void* out_lo;          //   hi = bf16(out), lo = bf16(out - hi) -- the operand form of the consuming GEMMs; `out` itself may then be null
};
// bf16 planes of a 32 x 32 fp32 tile (bf16x3 mode: common.h TileIO<__bf16> converts with round-to-nearest-even, as k_split_ft32 does)
__device__ __forceinline__ void store_tile_planes(void* hi_plane, void* lo_plane, int64_t s0, int c0, int K, int lane, const float (&v)[16]) {
    float lo[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) lo[r] = v[r] - (float)(__bf16)v[r];
    const int64_t tb = ft_tile_base<__bf16>(s0, c0, K);
    TileIO<__bf16>::store(reinterpret_cast<__bf16*>(hi_plane) + tb, lane, v);
    TileIO<__bf16>::store(reinterpret_cast<__bf16*>(lo_plane) + tb, lane, lo);
}
// RESID: -1 = decide at run time from Params::resid (and null-check the optional outputs); 0 / 1 = residual input absent /
// present at compile time AND no other branch in the code: dropout is always drawn
// (DropoutCfg must then be valid: thr = 65536, scale = 1 when disabled), outT must be non-null.  The branch-free form is what
// the pipelined kernel (gemm_pipe.h) needs: a branch would split the basic block its MFMA / VALU interleave is scheduled in.
template <typename T, bool TRAIN, int RESID = -1, bool ACTRT = false> struct EpiGN {
    typedef GNParams Params;
    static constexpr bool FLAT = RESID >= 0;
    static constexpr int kScratchPerWave = TRAIN ? TileT<T>::SCRATCH_BYTES : 0;
    static constexpr int kParamArrays = 3;
    // TRAIN: the residual input of sub-tile i + 2 is DMA'd (inline asm, common.h) into a per-wave double buffer in the idle
    // K-loop ring while sub-tile i is processed.  (Round 2 measured a builtin-issued DMA prefetch without gain: hipcc waits
    // vmcnt(0) for it in front of the LDS read, i.e. at once.  A layer with a residual input cost 34 us more than one without.)
    static constexpr int kTileBytes = 1024 * (int)sizeof(T);
    static constexpr int kRingPerWave = TRAIN ? 2 * kTileBytes : 0;
    __device__ static inline const float* param_array(const Params& p, int a) { return a == 0 ? p.bias : (a == 1 ? p.gamma : p.beta); }
    __device__ static inline void resid_dma(const T* resid, int64_t s0, int c0, int H, int it, int lane, unsigned ring_lds) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(resid + ft_tile_base<T>(s0, c0, H)) + lane * 16;
#pragma unroll
        for (int h = 0; h < kTileBytes / 1024; ++h) glds_asm_b128(src + h * 1024, ring_lds + (it & 1) * kTileBytes + h * 1024);
    }
    // One 32x32 sub-tile (tc, ts) of the wave tile; `a` = its accumulator.  Sub-tiles are independent of each other, which is
    // what lets the pipelined kernel (gemm_pipe.h) run them inside the k-loop of the NEXT tile.  The work is split in kPhases = 2
    // halves of similar VALU weight (PH = 0: statistics + dropout draw, PH = 1: normalise / SiLU / stores; PH = -1: both), the
    // state between them lives in Carry.
    static constexpr int kPhases = 2;
    struct Carry {
        float v[16];       // centred values (phase 0) -> x_hat
        uint32_t bits;     // dropout keep decisions (bit 4q + r), all ones when dropout is off
        float rstd;
        const unsigned char* ring = nullptr;   // TRAIN + apply_ring: residual tiles arrive here by DMA (see kRingPerWave)
        unsigned ring_lds = 0;
        // residual input: the tile of THIS sub-tile (rc) and the one prefetched for the NEXT sub-tile (rn).  Loaded where it
        // is added, each sub-tile paid an exposed HBM round trip (a layer with a residual input took 25-35 us longer than
        // one without); issued one sub-tile ahead the latency hides under ~500 VALU instructions.
        typename TileIO<T>::Raw rc, rn;
    };
    template <int TC, int TS, int tc, int ts, int PH = -1>
    __device__ static inline void sub(const Params& pp, Carry& cy, const f32x16& a, int cbase, int64_t sbase, int lane, int, int, const float* lpar, int lstride, unsigned char* scr) {
        struct { const float *bias, *gamma, *beta; T* out; const T* resid; T* xhat; GnAux* aux; int H; DropoutCfg drop; } p =
            {pp.bias, pp.gamma, pp.beta, (T*)pp.out, (const T*)pp.resid, (T*)pp.xhat, pp.aux, pp.H, pp.drop};
        constexpr bool PRECISE = sizeof(T) == 4;
        const int j = lane & 31, hi = lane >> 5;
        const int c0 = cbase + tc * 32;
        const int64_t s = sbase + ts * 32 + j;
        const int64_t tb = ft_tile_base<T>(sbase + ts * 32, c0, p.H);
        const bool drop = TRAIN && (FLAT || p.drop.p > 0.f);
        const bool has_res = RESID == 1 || (RESID < 0 && p.resid);
        // (inference only: 168 -> 152 us per 256x256 layer with a residual input.  In the training epilogue neither this
        //  (216 -> 222 us: 16 more live registers) nor a DMA prefetch through the K-loop ring (-> 224 us) helps: with three
        //  output streams per tile that epilogue is HBM-bound, the residual read costs its bytes)
        constexpr bool PREFETCH = !TRAIN;
        if constexpr (PH != 1 && PREFETCH) {
            if (has_res) {
                constexpr int I = tc * TS + ts;
                if constexpr (I == 0) TileIO<T>::load_raw(p.resid + tb, lane, cy.rn);
                cy.rc = cy.rn;
                if constexpr (I + 1 < TC * TS)
                    TileIO<T>::load_raw(p.resid + ft_tile_base<T>(sbase + ((I + 1) % TS) * 32, cbase + ((I + 1) / TS) * 32, p.H), lane, cy.rn);
            }
        }
        if constexpr (PH != 1) {
            // two independent partial sums each (even / odd registers): half the dependent-add chain, and the pairs pack
            float sum0 = 0.f, sum1 = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(lpar + tc * 32 + 8 * q + 4 * hi);   // channel inside the wave's LDS-staged slice
#pragma unroll
                for (int r = 0; r < 4; ++r) cy.v[4 * q + r] = a[4 * q + r] + b4[r];
                sum0 += cy.v[4 * q] + cy.v[4 * q + 2];
                sum1 += cy.v[4 * q + 1] + cy.v[4 * q + 3];
            }
            const float mean = sum_xor32(sum0 + sum1) * (1.0f / 32.0f);
            float ss0 = 0.f, ss1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                cy.v[r] -= mean; cy.v[r + 1] -= mean;
                ss0 += cy.v[r] * cy.v[r]; ss1 += cy.v[r + 1] * cy.v[r + 1];
            }
            const float var = sum_xor32(ss0 + ss1) * (1.0f / 32.0f);
            // (var + eps >= 1e-5: the bare v_rsq_f32 needs none of rsqrtf()'s denormal rescaling)
            cy.rstd = PRECISE ? 1.0f / sqrtf(var + 1e-5f) : __builtin_amdgcn_rsqf(var + 1e-5f);
            cy.bits = 0xffffu;
            if (drop) cy.bits = dropout_bits16(p.drop, s, c0 >> 5, hi);
            if (TRAIN) {
                GnAux rec = {cy.rstd, cy.bits};
                p.aux[gn_aux_index(sbase + ts * 32, c0 >> 5, p.H, lane)] = rec;
            }
        }
        if constexpr (PH != 0) {
            float o[16];
            // dropout: a decision is an AND mask (v_bfe_i32 + v_and); 1/(1-p) rides inside the sigmoid's reciprocal in bf16 mode:
            // a * rcp((1 + e)(1 - p)) -- the add becomes an fma, no extra multiply
            const float dscale = drop ? p.drop.scale : 1.0f, dinv = drop ? 1.0f - p.drop.p : 1.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int cl = tc * 32 + 8 * q + 4 * hi;
                const f32x4 g4 = *reinterpret_cast<const f32x4*>(lpar + lstride + cl);
                const f32x4 e4 = *reinterpret_cast<const f32x4*>(lpar + 2 * lstride + cl);
                if constexpr (!ACTRT && !PRECISE) {
                    // bf16-mode SiLU on register PAIRS: everything around the two transcendentals as packed fp32 math (as scalar code hipcc
                    // left the -log2(e) multiply and the 1 + e add unpacked: 2 of the ~12 VALU operations per value of the inference epilogue)
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        f32x2 v2 = {cy.v[4 * q + r], cy.v[4 * q + r + 1]};
                        v2 *= cy.rstd;                                            // x_hat
                        cy.v[4 * q + r] = v2[0]; cy.v[4 * q + r + 1] = v2[1];
                        const f32x2 g2 = {g4[r], g4[r + 1]}, e2 = {e4[r], e4[r + 1]};
                        const f32x2 a2 = g2 * v2 + e2;
                        const f32x2 t2 = a2 * -1.4426950408889634f;
                        const f32x2 ex = {__builtin_amdgcn_exp2f(t2[0]), __builtin_amdgcn_exp2f(t2[1])};
                        const f32x2 den = TRAIN ? ex * dinv + dinv : ex + 1.0f;   // training: 1 / (1 - p) rides inside the reciprocal
                        const f32x2 rc = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
                        const f32x2 y2 = a2 * rc;
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            float y = y2[u];
                            if constexpr (TRAIN) y = __uint_as_float(__float_as_uint(y) & bit_mask_rt(cy.bits, 4 * q + r + u));
                            o[4 * q + r + u] = y;
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        cy.v[4 * q + r] *= cy.rstd;                                   // x_hat
                        const float a_ = g4[r] * cy.v[4 * q + r] + e4[r];
                        float y;
                        if constexpr (ACTRT) y = TRAIN ? act_rt<PRECISE>(a_, pp.act) * dscale : act_rt<PRECISE>(a_, pp.act);
                        else y = TRAIN ? silu_f<true>(a_) * dscale : silu_f<true>(a_);
                        if constexpr (TRAIN) y = __uint_as_float(__float_as_uint(y) & bit_mask_rt(cy.bits, 4 * q + r));
                        o[4 * q + r] = y;
                    }
                }
            }
            if (TRAIN) TileIO<T>::store(p.xhat + tb, lane, cy.v);
            if (has_res) {
                float rr[16];
                if constexpr (PREFETCH) TileIO<T>::unpack(cy.rc, rr);
                else if (TRAIN && cy.ring) {
                    // VMEM operations younger than the DMA of sub-tile I when this point is reached (see apply_ring): the DMA of
                    // I + 1, and per sub-tile 1 record store + 2 x_hat stores ahead of this point, 2 output stores behind it
                    constexpr int I = tc * TS + ts, NSUB = TC * TS, ND = kTileBytes / 1024;
                    constexpr int younger = (I + 1 < NSUB ? ND : 0) + (I == 0 ? 3 : (I == 1 ? 8 : 10));
                    wait_vmcnt_n(younger);
                    asm volatile("" ::: "memory");
                    TileIO<T>::load(reinterpret_cast<const T*>(cy.ring + (I & 1) * kTileBytes), lane, rr);
                    if constexpr (I + 2 < NSUB) {
                        wait_lgkmcnt0();
                        asm volatile("" ::: "memory");
                        resid_dma(p.resid, sbase + ((I + 2) % TS) * 32, cbase + ((I + 2) / TS) * 32, p.H, I + 2, lane, cy.ring_lds);
                    }
                } else TileIO<T>::load(p.resid + tb, lane, rr);
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] += rr[r];
            }
            if constexpr (sizeof(T) == 4 && !FLAT) {
                if (p.out) TileIO<T>::store(p.out + tb, lane, o);
                if (pp.out_hi) store_tile_planes(pp.out_hi, pp.out_lo, sbase + ts * 32, c0, p.H, lane, o);
            } else {
                TileIO<T>::store(p.out + tb, lane, o);
            }
            if (TRAIN && (FLAT || pp.outT)) TileT<T>::store((T*)pp.outT + ft_tileT_base<T>(sbase + ts * 32, c0, pp.Spad), scr, lane, o);
        }
    }
    template <int TC, int TS>
    __device__ static inline void apply(const Params& pp, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, int wrow, int split, const float* lpar, int lstride, unsigned char* scr) {
        Carry c;
        epi_for_each_sub<EpiGN, TC, TS>(pp, c, acc, cbase, sbase, lane, wrow, split, lpar, lstride, scr);
    }
    template <int TC, int TS>
    __device__ static inline void apply_ring(const Params& pp, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, int wrow, int split, const float* lpar, int lstride, unsigned char* scr, unsigned char* ring) {
        typedef __attribute__((address_space(3))) void* lptr_t;
        Carry c;
        if (pp.resid) {
            c.ring = ring;
            c.ring_lds = (unsigned)(size_t)(lptr_t)ring;
            resid_dma((const T*)pp.resid, sbase, cbase, pp.H, 0, lane, c.ring_lds);
            if constexpr (TC * TS > 1) resid_dma((const T*)pp.resid, sbase + (1 % TS) * 32, cbase + (1 / TS) * 32, pp.H, 1, lane, c.ring_lds);
        }
        epi_for_each_sub<EpiGN, TC, TS>(pp, c, acc, cbase, sbase, lane, wrow, split, lpar, lstride, scr);
    }
};

// forward: temb = SiLU(acc + bias)  (shared_time_embed, model.py:124-127,164); TRAIN keeps u.
struct BiasSiLUParams {
    const float* bias;
    void* out;     // FT [Spad][N]
    void* pre;     // TRAIN: u = acc + bias, FT [Spad][N]
    int N;
    void* outT;    // TRAIN, optional: FT [N][Spad]
    int64_t Spad;
    int act;       // DP_ACT_* (ACTRT instantiations)
    void* out_hi;  // fp32-storage instantiations only (bf16x3 mode), optional: `out` again as two bf16 FT planes
    void* out_lo;
    DropoutCfg drop;   // TRAIN only: nn.Dropout BEHIND the activation (TimeMLPs blocks, model.py:78-82); p = 0: none (shared_time_embed has none)
};
template <typename T, bool TRAIN, bool ACTRT = false> struct EpiBiasSiLU {
    typedef BiasSiLUParams Params;
    static constexpr int kScratchPerWave = TRAIN ? TileT<T>::SCRATCH_BYTES : 0;
    static constexpr int kParamArrays = 1;
    __device__ static inline const float* param_array(const Params& p, int) { return p.bias; }
    template <int TC, int TS>
    __device__ static inline void apply(const Params& pp, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, int, int, const float* lpar, int lstride, unsigned char* scr) {
        struct { const float* bias; T* out; T* pre; int N; } p = {pp.bias, (T*)pp.out, (T*)pp.pre, pp.N};
        constexpr bool PRECISE = sizeof(T) == 4;
        const int hi = lane >> 5;
#pragma unroll
        for (int tc = 0; tc < TC; ++tc) {
            const int c0 = cbase + tc * 32;
            float bia[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 b4 = *reinterpret_cast<const f32x4*>(lpar + tc * 32 + 8 * q + 4 * hi);
#pragma unroll
                for (int r = 0; r < 4; ++r) bia[4 * q + r] = b4[r];
            }
#pragma unroll
            for (int ts = 0; ts < TS; ++ts) {
                const int64_t tb = ft_tile_base<T>(sbase + ts * 32, c0, p.N);
                float u[16], o[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) { u[r] = acc[tc][ts][r] + bia[r]; o[r] = ACTRT ? act_rt<PRECISE>(u[r], pp.act) : silu_f<PRECISE>(u[r]); }
                if (TRAIN && pp.drop.p > 0.f) {      // the decisions are re-drawn from the same counters by the backward epilogue (EpiSiLUBwd)
                    const uint32_t bits = dropout_bits16(pp.drop, sbase + ts * 32 + (lane & 31), c0 >> 5, hi);
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] = ((bits >> r) & 1u) ? o[r] * pp.drop.scale : 0.f;
                }
                if (TRAIN) TileIO<T>::store(p.pre + tb, lane, u);
                TileIO<T>::store(p.out + tb, lane, o);
                if constexpr (sizeof(T) == 4) {
                    if (pp.out_hi) store_tile_planes(pp.out_hi, pp.out_lo, sbase + ts * 32, c0, p.N, lane, o);
                }
                if (TRAIN && pp.outT) TileT<T>::store((T*)pp.outT + ft_tileT_base<T>(sbase + ts * 32, c0, pp.Spad), scr, lane, o);
            }
        }
    }
};

// out[s][c] = acc + bias[c] (fp32, row-major, leading dimension ldc) for c < C_valid, s < S_valid.
// Used by post_dense (model.py:189), the shared-t time-bias table and plain input-gradients.
struct RowMajorParams {
    const float* bias;   // may be null
    float* out;
    int64_t ldc;
    int C_valid;
    int64_t S_valid;
};
template <typename T> struct EpiRowMajor {
    typedef RowMajorParams Params;
    // Row-major fp32 output from an accumulator whose LANES are samples: stored directly, every instruction writes 64 dwords
    // that lie a whole output row apart.  apply_ring (used whenever the idle K-loop ring has 4.1 KiB per wave to spare, i.e.
    // always) transposes each 32x32 sub-tile through LDS and writes two 128-byte row segments per instruction instead
    // (post_dense at 65536 samples: 45 -> 41 us; same values, same order of operations).
    static constexpr int kRingPerWave = 32 * 33 * 4;
    template <int TC, int TS>
    __device__ static inline void apply(const Params& p, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, int, int, const float* lpar, int lstride, unsigned char* scr) {
        const int j = lane & 31, hi = lane >> 5;
#pragma unroll
        for (int tc = 0; tc < TC; ++tc)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = cbase + tc * 32 + 8 * q + 4 * hi;
                float b[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) b[r] = (p.bias && c + r < p.C_valid) ? p.bias[c + r] : 0.f;
#pragma unroll
                for (int ts = 0; ts < TS; ++ts) {
                    const int64_t s = sbase + ts * 32 + j;
                    if (s < p.S_valid) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (c + r < p.C_valid) p.out[s * p.ldc + c + r] = acc[tc][ts][4 * q + r] + b[r];
                    }
                }
            }
    }
    template <int TC, int TS>
    __device__ static inline void apply_ring(const Params& p, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, int, int, const float* lpar, int lstride, unsigned char* scr, unsigned char* ring) {
        const int j = lane & 31, hi = lane >> 5;
        float* t = reinterpret_cast<float*>(ring);          // [32 samples][33]
#pragma unroll
        for (int tc = 0; tc < TC; ++tc) {
            const int c0 = cbase + tc * 32;
            float b[16];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = c0 + 8 * q + 4 * hi + r;
                    b[4 * q + r] = (p.bias && c < p.C_valid) ? p.bias[c] : 0.f;
                }
            const int cc = c0 + j;                              // this lane's output column in the row-wise pass
#pragma unroll
            for (int ts = 0; ts < TS; ++ts) {
                const int64_t s0 = sbase + ts * 32;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) t[j * 33 + 8 * q + 4 * hi + r] = acc[tc][ts][4 * q + r] + b[4 * q + r];
                asm volatile("" ::: "memory");                  // wave-private scratch, LDS executes a wave's accesses in order
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int row = 2 * k + hi;
                    const float v = t[row * 33 + j];
                    const int64_t s = s0 + row;
                    if (s < p.S_valid && cc < p.C_valid) p.out[s * p.ldc + cc] = v;
                }
                asm volatile("" ::: "memory");
            }
        }
    }
};

// post_dense fused with one Euler-Maruyama predictor step (sampler fast path: no observation, in-kernel noise, no trajectory).
//   res    = acc + bias                                   model.py:189
//   score  = -(res / used_sigma) / std(t)                 model.py:194, utils.py:162
//   x_mean = x + (-1/2 beta x - g^2 score) * dt           sde_lib.py:98-104, sampling.py:186
//   x      = x_mean + g sqrt(-dt) z,  z ~ Philox          sampling.py:187
// The state x lives in HBM as fp32 FT [Spad][64] between steps (coalesced tile I/O; row-major only at the two ends of the
// sampler), and the next step's network input (FT of T) is written from the same registers.  Same fp32 operation order as
// k_em_update; the Philox counter (sample * QD + channel / 4, STREAM_EM_NOISE, step) is the one oracle/philox.py restates.
struct EmStepParams {
    const float* bias;     // [Cp] post_dense bias
    float* x_ft;           // FT fp32 [Spad][Cp], in/out
    float* x_mean_ft;      // FT fp32 [Spad][Cp] or null (only the last step's is needed)
    void* xin;             // FT [Spad][Dpad] next network input
    const float* sigmas;
    SdeDev sde;
    float t;
    int num_scales, scale_by_sigma;
    int D, Cp, QD;
    int64_t S_valid;
    uint64_t seed;
    uint32_t step;
};
template <typename T> struct EpiEmStep {
    typedef EmStepParams Params;
    // per-step scalars (identical for every sample: vec_t = ones(B) * t, sampling.py:458)
    struct Scal { float sd, beta, g, usig; };
    __device__ static inline Scal scalars(const Params& p) {
#pragma clang fp contract(off)
        Scal sc;
        const SdeAt at = sde_at(p.sde, p.t);
        sc.sd = at.sd_score;      // (only the score is formed with it: utils.py:155 / :160)
        sc.beta = at.beta;
        sc.g = at.g;
        sc.usig = p.scale_by_sigma ? used_sigma(p.sigmas, p.num_scales, at.label, p.scale_by_sigma == 2) : 1.0f;
        return sc;
    }
    // one 32x32 tile: x (in: the state, out: the next state) and x_mean from the post_dense accumulator `a`
    // the tile's 16 standard normals per lane (they do not depend on the accumulator: drawn in front of the K loop, see Pre)
    __device__ static inline void draw(const Params& p, int c0, int64_t s, int hi, float (&z)[16]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float z4[4];
            normals4((uint64_t)s * p.QD + ((c0 + 8 * q + 4 * hi) >> 2), STREAM_EM_NOISE, p.step, p.seed, z4);
#pragma unroll
            for (int r = 0; r < 4; ++r) z[4 * q + r] = z4[r];
        }
    }
    __device__ static inline void tile(const Params& p, const Scal& sc, const f32x16& a, int c0, int64_t s, int hi, float (&x)[16], float (&xm)[16], const float (&z)[16]) {
#pragma clang fp contract(off)
        const float sd = sc.sd, beta = sc.beta, g = sc.g, usig = sc.usig;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = c0 + 8 * q + 4 * hi;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * q + r;
                const bool valid = s < p.S_valid && c + r < p.D;
                const float model = (a[i] + (c + r < p.D ? p.bias[c + r] : 0.f)) / usig;
                const float score = sde_score(p.sde, model, sd);
                float drift = (-0.5f * beta) * x[i];
                drift = drift - ((g * g) * score) * 1.0f;
                const float mean = x[i] + drift * p.sde.dt;
                xm[i] = valid ? mean : 0.f;
                x[i] = valid ? mean + (g * p.sde.sqrt_mdt) * z[i] : 0.f;
            }
        }
    }
    // In front of the K loop (gemm.h EpiPre): the state tiles are requested before the first DMA and the normals are drawn behind the prologue's
    // DMA issue -- at 500 samples (16 workgroups, a latency-bound launch) both used to sit behind the last MFMA: 11.9 us per step, 6.3 of them K loop.
    template <int TC, int TS> struct Pre { float x[TC * TS][16]; float z[TC * TS][16]; };
    template <int TC, int TS>
    __device__ static inline void pre_issue(const Params& p, Pre<TC, TS>& pre, int cbase, int64_t sbase, int lane) {
#pragma unroll
        for (int tc = 0; tc < TC; ++tc)
#pragma unroll
            for (int ts = 0; ts < TS; ++ts) TileIO<float>::load(p.x_ft + ft_tile_base<float>(sbase + ts * 32, cbase + tc * 32, p.Cp), lane, pre.x[tc * TS + ts]);
    }
    template <int TC, int TS>
    __device__ static inline void pre_compute(const Params& p, Pre<TC, TS>& pre, int cbase, int64_t sbase, int lane) {
#pragma unroll
        for (int tc = 0; tc < TC; ++tc)
#pragma unroll
            for (int ts = 0; ts < TS; ++ts) {
                draw(p, cbase + tc * 32, sbase + ts * 32 + (lane & 31), lane >> 5, pre.z[tc * TS + ts]);
                // (pin: without it hipcc sinks the whole draw to its first use, behind the K loop)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(pre.z[tc * TS + ts][i]));
            }
    }
    template <int TC, int TS>
    __device__ static inline void apply_pre(const Params& p, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, Pre<TC, TS>& pre) {
        const int j = lane & 31, hi = lane >> 5;
        const Scal sc = scalars(p);
#pragma unroll
        for (int tc = 0; tc < TC; ++tc)
#pragma unroll
            for (int ts = 0; ts < TS; ++ts) {
                const int c0 = cbase + tc * 32;
                const int64_t s = sbase + ts * 32 + j;
                const int64_t tb = ft_tile_base<float>(sbase + ts * 32, c0, p.Cp);
                float xm[16];
                tile(p, sc, acc[tc][ts], c0, s, hi, pre.x[tc * TS + ts], xm, pre.z[tc * TS + ts]);
                TileIO<float>::store(p.x_ft + tb, lane, pre.x[tc * TS + ts]);
                if (p.x_mean_ft) TileIO<float>::store(p.x_mean_ft + tb, lane, xm);
                TileIO<T>::store((T*)p.xin + ft_tile_base<T>(sbase + ts * 32, c0, p.Cp), lane, pre.x[tc * TS + ts]);
            }
    }
    template <int TC, int TS>
    __device__ static inline void apply(const Params& p, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, int, int, const float*, int, unsigned char*) {
        const int j = lane & 31, hi = lane >> 5;
        const Scal sc = scalars(p);
#pragma unroll
        for (int tc = 0; tc < TC; ++tc)
#pragma unroll
            for (int ts = 0; ts < TS; ++ts) {
                const int c0 = cbase + tc * 32;
                const int64_t s = sbase + ts * 32 + j;
                const int64_t tb = ft_tile_base<float>(sbase + ts * 32, c0, p.Cp);
                float x[16], xm[16], z[16];
                TileIO<float>::load(p.x_ft + tb, lane, x);
                draw(p, c0, s, hi, z);
                tile(p, sc, acc[tc][ts], c0, s, hi, x, xm, z);
                TileIO<float>::store(p.x_ft + tb, lane, x);
                if (p.x_mean_ft) TileIO<float>::store(p.x_mean_ft + tb, lane, xm);
                TileIO<T>::store((T*)p.xin + ft_tile_base<T>(sbase + ts * 32, c0, p.Cp), lane, x);
            }
    }
};

// out FT = acc  (+ optional FT addend): plain fragment-tiled store (dgrad into the time branch).
struct PlainFTParams {
    void* out;
    int N;
};
template <typename T> struct EpiPlainFT {
    typedef PlainFTParams Params;
    static constexpr int kPhases = 1;
    struct Carry {};
    template <int TC, int TS, int tc, int ts, int PH = -1>
    __device__ static inline void sub(const Params& pp, Carry&, const f32x16& a, int cbase, int64_t sbase, int lane, int, int, const float*, int, unsigned char*) {
        float o[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = a[r];
        TileIO<T>::store((T*)pp.out + ft_tile_base<T>(sbase + ts * 32, cbase + tc * 32, pp.N), lane, o);
    }
    template <int TC, int TS>
    __device__ static inline void apply(const Params& pp, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, int wrow, int split, const float* lpar, int lstride, unsigned char* scr) {
        Carry c;
        epi_for_each_sub<EpiPlainFT, TC, TS>(pp, c, acc, cbase, sbase, lane, wrow, split, lpar, lstride, scr);
    }
};

// out[split] FT fp32 = acc: the partial sums of one k-split, whatever the operand type (the time-branch dgrad at small batches: one split
// per layer segment, summed and finished by k_silu_bwd_reduce).
struct PartialFTParams {
    float* out;            // [ksplit][Spad][N] fp32 FT
    int N;
    int64_t split_stride;  // elements between the splits' outputs
};
template <typename T> struct EpiPartialFT {
    typedef PartialFTParams Params;
    template <int TC, int TS>
    __device__ static inline void apply(const Params& pp, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, int, int split, const float*, int, unsigned char*) {
        float* base = pp.out + (int64_t)split * pp.split_stride;
#pragma unroll
        for (int tc = 0; tc < TC; ++tc)
#pragma unroll
            for (int ts = 0; ts < TS; ++ts) {
                float o[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = acc[tc][ts][r];
                TileIO<float>::store(base + ft_tile_base<float>(sbase + ts * 32, cbase + tc * 32, pp.N), lane, o);
            }
    }
};

// ----------------------------------------------------------------------------------------------
// backward helpers
// ----------------------------------------------------------------------------------------------
// All-to-all butterfly over the 32 lanes of a half-wave: on entry every lane holds N partial values; on exit lane
// (l & 31) == i holds the sum over the 32 lanes (same l >> 5) of value i.  A step with lane bit b pairs lane l with a partner
// that differs in bit b; lanes with the bit set keep the upper half of the live values, the others the lower half.
// The partner exchange never touches LDS (the first version used ds_bpermute: an LDS round trip and an s_waitcnt per
// shuffle, 47 per channel tile):
//   bit 4      v_permlane16_swap: one swap + one add per value pair, no select
//   bits 3, 2  DPP row_mirror (l ^ 15) / row_half_mirror (l ^ 7) folded into the adds
//   bits 1, 0  DPP quad_perm (l ^ 2, l ^ 1)
// (the mirror partners flip lower bits too; the lower bits are still "unreduced" at that point, so the orbit of a lane
//  under {^16, ^15, ^7, ^2, ^1} is still the whole half-wave and the result lands where the xor butterfly put it.)
template <int CTRL> __device__ __forceinline__ float dpp_f32(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
template <int D> struct DppCtrl;
template <> struct DppCtrl<8> { static constexpr int value = 0x140; };   // row_mirror
template <> struct DppCtrl<4> { static constexpr int value = 0x141; };   // row_half_mirror
template <> struct DppCtrl<2> { static constexpr int value = 0x4E; };    // quad_perm [2,3,0,1]
template <> struct DppCtrl<1> { static constexpr int value = 0xB1; };    // quad_perm [1,0,3,2]
template <int D> __device__ __forceinline__ void butterfly_step(float (&v)[32], int lane) {
    if constexpr (D == 16) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 16]), false, false);
            v[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
        }
    } else {
        const bool up = (lane & D) != 0;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const float lo = v[i] + dpp_f32<DppCtrl<D>::value>(v[i]);
            const float hi = v[i + D] + dpp_f32<DppCtrl<D>::value>(v[i + D]);
            v[i] = up ? hi : lo;
        }
    }
}
// 16 values per lane: lane (l & 15) == i ends with the sum over the 32 lanes (same l >> 5) of value i.
__device__ __forceinline__ float butterfly_reduce16(const float (&in)[16], int lane) {
    float v[32];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = in[r];
    butterfly_step<8>(v, lane);
    butterfly_step<4>(v, lane);
    butterfly_step<2>(v, lane);
    butterfly_step<1>(v, lane);
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[0]), __float_as_uint(v[0]), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int N> __device__ __forceinline__ void butterfly_reduce32(float (&v)[N], int lane) {
    static_assert(N == 32, "N must be 32");
    butterfly_step<16>(v, lane);
    butterfly_step<8>(v, lane);
    butterfly_step<4>(v, lane);
    butterfly_step<2>(v, lane);
    butterfly_step<1>(v, lane);
}
// the first version (ds_bpermute shuffles), kept for the tuner's A/B
template <int D> __device__ __forceinline__ void butterfly_step_bperm(float (&v)[32], int lane) {
    const bool up = (lane & D) != 0;
#pragma unroll
    for (int i = 0; i < D; ++i) {
        const float send = up ? v[i] : v[i + D];
        const float keep = up ? v[i + D] : v[i];
        v[i] = keep + __shfl_xor(send, D);
    }
}
__device__ __forceinline__ float butterfly_reduce16_bperm(const float (&in)[16], int lane) {
    float v[32];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = in[r];
    butterfly_step_bperm<8>(v, lane); butterfly_step_bperm<4>(v, lane); butterfly_step_bperm<2>(v, lane); butterfly_step_bperm<1>(v, lane);
    return v[0] + __shfl_xor(v[0], 16);
}
__device__ __forceinline__ void butterfly_reduce32_bperm(float (&v)[32], int lane) {
    butterfly_step_bperm<16>(v, lane); butterfly_step_bperm<8>(v, lane); butterfly_step_bperm<4>(v, lane); butterfly_step_bperm<2>(v, lane);
    butterfly_step_bperm<1>(v, lane);
}

// dgrad epilogue: acc = dL/d(out_l) contribution through the NEXT layer's weights.
//   g   = acc [+ carry_in]          (carry = gradient arriving over the residual connection)
//   da  = g * keep/(1-p) * silu'(a),   a = gamma*xhat + beta
//   dy  = rstd * (dx - mean_g(dx) - xhat * mean_g(dx * xhat)),   dx = da * gamma
// writes dy (FT), optional carry_out = g, and per-wave partial sums of dgamma, dbeta, dbias.
// Padded rows: the incoming gradient is zero there (dres is written as zeros on padded rows and every dgrad keeps that), so
// nothing has to be masked per element; the row validity is folded into the dropout decision bits once per sub-tile.
struct GNBwdParams {
    const void* carry_in;  // FT [Spad][H] or null
    void* carry_out;       // FT [Spad][H] or null
    const void* xhat;      // FT [Spad][H]
    const GnAux* aux;      // records of the forward epilogue (rstd + dropout decisions), tile-major
    const float* gamma;
    const float* beta;
    void* dy;              // FT [Spad][H]
    float* part;           // [n_rows][3][H] partial sums (row = wave row id), deterministic
    int H;
    int64_t S_valid;
    float drop_scale;      // 1/(1-p) of the forward pass (1 when dropout was off)
    void* dyT;             // optional: dy again as FT [H][Spad]
    int64_t Spad;
    int act;               // DP_ACT_* (ACTRT instantiations)
    void* dy_hi;           // fp32-storage instantiations only (bf16x3 mode), optional: dy as two bf16 FT planes [Spad][H]; `dy` may then be null
    void* dy_lo;
};
// ABL (tuner only): 1 = no parameter-gradient sums, 2 = no SiLU', 4 = ds_bpermute butterflies, 8 = loads / stores only
template <typename T, int ABL = 0, bool ACTRT = false> struct EpiGNBwd {
    typedef GNBwdParams Params;
    static constexpr int kScratchPerWave = TileT<T>::SCRATCH_BYTES;
    static constexpr int kMinWaves = 2;   // keep two 256-thread workgroups per CU (register-heavy epilogue)
    static constexpr int kParamArrays = 2;
    // Operand prefetch through the (then idle) K-loop ring: x_hat and the residual carry of sub-tile i+1 are DMA'd
    // (global_load_lds, 1 KiB pieces) into a per-wave double buffer while sub-tile i is being processed.  Loaded straight
    // into registers, each sub-tile paid a full HBM round trip behind an s_waitcnt (8 per wave tile, and both waves of a
    // SIMD are in the same place at the same time); the register budget leaves no room to prefetch there.
    static constexpr int kTileBytes = 1024 * (int)sizeof(T);
    static constexpr bool kDmaAll = sizeof(T) == 2;           // bf16: the GnAux records travel through the ring too (run_dma)
    static constexpr int kBufBytes = 2 * kTileBytes + (kDmaAll ? 512 : 0);   // x_hat | carry | rstd words, keep words
    static constexpr int kRingPerWave = 2 * kBufBytes;        // two buffers
    __device__ static inline const float* param_array(const Params& p, int a) { return a == 0 ? p.gamma : p.beta; }
    template <int TC, int TS>
    __device__ static inline void apply(const Params& pp, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, int wrow, int split, const float* lpar, int lstride, unsigned char* scr) {
        run<TC, TS, false>(pp, acc, cbase, sbase, lane, wrow, lpar, lstride, scr, nullptr);
    }
    template <int TC, int TS>
    __device__ static inline void apply_ring(const Params& pp, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, int wrow, int split, const float* lpar, int lstride, unsigned char* scr, unsigned char* ring) {
        if constexpr (kDmaAll && (ABL & 16) == 0) run_dma<TC, TS>(pp, acc, cbase, sbase, lane, wrow, lpar, lstride, scr, ring);
        else run<TC, TS, true>(pp, acc, cbase, sbase, lane, wrow, lpar, lstride, scr, ring);
    }
    // bf16 pipeline, round 3.  What the epilogue was waiting for (round-2 ablation: loads + stores alone cost 31 us per launch,
    // i.e. ~1 us per 32x32 sub-tile and wave with NO arithmetic): every sub-tile (1) loaded its GnAux record with an ordinary
    // load right where it is needed, (2) waited with a counted vmcnt that also covers the stores of the sub-tile before, and
    // (3) hipcc, seeing an ordinary load beside global_load_lds, drained the queue with vmcnt(0) -- the "prefetch" of the next
    // sub-tile was waited for at once.  Here EVERYTHING the sub-tile reads (x_hat, residual carry, the record as two dword
    // pieces) arrives by DMA issued from inline asm (common.h), two sub-tiles ahead (the buffer of sub-tile k is refilled
    // for k + 2 as soon as its LDS reads have returned), and the wait in front of sub-tile k allows every younger operation
    // to stay in flight: the DMA of k + 1 and the stores of k - 1 and k - 2.
    template <int TC, int TS>
    __device__ static inline void run_dma(const Params& pp, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, int wrow, const float* lpar, int lstride, unsigned char* scr, unsigned char* ring) {
        static_assert(sizeof(T) == 2, "bf16 only");
        struct { const T* carry_in; T* carry_out; const T* xhat; const GnAux* aux; T* dy; float* part; int H; int64_t S_valid; float scale; } p =
            {(const T*)pp.carry_in, (T*)pp.carry_out, (const T*)pp.xhat, pp.aux, (T*)pp.dy, pp.part, pp.H, pp.S_valid, pp.drop_scale};
        constexpr bool PRECISE = false;
        constexpr int NSUB = TC * TS;
        typedef __attribute__((address_space(3))) void* lptr_t;
        const int j = lane & 31, hi = lane >> 5;
        const bool has_carry = p.carry_in != nullptr;
        const int nd = has_carry ? 6 : 4;                                 // DMA pieces per sub-tile
        const unsigned ring_lds = (unsigned)(size_t)(lptr_t)ring;
        auto dma = [&](int it) __attribute__((always_inline)) {
            const int tc = it / TS, ts = it % TS;
            const int64_t tb = ft_tile_base<T>(sbase + ts * 32, cbase + tc * 32, p.H);
            const unsigned dst = ring_lds + (it & 1) * kBufBytes;
            const unsigned char* xs = reinterpret_cast<const unsigned char*>(p.xhat + tb) + lane * 16;
            glds_asm_b128(xs, dst);
            glds_asm_b128(xs + 1024, dst + 1024);
            if (has_carry) {
                const unsigned char* cs = reinterpret_cast<const unsigned char*>(p.carry_in + tb) + lane * 16;
                glds_asm_b128(cs, dst + 2 * 1024);
                glds_asm_b128(cs + 1024, dst + 3 * 1024);
            }
            const unsigned char* as = reinterpret_cast<const unsigned char*>(p.aux + gn_aux_index(sbase + ts * 32, (cbase + tc * 32) >> 5, p.H, lane));
            glds_asm_b32(as, dst + 4 * 1024);             // rstd words
            glds_asm_b32(as + 4, dst + 4 * 1024 + 256);   // keep words
        };
        dma(0);
        if constexpr (NSUB > 1) dma(1);
#pragma unroll
        for (int tc = 0; tc < TC; ++tc) {
            const int c0 = cbase + tc * 32;
            float stat[32];   // [0..15] dgamma, [16..31] dbeta (this lane's 16 channels), without the 1/(1-p) factor
            float dbias[16];
#pragma unroll
            for (int r = 0; r < 32; ++r) stat[r] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) dbias[r] = 0.f;
#pragma unroll
            for (int ts = 0; ts < TS; ++ts) {
                const int it = tc * TS + ts;                     // (compile-time after unrolling)
                const int64_t s = sbase + ts * 32 + j;
                const int64_t tb = ft_tile_base<T>(sbase + ts * 32, c0, p.H);
                // operations younger than the DMA of sub-tile `it` that may stay in flight (VMEM retires in issue order):
                //   it = 0: the DMA of 1;   it = 1: the DMA of 2 and the two dy stores of 0;
                //   it >= 2: the dy stores of it - 2, the DMA of it + 1, the dy stores of it - 1
                // (carry_out / partial-sum stores are younger operations too: not counting them only waits a little longer)
                const int younger_dma = (it + 1 < NSUB) ? nd : 0;
                wait_vmcnt_n(it == 0 ? younger_dma : (it == 1 ? younger_dma + 2 : younger_dma + 4));
                asm volatile("" ::: "memory");
                const unsigned char* buf = ring + (it & 1) * kBufBytes;
                float xh[16], g[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) g[r] = acc[tc][ts][r];
                TileIO<T>::load(reinterpret_cast<const T*>(buf), lane, xh);
                if (has_carry) {
                    float ci[16];
                    TileIO<T>::load(reinterpret_cast<const T*>(buf + 2 * 1024), lane, ci);
#pragma unroll
                    for (int r = 0; r < 16; ++r) g[r] += ci[r];
                }
                const float rstd = *reinterpret_cast<const float*>(buf + 4 * 1024 + lane * 4);
                const uint32_t keep_bits = *reinterpret_cast<const uint32_t*>(buf + 4 * 1024 + 256 + lane * 4);
                const uint32_t bits = s < p.S_valid ? keep_bits : 0u;
                if (it + 2 < NSUB) {                             // this buffer is free once its reads have returned
                    wait_lgkmcnt0();
                    asm volatile("" ::: "memory");
                    dma(it + 2);
                }
                if (p.carry_out) TileIO<T>::store(p.carry_out + tb, lane, g);
                float s1 = 0.f, s2 = 0.f;
                if constexpr (!ACTRT && (ABL & 32) == 0) {
                    // register PAIRS (packed fp32 math around the two transcendentals; even / odd partial sums): as scalar code hipcc left ~850
                    // of this epilogue's ~3500 VALU instructions per wave tile as unpacked v_add_f32 (the running sums are dependent chains)
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    f32x2 s1p = {0.f, 0.f}, s2p = {0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int cl = tc * 32 + 8 * q + 4 * hi;
                        const f32x4 g4 = *reinterpret_cast<const f32x4*>(lpar + cl);
                        const f32x4 e4 = *reinterpret_cast<const f32x4*>(lpar + lstride + cl);
#pragma unroll
                        for (int r = 0; r < 4; r += 2) {
                            const int i = 4 * q + r;
                            const f32x2 xh2 = {xh[i], xh[i + 1]}, gm2 = {g4[r], g4[r + 1]}, be2 = {e4[r], e4[r + 1]};
                            const f32x2 gg2 = {__uint_as_float(__float_as_uint(g[i]) & bit_mask_rt(bits, i)),
                                               __uint_as_float(__float_as_uint(g[i + 1]) & bit_mask_rt(bits, i + 1))};
                            const f32x2 a2 = gm2 * xh2 + be2;
                            const f32x2 t2 = a2 * -1.4426950408889634f;
                            const f32x2 ex = {__builtin_amdgcn_exp2f(t2[0]), __builtin_amdgcn_exp2f(t2[1])};
                            const f32x2 den = ex + 1.0f;
                            const f32x2 sg = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
                            const f32x2 ds2 = sg * (a2 * (1.0f - sg) + 1.0f);          // silu'(a) = s (1 + a (1 - s))
                            const f32x2 da2 = gg2 * ds2;                               // (the 1/(1-p) factor is applied to gamma and to the sums)
                            f32x2 st0 = {stat[i], stat[i + 1]}, st1 = {stat[16 + i], stat[17 + i]};
                            st0 += da2 * xh2;
                            st1 += da2;
                            stat[i] = st0[0]; stat[i + 1] = st0[1]; stat[16 + i] = st1[0]; stat[17 + i] = st1[1];
                            const f32x2 dx2 = da2 * (gm2 * p.scale);                   // dx
                            g[i] = dx2[0]; g[i + 1] = dx2[1];
                            s1p += dx2;
                            s2p += dx2 * xh2;
                        }
                    }
                    s1 = sum_xor32(s1p[0] + s1p[1]);
                    s2 = sum_xor32(s2p[0] + s2p[1]);
                    const float m1 = s1 * (1.0f / 32.0f), m2 = s2 * (1.0f / 32.0f);
#pragma unroll
                    for (int i = 0; i < 16; i += 2) {
                        const f32x2 xh2 = {xh[i], xh[i + 1]};
                        f32x2 g2 = {g[i], g[i + 1]};
                        g2 = ((g2 - m1) - xh2 * m2) * rstd;                           // dy
                        g[i] = g2[0]; g[i + 1] = g2[1];
                        f32x2 db2 = {dbias[i], dbias[i + 1]};
                        db2 += g2;
                        dbias[i] = db2[0]; dbias[i + 1] = db2[1];
                    }
                } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int cl = tc * 32 + 8 * q + 4 * hi;
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(lpar + cl);
                    const f32x4 e4 = *reinterpret_cast<const f32x4*>(lpar + lstride + cl);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 4 * q + r;
                        const uint32_t m = bit_mask_rt(bits, i);
                        const float gg = __uint_as_float(__float_as_uint(g[i]) & m);
                        const float ds = ACTRT ? dact_rt<PRECISE>(g4[r] * xh[i] + e4[r], pp.act) : dsilu_f<PRECISE>(g4[r] * xh[i] + e4[r]);
                        const float da = gg * ds;                  // (the 1/(1-p) factor is applied to gamma and to the sums)
                        stat[i] += da * xh[i];
                        stat[16 + i] += da;
                        g[i] = da * (g4[r] * p.scale);             // dx
                        s1 += g[i];
                        s2 += g[i] * xh[i];
                    }
                }
                s1 = sum_xor32(s1);
                s2 = sum_xor32(s2);
                const float m1 = s1 * (1.0f / 32.0f), m2 = s2 * (1.0f / 32.0f);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    g[i] = rstd * (g[i] - m1 - xh[i] * m2);   // dy
                    dbias[i] += g[i];
                }
                }
                TileIO<T>::store(p.dy + tb, lane, g);
                if (pp.dyT) TileT<T>::store((T*)pp.dyT + ft_tileT_base<T>(sbase + ts * 32, c0, pp.Spad), scr, lane, g);
            }
            butterfly_reduce32(stat, lane);
            const float db = butterfly_reduce16(dbias, lane);
            float* row = p.part + (int64_t)wrow * 3 * p.H;
            const int i = j & 15;                              // register index this lane ended up with
            const int c = c0 + (i & 3) + 8 * (i >> 2) + 4 * hi;
            row[(j >> 4) * p.H + c] = stat[0] * p.scale;       // j<16: dgamma, j>=16: dbeta
            if (j < 16) row[2 * p.H + c] = db;
        }
    }
    template <int TC, int TS, bool RING>
    __device__ static inline void run(const Params& pp, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, int wrow, const float* lpar, int lstride, unsigned char* scr, unsigned char* ring) {
        struct { const T* carry_in; T* carry_out; const T* xhat; const GnAux* aux; T* dy; float* part; int H; int64_t S_valid; float scale; } p =
            {(const T*)pp.carry_in, (T*)pp.carry_out, (const T*)pp.xhat, pp.aux, (T*)pp.dy, pp.part, pp.H, pp.S_valid, pp.drop_scale};
        constexpr bool PRECISE = sizeof(T) == 4;
        constexpr int NBLK = kTileBytes / 1024;
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        const int j = lane & 31, hi = lane >> 5;
        const bool has_carry = p.carry_in != nullptr;
        // DMA of sub-tile i = (tc, ts) into ring buffer i & 1
        auto prefetch = [&](int tc, int ts) __attribute__((always_inline)) {
            const int64_t tb = ft_tile_base<T>(sbase + ts * 32, cbase + tc * 32, p.H);
            unsigned char* dst = ring + ((tc * TS + ts) & 1) * 2 * kTileBytes;
#pragma unroll
            for (int h = 0; h < NBLK; ++h)
                __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const unsigned char*>(p.xhat + tb) + h * 1024 + lane * 16), (lptr_t)(dst + h * 1024), 16, 0, 0);
            if (has_carry) {
#pragma unroll
                for (int h = 0; h < NBLK; ++h)
                    __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const unsigned char*>(p.carry_in + tb) + h * 1024 + lane * 16), (lptr_t)(dst + kTileBytes + h * 1024), 16, 0, 0);
            }
        };
        if constexpr (RING) prefetch(0, 0);
#pragma unroll
        for (int tc = 0; tc < TC; ++tc) {
            const int c0 = cbase + tc * 32;
            // (gamma / beta are re-read from their LDS copy quad by quad: holding 16 + 16 of them would push the 256x256
            //  tile past 256 VGPRs)
            float stat[32];   // [0..15] dgamma, [16..31] dbeta (this lane's 16 channels), without the 1/(1-p) factor
            float dbias[16];
#pragma unroll
            for (int r = 0; r < 32; ++r) stat[r] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) dbias[r] = 0.f;
#pragma unroll
            for (int ts = 0; ts < TS; ++ts) {
                const int64_t s = sbase + ts * 32 + j;
                const int64_t tb = ft_tile_base<T>(sbase + ts * 32, c0, p.H);
                const GnAux rec = p.aux[gn_aux_index(sbase + ts * 32, c0 >> 5, p.H, lane)];
                const float rstd = rec.rstd;
                const uint32_t bits = s < p.S_valid ? rec.keep : 0u;
                float xh[16], g[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) g[r] = acc[tc][ts][r];
                if constexpr (RING) {
                    const int it = tc * TS + ts;                 // (compile-time after unrolling)
                    // sub-tile `it` has landed once everything but the pieces just issued for `it + 1` has returned (loads
                    // return in order; stores share the counter, so this also waits for the previous sub-tile's stores)
                    if (it + 1 < TC * TS) {
                        prefetch((it + 1) / TS, (it + 1) % TS);
                        if (has_carry) __builtin_amdgcn_s_waitcnt(waitcnt_vm(2 * NBLK));
                        else __builtin_amdgcn_s_waitcnt(waitcnt_vm(NBLK));
                    } else {
                        __builtin_amdgcn_s_waitcnt(waitcnt_vm(0));
                    }
                    asm volatile("" ::: "memory");
                    const T* src = reinterpret_cast<const T*>(ring + (it & 1) * 2 * kTileBytes);
                    TileIO<T>::load(src, lane, xh);
                    if (has_carry) {
                        float ci[16];
                        TileIO<T>::load(src + 1024, lane, ci);
#pragma unroll
                        for (int r = 0; r < 16; ++r) g[r] += ci[r];
                    }
                } else {
                    TileIO<T>::load(p.xhat + tb, lane, xh);
                    if (has_carry) {
                        float ci[16];
                        TileIO<T>::load(p.carry_in + tb, lane, ci);
#pragma unroll
                        for (int r = 0; r < 16; ++r) g[r] += ci[r];
                    }
                }
                if (p.carry_out) TileIO<T>::store(p.carry_out + tb, lane, g);
                if constexpr ((ABL & 8) == 0) {
                    float s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int cl = tc * 32 + 8 * q + 4 * hi;
                        const f32x4 g4 = *reinterpret_cast<const f32x4*>(lpar + cl);
                        const f32x4 e4 = *reinterpret_cast<const f32x4*>(lpar + lstride + cl);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 4 * q + r;
                            // keep decision as an all-ones / all-zeros word: one v_bfe_i32 + one v_and per element
                            const uint32_t m = bit_mask_rt(bits, i);
                            const float gg = __uint_as_float(__float_as_uint(g[i]) & m);
                            const float ds = (ABL & 2) ? 1.0f : (ACTRT ? dact_rt<PRECISE>(g4[r] * xh[i] + e4[r], pp.act) : dsilu_f<PRECISE>(g4[r] * xh[i] + e4[r]));
                            const float da = gg * ds;                  // (the 1/(1-p) factor is applied to gamma and to the sums)
                            if constexpr ((ABL & 1) == 0) {
                                stat[i] += da * xh[i];
                                stat[16 + i] += da;
                            }
                            g[i] = da * (g4[r] * p.scale);             // dx
                            s1 += g[i];
                            s2 += g[i] * xh[i];
                        }
                    }
                    s1 = sum_xor32(s1);
                    s2 = sum_xor32(s2);
                    const float m1 = s1 * (1.0f / 32.0f), m2 = s2 * (1.0f / 32.0f);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        g[i] = rstd * (g[i] - m1 - xh[i] * m2);   // dy
                        if constexpr ((ABL & 1) == 0) dbias[i] += g[i];
                    }
                }
                if constexpr (sizeof(T) == 4) {
                    if (p.dy) TileIO<T>::store(p.dy + tb, lane, g);
                    if (pp.dy_hi) store_tile_planes(pp.dy_hi, pp.dy_lo, sbase + ts * 32, c0, p.H, lane, g);
                } else {
                    TileIO<T>::store(p.dy + tb, lane, g);
                }
                if (pp.dyT) TileT<T>::store((T*)pp.dyT + ft_tileT_base<T>(sbase + ts * 32, c0, pp.Spad), scr, lane, g);
            }
            if constexpr ((ABL & 9) == 0) {
                // reduce over the 32 samples of the lane group; lane i ends with value i.
                float db;
                if constexpr (ABL & 4) { butterfly_reduce32_bperm(stat, lane); db = butterfly_reduce16_bperm(dbias, lane); }
                else { butterfly_reduce32(stat, lane); db = butterfly_reduce16(dbias, lane); }
                float* row = p.part + (int64_t)wrow * 3 * p.H;
                const int i = j & 15;                              // register index this lane ended up with
                const int c = c0 + (i & 3) + 8 * (i >> 2) + 4 * hi;
                row[(j >> 4) * p.H + c] = stat[0] * p.scale;       // j<16: dgamma, j>=16: dbeta
                if (j < 16) row[2 * p.H + c] = db;
            }
        }
    }
};

// dU = acc * silu'(u)  (backward of shared_time_embed's SiLU), FT store.
struct SiLUBwdParams {
    const void* pre;   // u, FT [Spad][N]
    void* out;         // dU, FT [Spad][N]
    int N;
    int64_t S_valid;
    void* outT;        // optional: FT [N][Spad]
    int64_t Spad;
    int act;           // DP_ACT_* (ACTRT instantiations)
    float* part;       // optional: [wave rows][N] column sums of dU over each wave's samples (the bias gradient's partials: a
                       // k_colsum launch over dU otherwise), summed like the GroupNorm-backward epilogue sums its bias partials
    DropoutCfg drop;   // the forward epilogue's dropout behind the activation (TimeMLPs): same counters, same decisions; p = 0: none
};
template <typename T, bool ACTRT = false> struct EpiSiLUBwd {
    typedef SiLUBwdParams Params;
    static constexpr int kScratchPerWave = TileT<T>::SCRATCH_BYTES;
    template <int TC, int TS>
    __device__ static inline void apply(const Params& pp, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, int wrow, int, const float* lpar, int lstride, unsigned char* scr) {
        struct { const T* pre; T* out; int N; int64_t S_valid; } p = {(const T*)pp.pre, (T*)pp.out, pp.N, pp.S_valid};
        constexpr bool PRECISE = sizeof(T) == 4;
        const int j = lane & 31, hi = lane >> 5;
#pragma unroll
        for (int tc = 0; tc < TC; ++tc) {
            float dsum[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) dsum[r] = 0.f;
#pragma unroll
            for (int ts = 0; ts < TS; ++ts) {
                const int64_t s = sbase + ts * 32 + j;
                const int64_t tb = ft_tile_base<T>(sbase + ts * 32, cbase + tc * 32, p.N);
                float u[16], o[16];
                TileIO<T>::load(p.pre + tb, lane, u);
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = (s < p.S_valid) ? acc[tc][ts][r] * (ACTRT ? dact_rt<PRECISE>(u[r], pp.act) : dsilu_f<PRECISE>(u[r])) : 0.f;
                if (pp.drop.p > 0.f) {
                    const uint32_t bits = dropout_bits16(pp.drop, s, (cbase + tc * 32) >> 5, hi);
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] = ((bits >> r) & 1u) ? o[r] * pp.drop.scale : 0.f;
                }
                TileIO<T>::store(p.out + tb, lane, o);
                if (pp.outT) TileT<T>::store((T*)pp.outT + ft_tileT_base<T>(sbase + ts * 32, cbase + tc * 32, pp.Spad), scr, lane, o);
#pragma unroll
                for (int r = 0; r < 16; ++r) dsum[r] += o[r];
            }
            if (pp.part) {      // column sums over the wave's samples: lane (j & 15) == i ends with the sum of register i (both halves)
                const float db = butterfly_reduce16(dsum, lane);
                const int i = j & 15;
                const int c = cbase + tc * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi;
                if (j < 16) pp.part[(int64_t)wrow * p.N + c] = db;
            }
        }
    }
};

// post_dense + the denoising-score-matching loss + d loss / d res in ONE launch (the fused training step): what EpiRowMajor -> `res`
// -> k_dsm did in two launches and a round trip of `res`.  Per element, in k_dsm's operation order (losses.py:121-131, utils.py:162,
// model.py:192-194):
//   model = (acc + bias) / used_sigma(t);  score = -model / std(t);  e = score * std(t) + z;  loss += e^2;
//   d res = (-2 e) * grad_scale / used_sigma(t)          (zero on padded samples / channels)
// t differs per sample (lane = sample): each lane derives std / used_sigma of its own sample once per 32-sample tile.
// Outputs: dres (FT, the operand of the backward GEMMs), per-wave column sums of the STORED dres (post_dense's bias gradient: summed
// over the wave rows by the step's reduction launch, like the GroupNorm-backward partials) and one loss partial per wave.
struct DsmStepParams {
    const float* bias;     // [D] post_dense bias
    const float* t;        // [Spad] t of every sample
    const float* z;        // [Spad][Dpad] fp32 row-major: the noise k_prep_train kept
    const float* sigmas;
    void* dres;            // FT [Spad][Cp]
    float* loss_part;      // [wave rows * waves along the channels]
    float* cs_part;        // [wave rows][Cp]
    SdeDev sde;
    float grad_scale;
    int num_scales, scale_by_sigma, fourier;
    int D, Dpad, Cp;
    int64_t S_valid;
};
template <typename T> struct EpiDsm {
    typedef DsmStepParams Params;
    template <int TC, int TS>
    __device__ static inline void apply(const Params& p, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, int wrow, int, const float*, int, unsigned char*) {
#pragma clang fp contract(off)
        const int j = lane & 31, hi = lane >> 5;
        float lsum = 0.f;
        float csum[TC][16];
#pragma unroll
        for (int tc = 0; tc < TC; ++tc)
#pragma unroll
            for (int r = 0; r < 16; ++r) csum[tc][r] = 0.f;
#pragma unroll
        for (int ts = 0; ts < TS; ++ts) {
            const int64_t s = sbase + ts * 32 + j;
            const bool live = s < p.S_valid;
            const float t = live ? p.t[s] : 0.5f;
            const float lmc = sde_lmc(p.sde, t);
            const float sd = sde_std(p.sde, lmc);
            const float usig = p.scale_by_sigma ? used_sigma(p.sigmas, p.num_scales, t * 999.0f, p.fourier) : 1.0f;
#pragma unroll
            for (int tc = 0; tc < TC; ++tc) {
                const int c0 = cbase + tc * 32;
                float o[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = c0 + 8 * q + 4 * hi;
                    f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
                    if (live && c < p.Dpad) z4 = *reinterpret_cast<const f32x4*>(p.z + s * p.Dpad + c);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 4 * q + r;
                        const bool ok = live && c + r < p.D;
                        const float model = (acc[tc][ts][i] + (c + r < p.D ? p.bias[c + r] : 0.f)) / usig;
                        const float score = -model / sd;                                   // utils.py:162
                        const float e = score * sd + z4[r];                                // losses.py:124
                        if (ok) lsum += e * e;
                        o[i] = ok ? (-2.0f * e) * p.grad_scale / usig : 0.f;
                    }
                }
                TileIO<T>::store((T*)p.dres + ft_tile_base<T>(sbase + ts * 32, c0, p.Cp), lane, o);
#pragma unroll
                for (int r = 0; r < 16; ++r) csum[tc][r] += (float)(T)o[r];               // what the backward GEMMs read
            }
        }
#pragma unroll
        for (int tc = 0; tc < TC; ++tc) {
            const float db = butterfly_reduce16(csum[tc], lane);
            const int i = j & 15;
            const int c = cbase + tc * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi;
            if (j < 16) p.cs_part[(int64_t)wrow * p.Cp + c] = db;
        }
        // one loss partial per wave: lanes 0..31 / 32..63 by row-shift adds, then the two halves
        float v = lsum;
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        v = sum_xor32(v);
        const int chan_waves = p.Cp / (TC * 32);
        if (lane == 0) p.loss_part[(int64_t)wrow * chan_waves + (cbase / (TC * 32)) % chan_waves] = v * p.grad_scale;
    }
};

// wgrad: slab[split][n][k] = acc   (fp32 row-major, lane = k column => 128-B coalesced rows)
struct WgradParams {
    float* slab;            // base of this parameter inside slab 0
    int64_t slab_stride;    // elements between consecutive split slabs
    int ld;                 // = K_valid (row length of the parameter)
    int N_valid, K_valid;
};
template <typename T> struct EpiWgrad {
    typedef WgradParams Params;
    template <int TC, int TS>
    __device__ static inline void apply(const Params& p, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, int, int split, const float*, int, unsigned char*) {
        const int j = lane & 31, hi = lane >> 5;
        float* base = p.slab + (int64_t)split * p.slab_stride;
        // wave tile completely inside the matrix and the slab below 4 GB (every training wgrad): one 32-bit byte offset per lane, row /
        // tile steps as scalar multiples of ld -- no per-element 64-bit address product, no per-element bounds test (the general loop
        // below costs ~4 VALU per stored value: 520 of this epilogue's ~1000 instructions per wave)
        if ((int)sbase + TS * 32 <= p.K_valid && cbase + TC * 32 <= p.N_valid && (uint64_t)p.N_valid * (uint64_t)p.ld < (1ull << 30)) {
            const uint32_t off0 = ((uint32_t)(cbase + 4 * hi) * (uint32_t)p.ld + (uint32_t)sbase + (uint32_t)j) * 4u;
            char* b0 = reinterpret_cast<char*>(base);
#pragma unroll
            for (int tc = 0; tc < TC; ++tc)
#pragma unroll
                for (int ts = 0; ts < TS; ++ts)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const uint32_t o = off0 + ((uint32_t)(tc * 32 + (r & 3) + 8 * (r >> 2)) * (uint32_t)p.ld + (uint32_t)(ts * 32)) * 4u;
                        *reinterpret_cast<float*>(b0 + o) = acc[tc][ts][r];
                    }
            return;
        }
#pragma unroll
        for (int tc = 0; tc < TC; ++tc)
#pragma unroll
            for (int ts = 0; ts < TS; ++ts) {
                const int k = (int)sbase + ts * 32 + j;
                if (k < p.K_valid) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = cbase + tc * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                        if (n < p.N_valid) base[(int64_t)n * p.ld + k] = acc[tc][ts][r];
                    }
                }
            }
    }
};

// ----------------------------------------------------------------------------------------------
// GroupNorm with 16 or 64 channels per group (hidden_dim 512 / 2048 under nn.GroupNorm(32, H), model.py:112)
// ----------------------------------------------------------------------------------------------
// EpiGN / EpiGNBwd above are the shipped hidden_dim = 1024 case: one 32x32 accumulator tile IS one group.  These generic forms
// cover the neighbouring sizes with the same register layout:
//   GS = 16: a tile holds TWO groups -- registers 0..7 of both lane halves are channels 0..15, registers 8..15 are 16..31 -- so
//            every statistic exists twice per tile and still needs one lane^32 exchange each;
//   GS = 64: a group spans the two adjacent 32-channel tiles (2k, 2k+1) of a wave, whose accumulators live in the same lane:
//            sums run over both tiles before the exchange.
// No residual / operand prefetch here: these sizes are not the benchmark configuration; correctness and the same fusion matter.
struct GnAuxG { float rstd[2]; uint32_t keep; uint32_t pad; };   // GS = 16: rstd of both groups; GS = 64: rstd[0]

template <typename T, bool TRAIN, int GS> struct EpiGNG {
    static_assert(GS == 16 || GS == 64, "generic GroupNorm epilogue: 16 or 64 channels per group");
    typedef GNParams Params;                  // (aux points at GnAuxG records)
    static constexpr int kScratchPerWave = TRAIN ? TileT<T>::SCRATCH_BYTES : 0;
    static constexpr int kParamArrays = 3;
    static constexpr int NT = GS == 64 ? 2 : 1;          // tiles per statistics set
    static constexpr int NG = GS == 16 ? 2 : 1;          // groups per statistics set
    __device__ static inline const float* param_array(const Params& p, int a) { return a == 0 ? p.bias : (a == 1 ? p.gamma : p.beta); }
    template <int TC, int TS>
    __device__ static inline void apply(const Params& pp, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, int, int, const float* lpar, int lstride, unsigned char* scr) {
        static_assert(TC % NT == 0, "a 64-channel group needs both of its tiles in one wave");
        constexpr bool PRECISE = sizeof(T) == 4;
        const int j = lane & 31, hi = lane >> 5;
        T* out = (T*)pp.out;
        const T* resid = (const T*)pp.resid;
        T* xhat = (T*)pp.xhat;
        GnAuxG* aux = reinterpret_cast<GnAuxG*>(pp.aux);
        const bool drop = TRAIN && pp.drop.p > 0.f;
#pragma unroll
        for (int ts = 0; ts < TS; ++ts) {
            const int64_t s = sbase + ts * 32 + j;
#pragma unroll
            for (int t0 = 0; t0 < TC; t0 += NT) {
                float v[NT][16];
                float sum[NG] = {};
#pragma unroll
                for (int u = 0; u < NT; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(lpar + (t0 + u) * 32 + 8 * q + 4 * hi);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            v[u][4 * q + r] = acc[t0 + u][ts][4 * q + r] + b4[r];
                            sum[GS == 16 ? (q >> 1) : 0] += v[u][4 * q + r];
                        }
                    }
                float mean[NG], rstd[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) mean[g] = sum_xor32(sum[g]) * (1.0f / GS);
                float ss[NG] = {};
#pragma unroll
                for (int u = 0; u < NT; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int g = GS == 16 ? (i >> 3) : 0;
                        v[u][i] -= mean[g];
                        ss[g] += v[u][i] * v[u][i];
                    }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const float var = sum_xor32(ss[g]) * (1.0f / GS);
                    rstd[g] = PRECISE ? 1.0f / sqrtf(var + 1e-5f) : rsqrtf(var + 1e-5f);
                }
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    const int c0 = cbase + (t0 + u) * 32;
                    const int64_t tb = ft_tile_base<T>(sbase + ts * 32, c0, pp.H);
                    float keep[16];
                    uint32_t bits = 0xffffu;
                    if (drop) bits = dropout_mask16_bits(pp.drop, s, c0 >> 5, hi, keep);
                    if (TRAIN) {
                        GnAuxG rec = {{rstd[0], rstd[NG - 1]}, bits, 0u};
                        aux[gn_aux_index(sbase + ts * 32, c0 >> 5, pp.H, lane)] = rec;
                    }
                    float o[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int cl = (t0 + u) * 32 + 8 * q + 4 * hi;
                        const f32x4 g4 = *reinterpret_cast<const f32x4*>(lpar + lstride + cl);
                        const f32x4 e4 = *reinterpret_cast<const f32x4*>(lpar + 2 * lstride + cl);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 4 * q + r;
                            v[u][i] *= rstd[GS == 16 ? (i >> 3) : 0];                         // x_hat
                            float y = silu_f<PRECISE>(g4[r] * v[u][i] + e4[r]);
                            if (drop) y *= keep[i];
                            o[i] = y;
                        }
                    }
                    if (TRAIN) TileIO<T>::store(xhat + tb, lane, v[u]);
                    if (resid) {
                        float rr[16];
                        TileIO<T>::load(resid + tb, lane, rr);
#pragma unroll
                        for (int i = 0; i < 16; ++i) o[i] += rr[i];
                    }
                    if constexpr (sizeof(T) == 4) {       // (bf16x3 mode: `out` is optional, the operand planes beside or instead of it)
                        if (out) TileIO<T>::store(out + tb, lane, o);
                        if (pp.out_hi) store_tile_planes(pp.out_hi, pp.out_lo, sbase + ts * 32, c0, pp.H, lane, o);
                    } else {
                        TileIO<T>::store(out + tb, lane, o);
                    }
                    if (TRAIN && pp.outT) TileT<T>::store((T*)pp.outT + ft_tileT_base<T>(sbase + ts * 32, c0, pp.Spad), scr, lane, o);
                }
            }
        }
    }
};

template <typename T, int GS> struct EpiGNBwdG {
    static_assert(GS == 16 || GS == 64, "generic GroupNorm-backward epilogue: 16 or 64 channels per group");
    typedef GNBwdParams Params;               // (aux points at GnAuxG records)
    static constexpr int kScratchPerWave = TileT<T>::SCRATCH_BYTES;
    static constexpr int kParamArrays = 2;
    static constexpr int NT = GS == 64 ? 2 : 1;
    static constexpr int NG = GS == 16 ? 2 : 1;
    __device__ static inline const float* param_array(const Params& p, int a) { return a == 0 ? p.gamma : p.beta; }
    template <int TC, int TS>
    __device__ static inline void apply(const Params& pp, f32x16 (&acc)[TC][TS], int cbase, int64_t sbase, int lane, int wrow, int, const float* lpar, int lstride, unsigned char* scr) {
        static_assert(TC % NT == 0, "a 64-channel group needs both of its tiles in one wave");
        constexpr bool PRECISE = sizeof(T) == 4;
        const int j = lane & 31, hi = lane >> 5;
        const T* carry_in = (const T*)pp.carry_in;
        T* carry_out = (T*)pp.carry_out;
        const T* xhat = (const T*)pp.xhat;
        T* dy = (T*)pp.dy;
        const GnAuxG* aux = reinterpret_cast<const GnAuxG*>(pp.aux);
#pragma unroll
        for (int t0 = 0; t0 < TC; t0 += NT) {
            float stat[NT][32], dbias[NT][16];
#pragma unroll
            for (int u = 0; u < NT; ++u) {
#pragma unroll
                for (int r = 0; r < 32; ++r) stat[u][r] = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) dbias[u][r] = 0.f;
            }
#pragma unroll
            for (int ts = 0; ts < TS; ++ts) {
                const int64_t s = sbase + ts * 32 + j;
                float xh[NT][16], g[NT][16], rstd[NG];
                float s1[NG] = {}, s2[NG] = {};
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    const int c0 = cbase + (t0 + u) * 32;
                    const int64_t tb = ft_tile_base<T>(sbase + ts * 32, c0, pp.H);
                    const GnAuxG rec = aux[gn_aux_index(sbase + ts * 32, c0 >> 5, pp.H, lane)];
                    if (u == 0) { rstd[0] = rec.rstd[0]; rstd[NG - 1] = rec.rstd[NG - 1]; }
                    const uint32_t bits = s < pp.S_valid ? rec.keep : 0u;
#pragma unroll
                    for (int r = 0; r < 16; ++r) g[u][r] = acc[t0 + u][ts][r];
                    TileIO<T>::load(xhat + tb, lane, xh[u]);
                    if (carry_in) {
                        float ci[16];
                        TileIO<T>::load(carry_in + tb, lane, ci);
#pragma unroll
                        for (int r = 0; r < 16; ++r) g[u][r] += ci[r];
                    }
                    if (carry_out) TileIO<T>::store(carry_out + tb, lane, g[u]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int cl = (t0 + u) * 32 + 8 * q + 4 * hi;
                        const f32x4 g4 = *reinterpret_cast<const f32x4*>(lpar + cl);
                        const f32x4 e4 = *reinterpret_cast<const f32x4*>(lpar + lstride + cl);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = 4 * q + r;
                            const uint32_t m = bit_mask_rt(bits, i);
                            const float gg = __uint_as_float(__float_as_uint(g[u][i]) & m);
                            const float da = gg * dsilu_f<PRECISE>(g4[r] * xh[u][i] + e4[r]);
                            stat[u][i] += da * xh[u][i];
                            stat[u][16 + i] += da;
                            g[u][i] = da * (g4[r] * pp.drop_scale);                            // dx
                            s1[GS == 16 ? (i >> 3) : 0] += g[u][i];
                            s2[GS == 16 ? (i >> 3) : 0] += g[u][i] * xh[u][i];
                        }
                    }
                }
                float m1[NG], m2[NG];
#pragma unroll
                for (int k = 0; k < NG; ++k) { m1[k] = sum_xor32(s1[k]) * (1.0f / GS); m2[k] = sum_xor32(s2[k]) * (1.0f / GS); }
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    const int c0 = cbase + (t0 + u) * 32;
                    const int64_t tb = ft_tile_base<T>(sbase + ts * 32, c0, pp.H);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int k = GS == 16 ? (i >> 3) : 0;
                        g[u][i] = rstd[k] * (g[u][i] - m1[k] - xh[u][i] * m2[k]);              // dy
                        dbias[u][i] += g[u][i];
                    }
                    if constexpr (sizeof(T) == 4) {
                        if (dy) TileIO<T>::store(dy + tb, lane, g[u]);
                        if (pp.dy_hi) store_tile_planes(pp.dy_hi, pp.dy_lo, sbase + ts * 32, c0, pp.H, lane, g[u]);
                    } else {
                        TileIO<T>::store(dy + tb, lane, g[u]);
                    }
                    if (pp.dyT) TileT<T>::store((T*)pp.dyT + ft_tileT_base<T>(sbase + ts * 32, c0, pp.Spad), scr, lane, g[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                butterfly_reduce32(stat[u], lane);
                const float db = butterfly_reduce16(dbias[u], lane);
                float* row = pp.part + (int64_t)wrow * 3 * pp.H;
                const int i = j & 15;
                const int c = cbase + (t0 + u) * 32 + (i & 3) + 8 * (i >> 2) + 4 * hi;
                row[(j >> 4) * pp.H + c] = stat[u][0] * pp.drop_scale;
                if (j < 16) row[2 * pp.H + c] = db;
            }
        }
    }
};
