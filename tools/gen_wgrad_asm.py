#!/usr/bin/env python3
"""Generates dposer_amd/csrc/gemm_wgrad_tr_asm.inc: the hand-placed K loop of gemm_wgrad_tr_kernel<2, 4, 4, 2, 4> as ONE asm statement
(string literal + operand lists).        python tools/gen_wgrad_asm.py > dposer_amd/csrc/gemm_wgrad_tr_asm.inc

Why a generator: an MFMA operand is four consecutive VGPRs and a transposing LDS read (ds_read_b64_tr_b16) fills two of them -- inline asm
cannot name half of a 4-register operand, so the fragments live in FIXED registers (v208 ... v255, on the clobber list) and the whole
loop (entry reads, the `rem` leading stages, the groups of four, the three tail stages) is one statement: nothing the compiler schedules
can sit between two uses of those registers.  Every LDS offset / M0 value is a number in the text (the ring slot of a stage is fixed by
its position in the loop body).

Ring: 4 slots x 32 KiB; slot = [A blocks 16 KiB | B blocks 16 KiB]; a stage = one 32-sample block row = two k-blocks of 16 samples.
Per wave and stage: 16 MFMA (4 x 2 tiles x 2 k-blocks), 24 transposing reads, 4 DMA pieces (2 x dY, 2 x H) for stage t + 3.
Stage contract = gemm_kloop_asm.h (same waits: vmcnt(6) in front of the barrier of a fetching stage, 4 / 0 in the tail)."""

F0A = [208, 212, 216, 220]      # fragments of k-block 0 (A = dY tiles, B = H tiles)
F0B = [224, 228]
F1A = [232, 236, 240, 244]      # k-block 1
F1B = [248, 252]
ACC = [["c00", "c01"], ["c10", "c11"], ["c20", "c21"], ["c30", "c31"]]


def vreg(base, n=4):
    return f"v[{base}:{base + n - 1}]"


def mfma(i, j, fa, fb):
    c = ACC[i][j]
    return f"v_mfma_f32_32x32x16_bf16 %[{c}], {vreg(fa[i])}, {vreg(fb[j])}, %[{c}]"


def read(op, slot, kb, tile, dst):
    """The two transposing reads of one fragment.  Address VGPRs (per operand, per 64 KiB half of the ring): <op>0 covers k-block 0
    (lo, hi = +128) and the lo read of k-block 1 (+512); <op>1 is the hi read of k-block 1 (its +128 wraps inside the 1-KiB block for
    the lanes with kh = gq = 1, so it is a separate per-lane address)."""
    bank = "h" if slot >= 2 else "l"
    rel = (slot & 1) * 32768 + tile * 2048
    if kb == 0:
        lo = (f"%[{op}0{bank}]", rel)
        hi = (f"%[{op}0{bank}]", rel + 128)
    else:
        lo = (f"%[{op}0{bank}]", rel + 512)
        hi = (f"%[{op}1{bank}]", rel)
    return [f"ds_read_b64_tr_b16 {vreg(dst, 2)}, {lo[0]} offset:{lo[1]}", f"ds_read_b64_tr_b16 {vreg(dst + 2, 2)}, {hi[0]} offset:{hi[1]}"]


def half(slot_reads, kb_reads, fa, fb, ra, rb, dma, wait, barrier):
    """8 MFMAs on (fa, fb); the reads of k-block kb_reads of slot slot_reads into (ra, rb) (None: no reads); dma = (m0 values, offset
    vgpr, base sgprs, stride sgpr) or None."""
    out = []
    order = [("a", 0), ("b", 0), ("b", 1), ("a", 1), ("a", 2), ("a", 3)]          # in the order the next half consumes them
    k = 0
    for i in range(4):
        for j in range(2):
            out.append(mfma(i, j, fa, fb))
            if slot_reads is not None and k < 6:
                op, t = order[k]
                out += read(op, slot_reads, kb_reads, t, (ra if op == "a" else rb)[t])
            if dma is not None:
                m0s, vofs, bases, stride = dma
                if k == 2:
                    out.append(f"s_add_i32 m0, %[sm0], {m0s[0]}")
                if k == 3:
                    out.append(f"global_load_lds_dwordx4 %[{vofs}], %[{bases[0]}]")
                if k == 4:
                    out.append(f"s_add_i32 m0, %[sm0], {m0s[1]}")
                if k == 5:
                    out.append(f"global_load_lds_dwordx4 %[{vofs}], %[{bases[1]}]")
                    out.append(f"v_add_u32 %[{vofs}], %[{stride}], %[{vofs}]")
            if k == 6 and wait:
                out.append(wait)
            if k == 7 and barrier:
                out.append("s_barrier")
            k += 1
    return out


def stage(S, mode):
    """mode 0: fetching stage; 1 / 2: third-last / second-last (no DMA, vmcnt 4 / 0); 3: last."""
    S1, D = (S + 1) & 3, (S + 3) & 3
    M = D * 32768
    out = [f"; ---- stage on slot {S}, mode {mode}"]
    dma_a = ((M, M + 8192), "vaofs", ("sA0", "sA1"), "strA") if mode == 0 else None
    dma_b = ((M + 16384, M + 24576), "vbofs", ("sB0", "sB1"), "strB") if mode == 0 else None
    if mode == 3:
        out += half(S, 1, F0A, F0B, F1A, F1B, None, "s_waitcnt lgkmcnt(0)", False)
        out += half(None, 0, F1A, F1B, None, None, None, None, False)
        return out
    wait = {0: "s_waitcnt vmcnt(6) lgkmcnt(0)", 1: "s_waitcnt vmcnt(4) lgkmcnt(0)", 2: "s_waitcnt vmcnt(0) lgkmcnt(0)"}[mode]
    out += half(S, 1, F0A, F0B, F1A, F1B, dma_a, wait, True)
    out += half(S1, 0, F1A, F1B, F0A, F0B, dma_b, None, False)
    out.append("s_waitcnt lgkmcnt(0)")
    return out


def entry_reads(S):
    out = [f"; ---- k-block 0 of slot {S} (the stage the loop starts with)"]
    for op, t in [("a", 0), ("b", 0), ("b", 1), ("a", 1), ("a", 2), ("a", 3)]:
        out += read(op, S, 0, t, (F0A if op == "a" else F0B)[t])
    out.append("s_waitcnt lgkmcnt(0)")
    return out


def main():
    L = []
    # entry: the ring starts at slot (4 - rem) & 3, so that the groups of four start on slot 0
    L += ["s_cmp_eq_u32 %[rem], 3", "s_cbranch_scc0 Ln3_%="] + entry_reads(1) + ["s_branch Ls1_%=", "Ln3_%=:"]
    L += ["s_cmp_eq_u32 %[rem], 2", "s_cbranch_scc0 Ln2_%="] + entry_reads(2) + ["s_branch Ls2_%=", "Ln2_%=:"]
    L += ["s_cmp_eq_u32 %[rem], 1", "s_cbranch_scc0 Ln1_%="] + entry_reads(3) + ["s_branch Ls3_%=", "Ln1_%=:"]
    L += entry_reads(0) + ["s_branch Ls0_%="]
    L += ["Ls1_%=:"] + stage(1, 0) + ["Ls2_%=:"] + stage(2, 0) + ["Ls3_%=:"] + stage(3, 0)
    L += ["Ls0_%=:", "s_cmp_eq_u32 %[grp], 0", "s_cbranch_scc1 Lt_%=", "Lg_%=:"]
    for S in range(4):
        L += stage(S, 0)
    L += ["s_sub_u32 %[grp], %[grp], 1", "s_cmp_lg_u32 %[grp], 0", "s_cbranch_scc1 Lg_%=", "Lt_%=:"]
    L += stage(0, 1) + stage(1, 2) + stage(2, 3)
    # the compiler does not know that the statement ends on MFMAs: the wait states between an 8-pass XDL write and a VALU read of its
    # result (the hazard recognizer inserts them for MFMAs it sees) are spent here
    L += ["s_nop 7", "s_nop 7"]
    print("// GENERATED by tools/gen_wgrad_asm.py -- do not edit.  The K loop of gemm_wgrad_tr_kernel<2, 4, 4, 2, 4> (see the generator's header).")
    print("asm volatile(")
    for line in L:
        if line.startswith(";"):
            print(f"    // {line[2:]}")
        else:
            print(f'    "{line}\\n"')
    print("    : [c00] \"+v\"(acc[0][0]), [c01] \"+v\"(acc[0][1]), [c10] \"+v\"(acc[1][0]), [c11] \"+v\"(acc[1][1]), [c20] \"+v\"(acc[2][0]),")
    print("      [c21] \"+v\"(acc[2][1]), [c30] \"+v\"(acc[3][0]), [c31] \"+v\"(acc[3][1]), [vaofs] \"+v\"(v_aofs), [vbofs] \"+v\"(v_bofs), [grp] \"+s\"(grp)")
    print("    : [a0l] \"v\"(vA0_lo), [a1l] \"v\"(vA1_lo), [a0h] \"v\"(vA0_hi), [a1h] \"v\"(vA1_hi), [b0l] \"v\"(vB0_lo), [b1l] \"v\"(vB1_lo), [b0h] \"v\"(vB0_hi),")
    print("      [b1h] \"v\"(vB1_hi), [sA0] \"s\"(sA[0]), [sA1] \"s\"(sA[1]), [sB0] \"s\"(sB[0]), [sB1] \"s\"(sB[1]), [sm0] \"s\"(s_m0), [strA] \"s\"(strA), [strB] \"s\"(strB),")
    print("      [rem] \"s\"(rem)")
    clob = ", ".join(f'"v{r}"' for r in range(208, 256))
    print(f"    : \"memory\", \"scc\", {clob});")


if __name__ == "__main__":
    main()
