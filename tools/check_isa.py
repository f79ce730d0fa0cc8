#!/usr/bin/env python3
"""Audits of the ISA hipcc emits for the library, for the things the compiler cannot see inside `asm` statements
(run after touching any kernel with inline asm; `python tools/check_isa.py [--keep DIR]`, ~2-3 min: every .hip of dposer_amd/csrc is
cross-compiled to gfx950 ISA with the Makefile's flags, in parallel).

1. K-loop accumulators (gemm_launch.hip, gemm_sampler.hip): the MFMAs of the hand-placed stage statements are invisible to hipcc's
   hazard recogniser, so between two neighbouring stage statements nothing may read or write an ACCUMULATOR register (a register copy
   there would read an XDL result before its wait states have passed), and those kernels must not spill.

2. M0 (every kernel of every file).  M0 holds the LDS destination of `global_load_lds` / `buffer_load ... lds`; it is compiler-reserved,
   an "m0" entry in an asm clobber list is NOT honoured (hipcc only warns), and hipcc does not know that an asm statement wrote it.  So:
     * every compiler-issued M0 reader (global_load_lds*, buffer_load* ... lds, s_movrel* / v_movrel*, ds_gws*, s_sendmsg*, ds_* gds, or
       m0 as a source operand) must be reached, on EVERY path, with M0 last written by the compiler itself -- not by an asm statement
       (a forward may-analysis over the kernel's control-flow graph: state "asm wrote M0 last" reaching such a reader is a violation);
     * every asm-issued M0 reader must follow an M0 write inside the SAME asm statement.
   An asm statement that saves M0 first and restores it last (s_mov_b32 sN, m0 ... s_mov_b32 m0, sN) counts as not writing it.

Prints one line per kernel that contains asm; exits non-zero on a violation."""
import concurrent.futures
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dposer_amd", "csrc")
CONTRACT_OK = {"gemm_launch.hip", "gemm_launch_x3.hip", "gemm_sampler.hip"}          # (Makefile: everything else is built with -ffp-contract=off)

# the opt-in persistent / cluster samplers (DPOSER_SAMPLER_PERSISTENT=1 / =2, off by default: measured slower, profiles/r05_sampler_small_ab.txt)
# keep 3..8 registers of their prologue in scratch (12..28 bytes per lane); nothing between their stage statements touches scratch (that
# check still applies to them)
SPILL_NOTED = re.compile(r"k_sampler_(persistent|cluster)")
M0_READER = re.compile(r"^(global_load_lds|s_movrel|v_movrel|ds_gws|s_sendmsg)")


def compile_isa(name, outdir):
    out = os.path.join(outdir, name.replace(".hip", ".s"))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{CSRC}", f"-I{ROOT}/include", "--cuda-device-only", "-S",
           os.path.join(CSRC, name), "-o", out]
    if name not in CONTRACT_OK:
        cmd.insert(5, "-ffp-contract=off")
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {name}:\n{r.stderr[-2000:]}")
    return name, out


def kernels(text):
    """(name, body) of every function of the ISA text: from its label to its .Lfunc_end."""
    for m in re.finditer(r"^(_Z\S+|[A-Za-z_]\w*):\s*;\s*@\1\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        yield m.group(1), m.group(2)


def instructions(body):
    """[(mnemonic, operand text, in_asm, asm_block_id, label)] in program order; labels come as ('', '', False, -1, name)."""
    out, in_asm, blk = [], False, -1
    for raw in body.splitlines():
        line = raw.strip()
        if line.startswith(";;#ASMSTART"):
            in_asm, blk = True, blk + 1
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            out.append(("", "", False, -1, m.group(1)))
            continue
        line = line.split(";")[0].strip()
        if not line or line.startswith(".") or line.endswith(":"):
            continue
        parts = line.split(None, 1)
        out.append((parts[0], parts[1] if len(parts) > 1 else "", in_asm, blk if in_asm else -1, None))
    return out


def writes_m0(mn, ops):
    return bool(re.match(r"^m0\b", ops.strip())) and not mn.startswith(("s_cmp", "s_bitcmp", "s_waitcnt"))


def reads_m0(mn, ops):
    if M0_READER.match(mn):
        return True
    if mn.startswith("buffer_load") and re.search(r"\blds\b", ops):
        return True
    if mn.startswith("ds_") and re.search(r"\bgds\b", ops):
        return True
    srcs = ops.split(",", 1)[1] if "," in ops else ""
    return bool(re.search(r"\bm0\b", srcs))


def audit_m0(body):
    """-> (violations, n_asm_m0_writes, n_compiler_readers)."""
    ins = instructions(body)
    # asm statements that save and restore M0 do not count as writers
    preserving = set()
    by_blk = {}
    for i, (mn, ops, in_asm, blk, lab) in enumerate(ins):
        if in_asm:
            by_blk.setdefault(blk, []).append((mn, ops))
    for blk, lst in by_blk.items():
        w = [(mn, ops) for mn, ops in lst if writes_m0(mn, ops)]
        first = lst[0]
        m = re.match(r"^(s\d+),\s*m0$", first[1].strip()) if first[0] == "s_mov_b32" else None
        if w and m and w[-1][0] == "s_mov_b32" and w[-1][1].replace(" ", "") == f"m0,{m.group(1)}":
            preserving.add(blk)
    # basic blocks
    leaders = {0}
    label_at = {}
    for i, (mn, ops, in_asm, blk, lab) in enumerate(ins):
        if lab:
            leaders.add(i)
            label_at[lab] = i
        elif mn.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            leaders.add(i + 1)
    leaders = sorted(x for x in leaders if x < len(ins))
    bb_of = {}
    for b, start in enumerate(leaders):
        end = leaders[b + 1] if b + 1 < len(leaders) else len(ins)
        for i in range(start, end):
            bb_of[i] = b
    succ = {b: set() for b in range(len(leaders))}
    for b, start in enumerate(leaders):
        end = leaders[b + 1] if b + 1 < len(leaders) else len(ins)
        last = None
        for i in range(end - 1, start - 1, -1):
            if not ins[i][4]:
                last = ins[i]
                break
        fall = True
        if last:
            mn, ops = last[0], last[1]
            if mn.startswith(("s_cbranch", "s_branch")):
                tgt = ops.strip().split()[0] if ops.strip() else ""
                if tgt in label_at:
                    succ[b].add(bb_of[label_at[tgt]])
                if mn.startswith("s_branch"):
                    fall = False
            elif mn.startswith(("s_endpgm", "s_setpc")):
                fall = False
        if fall and b + 1 < len(leaders):
            succ[b].add(b + 1)
    # forward may-analysis: state True = "an asm statement wrote M0 last on some path"
    n = len(leaders)
    st_in = [False] * n
    viol = []
    n_asm_w = sum(1 for mn, ops, in_asm, blk, lab in ins if in_asm and writes_m0(mn, ops) and blk not in preserving)
    n_rd = 0

    def run_block(b, state, record):
        nonlocal n_rd
        start = leaders[b]
        end = leaders[b + 1] if b + 1 < n else len(ins)
        seen_write_in_blk = {}
        for i in range(start, end):
            mn, ops, in_asm, blk, lab = ins[i]
            if lab:
                continue
            if in_asm:
                if reads_m0(mn, ops) and not writes_m0(mn, ops):
                    if record and not seen_write_in_blk.get(blk, False) and blk not in preserving:
                        viol.append(f"asm statement reads M0 it did not write itself: {mn} {ops}")
                if writes_m0(mn, ops):
                    seen_write_in_blk[blk] = True
                    if blk not in preserving:
                        state = True
            else:
                if reads_m0(mn, ops) and not writes_m0(mn, ops):
                    if record:
                        n_rd += 1
                        if state:
                            viol.append(f"compiler-issued M0 reader behind an asm M0 write: {mn} {ops}")
                if writes_m0(mn, ops):
                    state = False
        return state

    work = list(range(n))
    st_out = [None] * n
    while work:
        b = work.pop(0)
        o = run_block(b, st_in[b], False)
        if st_out[b] != o:
            st_out[b] = o
            for s2 in succ[b]:
                if o and not st_in[s2]:
                    st_in[s2] = True
                if s2 not in work:
                    work.append(s2)
    for b in range(n):
        run_block(b, st_in[b], True)
    return viol, n_asm_w, n_rd


def regs(line):
    out = set()
    for r in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", line):
        out.update(range(int(r.group(1)), int(r.group(2)) + 1) if r.group(1) else [int(r.group(3))])
    return out


def audit_kloop(body, name=""):
    """-> None when the kernel has no stage statements, else (violations, n_stage_statements)."""
    blocks = [b for b in re.finditer(r";;#ASMSTART\n(.*?);;#ASMEND", body, re.S) if b.group(1).count("v_mfma") >= 8]
    if not blocks:
        return None
    acc = set()
    for d in re.finditer(r"v_mfma\S+ v\[(\d+):(\d+)\]", blocks[0].group(1)):
        acc.update(range(int(d.group(1)), int(d.group(2)) + 1))
    viol = []
    # straight-line neighbours only: the group of four inside the loop (statements 3..6 of 10) and the three tail stages (7..9); the
    # text between other pairs holds other paths' code (accumulator zero-fill of the short-K fallback)
    pairs = [(3, 4), (4, 5), (5, 6), (7, 8), (8, 9)] if len(blocks) == 10 else []
    for i, j in pairs:
        between = body[blocks[i].end():blocks[j].start()]
        for line in between.splitlines():
            ins = line.strip().split()
            if ins and not ins[0].startswith(";") and (regs(line) & acc or ins[0].startswith("scratch_")):
                viol.append("between stages: " + line.strip())
    if "scratch_" in body and not SPILL_NOTED.search(name):
        viol.append("kernel spills (scratch_ instructions)")
    return viol, len(blocks)


def main():
    keep = sys.argv[sys.argv.index("--keep") + 1] if "--keep" in sys.argv else None
    files = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        outdir = keep or tmp
        os.makedirs(outdir, exist_ok=True)
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
            done = list(ex.map(lambda f: compile_isa(f, outdir), files))
        for name, path in done:
            text = open(path).read()
            n_k = n_asm = 0
            for kname, body in kernels(text):
                n_k += 1
                if ";;#ASMSTART" not in body and "global_load_lds" not in body:
                    continue
                n_asm += 1
                viol, n_w, n_rd = audit_m0(body)
                kl = audit_kloop(body, kname)
                if kl:
                    viol += kl[0]
                tag = f"{kl[1]:3d} stage statements" if kl else "                    "
                print(f"{'FAIL' if viol else 'ok  '} {name:18s} {tag}  asm M0 writes {n_w:3d}  compiler M0 readers {n_rd:3d}  {kname[:90]}")
                for v in viol[:6]:
                    print("      ", v)
                bad += bool(viol)
            print(f"---- {name}: {n_k} kernels, {n_asm} with asm / LDS DMA")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
