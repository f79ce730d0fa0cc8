#!/usr/bin/env python3
"""Audit of the hand-placed K loops in the ISA hipcc emits for the library's GEMM kernels (run after touching gemm.h / gemm_kloop_asm.h):
    python tools/check_kloop_isa.py            (compiles dposer_amd/csrc/gemm_launch.hip to ISA, ~2 min)
The compiler cannot see the MFMAs inside an asm statement, so two things must hold in what it wraps around them:
  * between two neighbouring stage statements nothing reads or writes an ACCUMULATOR register -- a register copy there would read an XDL
    result before its wait states have passed;
  * the kernels do not spill (scratch) -- a spill inside the K loop would do the same.
Prints one line per kernel that contains stage statements; exits non-zero on a violation."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "dposer_amd", "csrc", "gemm_launch.hip")
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "k.s")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/dposer_amd/csrc", f"-I{ROOT}/include", "--cuda-device-only",
                    "-S", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
bad = 0
for m in re.finditer(r"^(_Z\S+):\s*;\s*@\1\n(.*?)s_endpgm", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    blocks = [b for b in re.finditer(r";;#ASMSTART\n(.*?);;#ASMEND", body, re.S) if b.group(1).count("v_mfma") >= 8]
    if not blocks:
        continue
    # accumulator registers = the MFMA destinations of a stage statement
    acc = set()
    for d in re.finditer(r"v_mfma\S+ v\[(\d+):(\d+)\]", blocks[0].group(1)):
        acc.update(range(int(d.group(1)), int(d.group(2)) + 1))
    def regs(line):
        out = set()
        for r in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", line):
            out.update(range(int(r.group(1)), int(r.group(2)) + 1) if r.group(1) else [int(r.group(3))])
        return out
    viol = []
    # straight-line neighbours only: the group of four inside the loop (statements 3..6 of 10) and the three tail stages (7..9); the
    # text between other pairs holds other paths' code (accumulator zero-fill of the short-K fallback)
    pairs = [(3, 4), (4, 5), (5, 6), (7, 8), (8, 9)] if len(blocks) == 10 else []
    for i, j in pairs:
        between = body[blocks[i].end():blocks[j].start()]
        for line in between.splitlines():
            ins = line.strip().split()
            if ins and not ins[0].startswith(";") and (regs(line) & acc or ins[0].startswith("scratch_")):
                viol.append(line.strip())
    scratch = "scratch_" in body
    print(f"{'FAIL' if viol or scratch else 'ok  '} {len(blocks):3d} stage statements  {name[:110]}")
    for v in viol[:5]:
        print("      between stages:", v)
    bad += bool(viol) or scratch
sys.exit(1 if bad else 0)
