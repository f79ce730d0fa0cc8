#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel table:
    python tools/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/r01_bench_kernel_stats.md
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    name = re.sub(r"gemm_ft_kernel<(__bf16|float), (\d+), (\d+), (\d+), (\d+), (\d+), (Epi\w+)<[^>]*>\s*>",
                  lambda m: f"gemm_ft_kernel<{'bf16' if m.group(1) == '__bf16' else 'fp32'},{int(m.group(2)) * int(m.group(4)) * 32}x"
                            f"{int(m.group(3)) * int(m.group(5)) * 32},{m.group(7)}>", name)
    return name[:110]


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = list(cur.execute(f"select {name_col}, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by {name_col}"))
    tot = sum(r[2] for r in rows)
    rows.sort(key=lambda r: -r[2])
    print(f"# rocprofv3 --kernel-trace --stats summary ({path})\n")
    print(f"total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for n, c, s, mn, mx in rows:
        print(f"| `{short(n)}` | {c} | {s / 1e6:.3f} | {s / c / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100.0 * s / tot:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1])
