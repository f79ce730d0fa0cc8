#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel table:
    python tools/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/r01_bench_kernel_stats.md
"""
import re
import sqlite3
import sys


_DEMANGLE = {}


def demangle(name):
    if not name.startswith("_Z"):
        return name
    if name not in _DEMANGLE:
        import os
        import shutil
        import subprocess
        tool = "/opt/rocm/lib/llvm/bin/llvm-cxxfilt" if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-cxxfilt") else shutil.which("c++filt")
        # GNU c++filt (binutils 2.38) does not know the __bf16 mangling (DF16b): hand it over as a class named __bf16
        try:
            out = subprocess.run([tool, name.replace("DF16b", "6__bf16")], capture_output=True, text=True, timeout=10).stdout.strip()
            _DEMANGLE[name] = out if out and not out.startswith("_Z") else name
        except Exception:
            _DEMANGLE[name] = name
    return _DEMANGLE[name]


def short(name):
    name = demangle(name).replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    name = name.replace("gemm_ft_kernel<bool _Accum, int, E,", "gemm_ft_kernel<__bf16, 1,")      # (a demangler that does not know DF16b)

    def gemm(m):
        t = "bf16" if m.group(1) == "__bf16" else "fp32"
        if t == "bf16" and m.group(8).split(",")[0].strip() == "float":
            t = "bf16x3"                           # bf16 operand planes under a fp32-storage epilogue (gemm_launch_x3.hip)
        return (f"gemm_ft_kernel<{t},{int(m.group(2)) * int(m.group(4)) * 32}x{int(m.group(3)) * int(m.group(5)) * 32},"
                f"{m.group(7)}{'<train>' if 'true' in m.group(8) else ''}>")
    name = re.sub(r"gemm_ft_kernel<(__bf16|float), (\d+), (\d+), (\d+), (\d+), (\d+), (Epi\w+)<([^>]*)>(, \d+)?\s*>", gemm, name)
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


def gaps(path):
    """Idle time between consecutive kernels (start[i+1] - end[i]) grouped by the kernel that follows the gap."""
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = list(cur.execute(f"select {name_col}, start, end from kernels order by start"))
    agg = {}
    for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
        g = s1 - e0
        if g > 200000:      # host-side pauses (phase changes, synchronisations) are not launch gaps
            continue
        a = agg.setdefault(short(n1), [0, 0.0, 0.0])
        a[0] += 1
        a[1] += g
        a[2] += e1 - s1
    print(f"# gaps between consecutive kernels ({path}); gaps > 200 us (host pauses) skipped\n")
    print("| kernel (the one AFTER the gap) | n | mean gap before it us | mean duration us | gap / duration |")
    print("|---|---:|---:|---:|---:|")
    for k, (n, g, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"| `{k}` | {n} | {g / n / 1e3:.2f} | {d / n / 1e3:.1f} | {g / max(d, 1):.3f} |")


def sequence(path, anchor="k_prep_train", which=3, title="one training step"):
    """The kernels of ONE training step in launch order (from the `which`-th occurrence of `anchor` to the next one)."""
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = list(cur.execute(f"select {name_col}, start, end from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if anchor in short(r[0])]
    lo, hi = idx[which], idx[which + 1]
    t0 = rows[lo][1]
    print(f"# {title}, kernels in start order ({path}); span = {(rows[hi][1] - t0) / 1e3:.1f} us\n")
    print("| # | kernel | start us | duration us | gap to previous end us |")
    print("|---:|---|---:|---:|---:|")
    prev_end = None
    for k, (n, s0, e0) in enumerate(rows[lo:hi]):
        gap = "" if prev_end is None else f"{(s0 - prev_end) / 1e3:.1f}"
        print(f"| {k} | `{short(n)[:70]}` | {(s0 - t0) / 1e3:.1f} | {(e0 - s0) / 1e3:.1f} | {gap} |")
        prev_end = e0 if prev_end is None else max(prev_end, e0)


def _pmc_agg(paths):
    agg = {}
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        for name, counter, n, avg in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                                 "group by kernel_name, counter_name"):
            agg.setdefault(short(name), {})[counter] = (n, avg)
    return agg


def pmc_json(paths):
    """{kernel: {read_MB, write_MB, launches}} with the gfx950 x2 FETCH_SIZE correction (see pmc())."""
    import json
    out = {}
    for k, v in _pmc_agg(paths).items():
        f = v.get("FETCH_SIZE", (0, 0.0))
        w = v.get("WRITE_SIZE", (0, 0.0))
        out[k] = {"read_MB": 2 * f[1] * 1024 / 1e6, "write_MB": w[1] * 1024 / 1e6, "launches": max(f[0], w[0])}
    print(json.dumps(out, indent=1, sort_keys=True))


def pmc(paths):
    """Per-kernel HBM traffic from separate FETCH_SIZE / WRITE_SIZE passes (values are KiB per dispatch)."""
    agg = _pmc_agg(paths)
    print("# HBM traffic per launch from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs)\n")
    print("Correction per MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports 1/2 of the bytes of wide coalesced "
          "streaming reads -> `read MB` = 2 x FETCH_SIZE; WRITE_SIZE taken as is.  Counters are KiB.\n")
    print("| kernel | launches | FETCH_SIZE KiB (raw avg) | read MB (x2) | WRITE_SIZE KiB (avg) | write MB | total MB / launch |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    rows = []
    for k, v in agg.items():
        f = v.get("FETCH_SIZE", (0, 0.0))
        w = v.get("WRITE_SIZE", (0, 0.0))
        rd, wr = 2 * f[1] * 1024 / 1e6, w[1] * 1024 / 1e6
        rows.append((rd + wr, k, max(f[0], w[0]), f[1], rd, w[1], wr))
    for tot, k, n, fr, rd, wrk, wr in sorted(rows, reverse=True)[:24]:
        print(f"| `{k}` | {n} | {fr:.0f} | {rd:.1f} | {wrk:.0f} | {wr:.1f} | {tot:.1f} |")


def counters(paths):
    """Per-kernel average of every counter found in the given PMC-pass databases."""
    agg, names = {}, []
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        for name, counter, n, avg in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                                 "group by kernel_name, counter_name"):
            agg.setdefault(short(name), {})[counter] = (n, avg)
            if counter not in names:
                names.append(counter)
    print("| kernel | launches | " + " | ".join(names) + " |")
    print("|---|---:|" + "---:|" * len(names))
    for k, v in agg.items():
        n = max(x[0] for x in v.values())
        print(f"| `{k}` | {n} | " + " | ".join(f"{v[c][1]:.4g}" if c in v else "n/a" for c in names) + " |")


def mfma(paths, as_json=False):
    """Matrix-pipe evidence per kernel from the SQ / GRBM passes of tools/profile_bench.sh (each pass also carries a kernel trace):
       effective clock  = GRBM_GUI_ACTIVE / 8 / kernel duration      (MI355X_MICROARCH.md, DVFS give-back; the counter is the SUM
                          over the 8 XCDs' GRBMs: a 2.4 ms fp32-MFMA kernel reads 19.0 "GHz" = 8 x 2.38.  It also counts the
                          dispatch ramp around a kernel: below ~50 us the quotient overshoots -- read it for the long GEMMs only)
       MFMA busy        = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)   (the counter adds 32 cycles per
                          v_mfma_f32_32x32x16_bf16 on the SIMD that issued it; 256 CUs x 4 SIMDs; cross-check: 32 x SQ_INSTS_MFMA)
       issue stall      = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES,  parked = SQ_WAIT_ANY / SQ_WAVE_CYCLES  (quad-cycles both)."""
    import json
    agg, dur = {}, {}
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        for name, counter, n, avg in cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                                 "group by kernel_name, counter_name"):
            agg.setdefault(short(name), {})[counter] = (n, avg)
        cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
        for name, n, avg in cur.execute(f"select {name_col}, count(*), avg(end-start) from kernels group by {name_col}"):
            d = dur.setdefault(short(name), [0, 0.0])
            d[0] += n
            d[1] += n * avg
    rows = []
    for k, v in agg.items():
        g = v.get("GRBM_GUI_ACTIVE", (0, 0.0))[1]
        if g <= 0 or k not in dur:
            continue
        us = dur[k][1] / dur[k][0] / 1e3
        # a counter NO pass collected for this kernel is None ("n/a" in the table / null in the JSON), never a zero: a pass that only
        # asked for GRBM_GUI_ACTIVE used to print a GEMM with SQ_INSTS_MFMA 0 and MFMA busy 0.000
        get = lambda name: v[name][1] if name in v else None
        busy, wc = get("SQ_VALU_MFMA_BUSY_CYCLES"), get("SQ_WAVE_CYCLES")
        frac = lambda name: (get(name) / wc) if (wc and get(name) is not None) else None
        rows.append({"kernel": k, "launches": dur[k][0], "avg_us": us, "grbm_gui_active": g, "clock_ghz": g / 8.0 / us / 1e3,
                     "mfma_busy_cycles": busy, "mfma_busy_frac": None if busy is None else busy / (g / 8.0 * 1024.0),
                     "insts_mfma": get("SQ_INSTS_MFMA"), "insts_valu": get("SQ_INSTS_VALU"),
                     "issue_stall_frac": frac("SQ_WAIT_INST_ANY"), "parked_frac": frac("SQ_WAIT_ANY"), "active_frac": frac("SQ_ACTIVE_INST_ANY")})
    rows.sort(key=lambda r: -r["avg_us"] * r["launches"])
    if as_json:
        print(json.dumps({r["kernel"]: r for r in rows}, indent=1, sort_keys=True))
        return
    print("# Matrix-pipe counters per kernel (rocprofv3 --kernel-trace --pmc, separate SQ / GRBM passes; durations under the profiler)\n")
    print(mfma.__doc__.strip() + "\n")
    print("| kernel | launches | avg us | clock GHz | MFMA busy | SQ_INSTS_MFMA | SQ_INSTS_VALU | issue-stall | parked | active |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    f = lambda x: "n/a" if x is None else f"{x:.3f}"
    g4 = lambda x: "n/a" if x is None else f"{x:.4g}"
    for r in rows[:24]:
        print(f"| `{r['kernel']}` | {r['launches']} | {r['avg_us']:.1f} | {r['clock_ghz']:.2f} | {f(r['mfma_busy_frac'])} | {g4(r['insts_mfma'])} | "
              f"{g4(r['insts_valu'])} | {f(r['issue_stall_frac'])} | {f(r['parked_frac'])} | {f(r['active_frac'])} |")


def dispatch_table(paths, needle, per_step=0):
    """Per-DISPATCH counters and durations of the kernels whose short name contains `needle`, in launch order, from PMC-pass databases
    (each pass also carries a kernel trace).  The n-th dispatch of a kernel name in one database is matched with the n-th in the
    others.  With `per_step` = launches of that kernel per training step, the table is folded over the steps (mean per position):
    position 0 of EpiGNBwd is the K = 64 post_dense launch -- a GroupNorm-backward epilogue with (almost) no K loop -- the others K = 1024."""
    per = {}          # short name -> {"dur": [..], counter: [..]}
    for path in paths:
        cur = sqlite3.connect(path).cursor()
        ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        order = "dispatch_id" if "dispatch_id" in ccols else ("id" if "id" in ccols else "rowid")
        for name, counter, value in cur.execute(f"select kernel_name, counter_name, value from counters_collection order by {order}"):
            k = short(name)
            if needle in k:
                per.setdefault(k, {}).setdefault(counter, []).append(value)
        kcols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        name_col = "name" if "name" in kcols else [c for c in kcols if "name" in c][0]
        durs = {}
        for name, s0, e0 in cur.execute(f"select {name_col}, start, end from kernels order by start"):
            k = short(name)
            if needle in k:
                durs.setdefault(k, []).append((e0 - s0) / 1e3)
        for k, v in durs.items():
            per.setdefault(k, {}).setdefault("duration_us", v)       # (first database wins: durations differ little between passes)
    for k, cols in per.items():
        names = sorted(cols)
        n = min(len(v) for v in cols.values())
        print(f"\n## `{k}`: {n} dispatches\n")
        if per_step > 0:
            print("| position in step | launches | " + " | ".join(names) + " |")
            print("|---:|---:|" + "---:|" * len(names))
            for pos in range(per_step):
                idx = [i for i in range(n) if i % per_step == pos]
                print(f"| {pos} | {len(idx)} | " + " | ".join(f"{sum(cols[c][i] for i in idx) / max(len(idx), 1):.5g}" for c in names) + " |")
        else:
            print("| # | " + " | ".join(names) + " |")
            print("|---:|" + "---:|" * len(names))
            for i in range(n):
                print(f"| {i} | " + " | ".join(f"{cols[c][i]:.5g}" for c in names) + " |")


USAGE = """usage: rocpd_summary.py <kernel-trace.db>                      per-kernel table (calls, total, avg, min, max)
       rocpd_summary.py --gaps <db>                            idle time in front of each kernel kind
       rocpd_summary.py --sequence <db> [anchor [which]]       kernels of one step in start order
       rocpd_summary.py --pmc | --pmc-json <db> ...            FETCH_SIZE / WRITE_SIZE passes -> HBM bytes per launch
       rocpd_summary.py --counters <db> ...                    per-kernel average of every counter
       rocpd_summary.py --mfma | --mfma-json <db> ...          matrix-pipe busy fraction, effective clock, stalls
       rocpd_summary.py --dispatches <needle> <per_step> <db> ...   per-dispatch counters of one kernel kind, folded over the steps"""


def cli(argv):
    if len(argv) < 2 or argv[1] in ("-h", "--help"):
        print(USAGE)
        return 0 if len(argv) >= 2 else 2
    mode, rest = argv[1], argv[2:]
    if mode.startswith("--") and not rest:
        print(USAGE, file=sys.stderr)
        return 2
    if mode in ("--mfma", "--mfma-json"):
        mfma(rest, as_json=mode == "--mfma-json")
    elif mode == "--sequence":
        if len(rest) > 1:      # --sequence db anchor [which]: e.g. EpiEmStep 500 -> one sampler step
            sequence(rest[0], rest[1], int(rest[2]) if len(rest) > 2 else 3, f"from one `{rest[1]}` to the next")
        else:
            sequence(rest[0])
    elif mode == "--gaps":
        gaps(rest[0])
    elif mode == "--pmc":
        pmc(rest)
    elif mode == "--pmc-json":
        pmc_json(rest)
    elif mode == "--counters":
        counters(rest)
    elif mode == "--dispatches":
        dispatch_table(rest[2:], rest[0], int(rest[1]))
    elif mode.startswith("--"):
        print(USAGE, file=sys.stderr)
        return 2
    else:
        main(mode)
    return 0


if __name__ == "__main__":
    sys.exit(cli(sys.argv))
