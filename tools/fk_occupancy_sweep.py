#!/usr/bin/env python3
"""Occupancy probe of the joints-only FK kernel (k_fk_joints_dma, 2^20 poses): the same kernel with extra, unused dynamic LDS per one-wave
workgroup (DPOSER_FK_LDS_PAD) so that fewer waves are resident per CU.  If throughput still rises towards the shipped 9 workgroups per CU
(16.9 KB each of 160 KB) the kernel is latency-bound on resident waves and a leaner LDS image would pay; if it has flattened, it would not.
    python tools/fk_occupancy_sweep.py        # one child process per setting, interleaved rounds; prints a markdown table"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PADS = [0, 3072, 6144, 10240, 15360, 23552, 36864]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    bm = BodyModel(make_synthetic_smplx_asset(seed=0)).to("cuda:0")
    n = 1 << 20
    pose = (torch.randn(n, 63, device="cuda:0") * 0.3).contiguous()
    for _ in range(3):
        bm.fk_joints(pose)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            bm.fk_joints(pose)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    ts.sort()
    print(f"US {ts[2]:.2f}")
else:
    res = {p: [] for p in PADS}
    for rnd in range(3):
        for p in PADS:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, DPOSER_FK_LDS_PAD=str(p)), capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("US ")]
            if line:
                res[p].append(float(line[0].split()[1]))
    print("| LDS per workgroup (KB) | workgroups (= waves) per CU by LDS | us per launch (median of 3 processes) | G poses/s | fraction of 8 TB/s (516 B / pose) |")
    print("|---:|---:|---:|---:|---:|")
    for p in PADS:
        if not res[p]:
            continue
        us = sorted(res[p])[len(res[p]) // 2]
        lds = 64 * 66 * 4 + p
        print(f"| {lds / 1024:.1f} | {160 * 1024 // lds} | {us:.1f} | {(1 << 20) / us * 1e-3:.2f} | {516.0 * (1 << 20) / (us * 1e-6) / 8e12:.3f} |")
