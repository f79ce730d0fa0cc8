#!/usr/bin/env python3
"""LBS forward + backward at N poses (default 4096 and 7680), given incoming gradients: median of 5 runs of 10 (HIP events).  One line per N.
A/B by environment (DPOSER_LBS_BWD_ROWCAT, DPOSER_SKIN_BWD_MFMA, ...), one process per setting:  python tools/lbs_fwd_bwd_time.py [N ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dposer_amd.body_model.body_model import BodyModel
from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset

dev = "cuda:0"
bm = BodyModel(make_synthetic_smplx_asset(seed=0)).to(dev)
for n in ([int(a) for a in sys.argv[1:]] or [4096, 7680]):
    pb = (torch.randn(n, 63, device=dev) * 0.3).requires_grad_(True)
    gv, gj = torch.ones(n, 10475, 3, device=dev), torch.ones(n, 127, 3, device=dev)

    def run():
        out = bm(pose_body=pb)
        torch.autograd.backward([out.v, out.Jtr], [gv, gj])
        pb.grad = None
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    ts.sort()
    print(f"LBS fwd+bwd n={n}: median {ts[2]:.4f} ms  min {ts[0]:.4f} ms")
    del gv, gj
