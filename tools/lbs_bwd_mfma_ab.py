"""A/B of the skinning-backward kernels inside the full LBS forward + backward at 4096 / 16384 poses: DPOSER_SKIN_BWD_MFMA = 0 (k_skin_bwd_fused:
joint lists walked through LDS) / 2 / 4 / 8 (k_skin_bwd_mfma with that many poses per workgroup), interleaved child processes; every child
prints the relative difference of its pose gradient to a float64 torch restatement-free reference: the mode-0 gradient of the same inputs
(saved by the first child).   python tools/lbs_bwd_mfma_ab.py [--modes 0,2,4]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    bm = BodyModel(make_synthetic_smplx_asset(seed=0)).to("cuda:0")
    mode = os.environ.get("DPOSER_SKIN_BWD_MFMA", "1")
    for n in (4096, 16384):
        gen = torch.Generator(device="cuda:0").manual_seed(n)
        pose = (torch.randn(n, 63, device="cuda:0", generator=gen) * 0.3).requires_grad_(True)
        gv = torch.randn(n, 10475, 3, device="cuda:0", generator=gen) * 0.01
        gj = torch.randn(n, 127, 3, device="cuda:0", generator=gen)

        def step():
            out = bm(pose_body=pose)
            torch.autograd.backward([out.v, out.Jtr], [gv, gj])
            g = pose.grad
            pose.grad = None
            return g

        for _ in range(3):
            g = step()
        torch.cuda.synchronize()
        ref_path = f"/tmp/lbs_bwd_ref_{n}.npy"
        if mode == "0" and not os.path.exists(ref_path):
            np.save(ref_path, g.cpu().numpy())
        err = float("nan")
        if os.path.exists(ref_path):
            r = torch.tensor(np.load(ref_path), device="cuda:0")
            err = float((g - r).norm() / r.norm())
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                step()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        ts.sort()
        print(f"mfma={mode:>2} n={n:6d}  fwd+bwd median {ts[2]:7.3f} ms  best {ts[0]:7.3f} ms   d pose vs mode 0: {err:.2e}  finite {bool(torch.isfinite(g).all())}", flush=True)
else:
    modes = "0,2,4"
    if "--modes" in sys.argv:
        modes = sys.argv[sys.argv.index("--modes") + 1]
    for f in ("/tmp/lbs_bwd_ref_4096.npy", "/tmp/lbs_bwd_ref_16384.npy"):
        if os.path.exists(f):
            os.remove(f)
    for rnd in range(2):
        for m in modes.split(","):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, DPOSER_SKIN_BWD_MFMA=m), check=False)
