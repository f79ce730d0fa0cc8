"""A/B of the LBS backward on one box:  python tools/lbs_bwd_ab.py
DPOSER_SKIN_BWD_FUSED = 0 (k_skin_bwd + k_skin_bwd_joints, v_posed through HBM) / 1 (one streaming pass per pose); forward + backward at
4096 and 16384 poses, interleaved child processes; the first child compares the two gradients."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from dposer_amd import _C
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    bm = BodyModel(make_synthetic_smplx_asset(seed=0)).to("cuda:0")
    if os.environ.get("CHECK") == "1":
        g = torch.Generator(device="cuda:0").manual_seed(3)
        outs = []
        for flag in ("0", "1"):
            os.environ["DPOSER_SKIN_BWD_FUSED"] = flag
            _C.lib().dposer_body_tuning_reload()
            pose = (torch.randn(2000, 63, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(3)) * 0.3).requires_grad_(True)
            wv = torch.randn(2000, 10475, 3, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(4))
            o = bm(pose_body=pose)
            ((o.v * wv).sum() + o.Jtr.sum()).backward()
            outs.append(pose.grad.clone())
        d = (outs[0] - outs[1]).abs().max().item()
        print(f"pose gradients, two-kernel vs fused: max abs diff {d:.3e} (scale {outs[0].abs().max().item():.3e})")
        sys.exit(0)
    for n in (4096, 16384):
        pose = (torch.randn(n, 63, device="cuda:0") * 0.3).contiguous().requires_grad_(True)
        gv, gj = torch.ones(n, 10475, 3, device="cuda:0"), torch.ones(n, 127, 3, device="cuda:0")
        def run():       # given incoming gradients: LBS forward + backward only (no torch reduction / stride-0 gradient copy in the timed region)
            o = bm(pose_body=pose)
            torch.autograd.backward([o.v, o.Jtr], [gv, gj])
            pose.grad = None
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                run()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5)
        print(f"fused={os.environ.get('DPOSER_SKIN_BWD_FUSED', '1')} n={n:6d} fwd+bwd {min(ts):7.3f} ms  {n / min(ts) / 1e3:6.2f} M poses/s", flush=True)
else:
    subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, CHECK="1"), check=True)
    for rnd in range(2):
        for flag in ("0", "1"):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, DPOSER_SKIN_BWD_FUSED=flag), check=True)
