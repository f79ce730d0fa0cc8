"""A/B of the skinning kernels and the LBS autograd wrapper on one box:  python tools/lbs_skin_ab.py
DPOSER_SKIN_WAVE = 2 (four vertices per thread in flight, one pose per block) / 3 (the same over runs of poses: k_skin_run; round 3
compared 0 / 2); forward and forward + backward at 4096
and 16384 poses, interleaved child processes; the first child also checks that both kernels return the same bits."""
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
        pose = (torch.randn(777, 63, device="cuda:0") * 0.3).contiguous()
        tr = torch.randn(777, 3, device="cuda:0")
        outs = []
        for flag in ("2", "3"):
            os.environ["DPOSER_SKIN_WAVE"] = flag
            _C.lib().dposer_body_tuning_reload()
            with torch.no_grad():
                outs.append(bm(pose_body=pose, trans=tr).v.clone())
        print("skin kernels bit-identical:", bool(torch.equal(outs[0], outs[1])), float((outs[0] - outs[1]).abs().max()))
        sys.exit(0)
    for n in (4096, 16384):
        pose = (torch.randn(n, 63, device="cuda:0") * 0.3).contiguous().requires_grad_(True)
        grads = [torch.ones(n, 10475, 3, device="cuda:0"), torch.ones(n, 127, 3, device="cuda:0")]   # given gradients: LBS only in the timed region
        for grad in (False, True):
            def run():
                if grad:
                    o = bm(pose_body=pose)
                    torch.autograd.backward([o.v, o.Jtr], grads)
                    pose.grad = None
                else:
                    with torch.no_grad():
                        bm(pose_body=pose)
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
            print(f"skin_wave={os.environ.get('DPOSER_SKIN_WAVE', '1')} n={n:6d} {'fwd+bwd' if grad else 'fwd    '} {min(ts):7.3f} ms  {n / min(ts) / 1e3:6.2f} M poses/s", flush=True)
else:
    subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, CHECK="1"), check=True)
    for rnd in range(2):
        for flag in ("2", "3"):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, DPOSER_SKIN_WAVE=flag), check=True)
