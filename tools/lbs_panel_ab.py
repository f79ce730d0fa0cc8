"""A/B of the block -> tile order of the LBS blend-gradient GEMMs on one box:  python tools/lbs_panel_ab.py
DPOSER_LBS_BWD_PANEL_ORDER = 0 (generic order) / 1 (the tiles that stream the same d_offsets panel side by side on one XCD); LBS forward +
backward with given gradients at 4096 / 7680 / 16384 poses, interleaved child processes; the first child checks that the pose gradients
are bit-identical (each (tile, split) writes its own slab: the order cannot change the sums)."""
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
        outs = []
        for flag in ("0", "1"):
            os.environ["DPOSER_LBS_BWD_PANEL_ORDER"] = flag
            _C.lib().dposer_body_tuning_reload()
            pose = (torch.randn(7680, 63, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(3)) * 0.3).requires_grad_(True)
            wv = torch.randn(7680, 10475, 3, device="cuda:0", generator=torch.Generator(device="cuda:0").manual_seed(4))
            o = bm(pose_body=pose)
            torch.autograd.backward([o.v, o.Jtr], [wv, torch.ones_like(o.Jtr)])
            outs.append(pose.grad.clone())
        print(f"pose gradients at 7680 poses, generic vs panel order: bit-identical = {torch.equal(outs[0], outs[1])}")
        sys.exit(0)
    for n in (4096, 7680, 16384):
        pose = (torch.randn(n, 63, device="cuda:0") * 0.3).contiguous().requires_grad_(True)
        gv, gj = torch.ones(n, 10475, 3, device="cuda:0"), torch.ones(n, 127, 3, device="cuda:0")
        def run():
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
        print(f"panel_order={os.environ.get('DPOSER_LBS_BWD_PANEL_ORDER', '1')} n={n:6d} fwd+bwd {min(ts):7.3f} ms  {n / min(ts) / 1e3:6.2f} M poses/s", flush=True)
else:
    subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, CHECK="1"), check=True)
    for rnd in range(2):
        for flag in ("0", "1"):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, DPOSER_LBS_BWD_PANEL_ORDER=flag), check=True)
