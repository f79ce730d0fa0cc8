"""A/B of the LBS forward (pose-blend GEMM tile order): python tools/lbs_ab.py  (DPOSER_LBS_CGROUP = 0 / 8, interleaved processes)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    bm = BodyModel(make_synthetic_smplx_asset(seed=0)).to("cuda:0")
    for n in (4096, 16384):
        pose = (torch.randn(n, 63, device="cuda:0") * 0.3).contiguous()
        with torch.no_grad():
            for _ in range(3):
                bm(pose_body=pose)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    bm(pose_body=pose)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 5)
        print(f"cgroup={os.environ.get('DPOSER_LBS_CGROUP', '8')} n={n:6d}  {min(ts):7.3f} ms  {n / min(ts) / 1e3:6.2f} M poses/s")
else:
    for rnd in range(2):
        for flag in ("0", "8"):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, DPOSER_LBS_CGROUP=flag), check=True)
