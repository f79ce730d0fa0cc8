"""A/B of the joints-only FK kernel: python tools/fk_ab.py  (runs itself twice, DPOSER_FK_DMA=0 / 1, interleaved rounds)."""
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
    g = torch.Generator(device="cuda:0").manual_seed(1)
    chk = (torch.randn(100_003, 63, device="cuda:0", generator=g) * 0.4).contiguous()
    j = bm.fk_joints(chk, trans=chk[:, :3].contiguous())
    print(f"dma={os.environ.get('DPOSER_FK_DMA', '0')} checksum {float(j.double().sum()):.9f} {float(j.double().abs().sum()):.6f} {float(j[77777, 13, 1]):.7f}")
    for n in (1 << 20, 1 << 18, 1 << 22):
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
        us = min(ts)
        print(f"dma={os.environ.get('DPOSER_FK_DMA', '0')} n={n:8d}  {us:8.1f} us  {n / us / 1e3:6.2f} G poses/s  {516 * n / us / 1e6:6.2f} TB/s")
else:
    for rnd in range(2):
        for flag in ("0", "1"):
            subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, DPOSER_FK_DMA=flag), check=True)
