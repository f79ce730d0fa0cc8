import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dposer_amd.body_model.body_model import BodyModel
from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
bm = BodyModel(make_synthetic_smplx_asset(seed=0)).to("cuda")
pb = (torch.randn(4096, 63, device="cuda") * 0.3).requires_grad_(True)
fwd_only = len(sys.argv) > 1 and sys.argv[1] == "fwd"      # (forward only: the byte count of the forward leg by itself)
for _ in range(4):
    if fwd_only:
        with torch.no_grad():
            bm(pose_body=pb)
        continue
    out = bm(pose_body=pb)
    (out.v.sum() + out.Jtr.sum()).backward()
    pb.grad = None
torch.cuda.synchronize()
