"""Driver of tools/fk_pmc.sh: the joints-only body query (k_fk_joints_dma) at 2^20 poses, six calls."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dposer_amd.body_model.body_model import BodyModel
from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
bm = BodyModel(make_synthetic_smplx_asset(seed=0)).to("cuda")
pose = (torch.randn(1 << 20, 63, device="cuda") * 0.3).contiguous()
for _ in range(6):
    bm.fk_joints(pose)
torch.cuda.synchronize()
