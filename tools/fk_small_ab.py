"""Small-batch (one wave per pose, one lane per joint) vs large-batch (one lane per pose) FK kernels inside LBS forward + backward.
Run under rocprofv3 on the GPU box for kernel times:  rocprofv3 --kernel-trace --stats ... -- python3 tools/fk_small_ab.py <poses>"""
import os, sys
import torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from dposer_amd.body_model.body_model import BodyModel
from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
DEV = "cuda:0"
B = int(sys.argv[1])
bm = BodyModel(make_synthetic_smplx_asset(seed=0)).to(DEV)
pose = torch.randn(B, 63, device=DEV) * 0.3
for env in ("1000000000", "0"):
    os.environ["DPOSER_FK_SMALL_MAX"] = env
    from dposer_amd import _C
    _C.lib().dposer_body_tuning_reload()      # the switch is read once per process otherwise
    p = pose.clone().requires_grad_(True)
    for _ in range(6):
        p.grad = None
        o = bm(pose_body=p)
        (o.v.sum() + o.Jtr.sum()).backward()
torch.cuda.synchronize()
