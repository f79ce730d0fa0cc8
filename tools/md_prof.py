"""Driver for a kernel trace of the batched motion-denoising loop: 128 sequences of 60 frames (7680 poses), 20 optimisation steps.
    rocprofv3 --kernel-trace --stats -d out -o md -- python3 tools/md_prof.py ; python3 tools/rocpd_summary.py out/.../md_results.db"""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from dposer_amd.algorithms.advanced.model import ScoreModelFC
from dposer_amd.body_model.body_model import BodyModel
from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
from dposer_amd.configs import load_config
from dposer_amd.dataset.AMASS import Posenormalizer
from dposer_amd.tasks.motion_denoising import MotionDenoise

dev = torch.device("cuda", 0)
cfg = load_config("configs.subvp.amass_scorefc_continuous.get_config")
torch.manual_seed(0)
model = ScoreModelFC(cfg, n_poses=21, pose_dim=3, hidden_dim=1024, embed_dim=512, n_blocks=2).to(dev).eval()
toy = torch.tensor(np.load(os.path.join(ROOT, "tests", "golden", "toy_poses.npy"))) if os.path.exists(os.path.join(ROOT, "tests", "golden", "toy_poses.npy")) else torch.randn(500, 63) * 0.3
stats = {"mean_poses": toy.mean(0), "std_poses": toy.std(0) + 1e-3, "min_poses": toy.min(0).values, "max_poses": toy.max(0).values}
norm = Posenormalizer(stats, device=dev, normalize=True, min_max=False, rot_rep="axis")
bm = BodyModel(make_synthetic_smplx_asset(seed=0), batch_size=60).to(dev)
args = types.SimpleNamespace(device=dev, dataset_folder="", version="", task="denoise")
md = MotionDenoise(cfg, args, model, bm, sde_N=1000, batch_size=60, normalizer=norm)
gt = toy[:60].to(dev).float()
with torch.no_grad():
    joints = bm(pose_body=gt, betas=md.betas).Jtr[:, :22] + 0.04 * torch.randn(60, 22, 3, device=dev)
S = 128
jb = joints[None].expand(S, -1, -1, -1).contiguous() + 0.01 * torch.randn(S, 60, 22, 3, device=dev)
gb = gt[None].expand(S, -1, -1).contiguous()
md.optimize_sequences(jb, gb, time_strategy="1", iterations=2, steps_per_iter=10)
torch.cuda.synchronize()
