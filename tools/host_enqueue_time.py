"""Host enqueue time vs wall time of the fused training step at several batch sizes (run on the GPU box from the repo root):
    python tools/host_enqueue_time.py
enqueue ~= wall  =>  the loop is bound by the host launching kernels, not by the GPU (docs/experiments_rounds_1-4.md, section 8 item 2)."""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
import bench
from dposer_amd.algorithms.advanced import losses, sde_lib
from dposer_amd.algorithms.advanced.model import ScoreModelFC
from dposer_amd.algorithms.ema import ExponentialMovingAverage
from dposer_amd.configs import load_config
dev = torch.device("cuda", 0)
cfg = load_config("configs.subvp.amass_scorefc_continuous.get_config")
torch.manual_seed(42)
model = ScoreModelFC(cfg, n_poses=21, pose_dim=3, hidden_dim=1024, embed_dim=512, n_blocks=2)
model.precision = "bf16"; model.to(dev); model._rng_seed = 42
sde = sde_lib.subVPSDE(beta_min=cfg.model.beta_min, beta_max=cfg.model.beta_max, N=cfg.model.num_scales)
state = dict(optimizer=losses.get_optimizer(cfg, model.parameters()), model=model,
             ema=ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate), step=0)
step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=True, continuous=True)
for B in (1280, 8192, 16384, 32768):
    batch, _ = bench.synthetic_poses(B, "cpu"); batch = batch.to(dev).contiguous()
    for _ in range(5): step_fn(state, batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): step_fn(state, batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"B={B:6d}  host enqueue {1e3*(t1-t0)/50:.3f} ms/step   wall {1e3*(t2-t0)/50:.3f} ms/step")
