"""Default hyper-parameters; field-for-field the values of the reference's
configs/default_amass_configs.py:5-67 (values are data: batch 1280, Adam 2e-4, warm-up 5000, clip 1,
1000 scales, beta 0.1..20, sigma 0.01..50, snr 0.16)."""
import torch

from .config_dict import ConfigDict

_DEFAULTS = dict(
    OUTPUT_DIR="output",
    DATASET=dict(TRAIN_DATASET="amass", TEST_DATASET="amass", HYBRID_JOINTS_TYPE=""),
    data=dict(normalize=True, rot_rep="axis", min_max=False),
    training=dict(batch_size=1280, n_iters=400001, log_freq=50, eval_freq=50000, save_freq=50000,
                  auxiliary_loss=False, denoise_steps=10, render=False, likelihood_weighting=False,
                  continuous=True, reduce_mean=True),
    sampling=dict(n_steps_each=1, noise_removal=True, probability_flow=False, snr=0.16),
    eval=dict(batch_size=50, num_samples=500),
    model=dict(sigma_min=0.01, sigma_max=50, num_scales=1000, beta_min=0.1, beta_max=20.0),
    optim=dict(weight_decay=0, optimizer="Adam", lr=2e-4, beta1=0.9, eps=1e-8, warmup=5000, grad_clip=1.0),
    seed=42,
)


def _to_cfg(d):
    return ConfigDict({k: (_to_cfg(v) if isinstance(v, dict) else v) for k, v in d.items()})


def get_default_configs():
    cfg = _to_cfg(_DEFAULTS)
    cfg.device = torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")
    return cfg
