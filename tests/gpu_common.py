"""Helpers for the GPU parity tests (test infrastructure; may import oracle/)."""
import numpy as np
import torch

from weights import make_weights
from oracle import score_ref as R

DEV = "cuda:0"


def make_model(seed, D=63, precision="fp32", dropout=0.1, embedding="positional", device=DEV, nonlinearity="swish"):
    from dposer_amd.algorithms.advanced.model import ScoreModelFC
    from dposer_amd.configs import load_config
    cfg = load_config("configs.subvp.amass_scorefc_continuous.get_config")
    cfg.model.dropout = dropout
    cfg.model.embedding_type = embedding
    cfg.model.nonlinearity = nonlinearity
    m = ScoreModelFC(cfg, n_poses=21, pose_dim=D // 21, hidden_dim=1024, embed_dim=512, n_blocks=2)
    w = make_weights(seed, D=D, fourier=(embedding == "fourier"))
    sd = m.state_dict()
    for k, v in w.items():
        sd[k] = v
    m.load_state_dict(sd)
    m.precision = precision
    m.to(device)
    m.eval()
    p = dict(w)
    p["sigmas"] = R.sigma_table()
    return cfg, m, p


def t2n(t):
    return t.detach().float().cpu().numpy()


# parity tolerances (relative L2 unless noted)
TOL_FP32 = 2e-5      # fp32 MFMA path vs fp32 CPU oracle/reference: summation order + libm ulps
TOL_BF16 = 1e-2      # bf16 MFMA path: 8-bit mantissa on weights and on every inter-layer activation (measured 4.4e-3 / 5.0e-3)
