"""sub-VP SDE + ScoreModelFC (H 1024, E 512, 2 blocks) -- the shipped configuration of the
reference (configs/subvp/amass_scorefc_continuous.py:21-47)."""
from dposer_amd.configs.default_amass_configs import get_default_configs

_OVERRIDES = {
    "training": dict(sde="subvpsde", continuous=True),
    "sampling": dict(method="pc", predictor="euler_maruyama", corrector="none"),
    "model": dict(type="ScoreModelFC", HIDDEN_DIM=1024, EMBED_DIM=512, N_BLOCKS=2, dropout=0.1, fourier_scale=16,
                  scale_by_sigma=True, ema_rate=0.9999, nonlinearity="swish", embedding_type="positional"),
}


def get_config():
    cfg = get_default_configs()
    for section, values in _OVERRIDES.items():
        for k, v in values.items():
            cfg[section][k] = v
    return cfg
