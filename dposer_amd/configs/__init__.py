"""Configuration entry points (counterpart of the reference's ``configs/`` package).

The reference builds ``ml_collections.ConfigDict`` objects (configs/default_amass_configs.py:5-67);
``ml_collections`` / ``absl`` are not available on the target image, so ``ConfigDict`` below is a
minimal attribute-dict with the same access pattern (``config.model.HIDDEN_DIM``).
"""
from .config_dict import ConfigDict, load_config  # noqa: F401
