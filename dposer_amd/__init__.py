"""dposer_amd -- MI355X (gfx950) native implementation of DPoser's diffusion hot path.

Host side: Python mirrors of the reference's call surface
(``ScoreModelFC``, ``sde_lib``, ``get_score_fn``, ``get_sampling_fn``, ``get_step_fn``,
``ExponentialMovingAverage``, ``BodyModel``, ``DPoser``, ``configs``); arithmetic: hand-written HIP
kernels behind the C ABI of ``include/dposer_hip.h`` (``dposer_amd/csrc``, ``libdposer_hip.so``).
There is no CPU fallback for the hot path: tensors must live on an AMD GPU.

``install_reference_aliases()`` registers the package's modules under the reference's import paths
(``lib.algorithms.advanced.model`` ...), so the reference's ``run/*.py`` scripts import them unchanged.
"""
import importlib
import sys

__version__ = "0.1.0"

_ALIASES = {
    "lib": "dposer_amd",
    "lib.algorithms": "dposer_amd.algorithms",
    "lib.algorithms.ema": "dposer_amd.algorithms.ema",
    "lib.algorithms.advanced": "dposer_amd.algorithms.advanced",
    "lib.algorithms.advanced.model": "dposer_amd.algorithms.advanced.model",
    "lib.algorithms.advanced.sde_lib": "dposer_amd.algorithms.advanced.sde_lib",
    "lib.algorithms.advanced.utils": "dposer_amd.algorithms.advanced.utils",
    "lib.algorithms.advanced.sampling": "dposer_amd.algorithms.advanced.sampling",
    "lib.algorithms.advanced.losses": "dposer_amd.algorithms.advanced.losses",
    "lib.algorithms.advanced.likelihood": "dposer_amd.algorithms.advanced.likelihood",
    "lib.body_model": "dposer_amd.body_model",
    "lib.body_model.body_model": "dposer_amd.body_model.body_model",
    "lib.body_model.smpl": "dposer_amd.body_model.smpl",
    "lib.body_model.utils": "dposer_amd.body_model.utils",
    "lib.body_model.constants": "dposer_amd.body_model.constants",
    "lib.utils": "dposer_amd.utils",
    "lib.utils.transforms": "dposer_amd.utils.transforms",
    "lib.utils.misc": "dposer_amd.utils.misc",
    "lib.utils.generic": "dposer_amd.utils.generic",
    "lib.utils.metric": "dposer_amd.utils.metric",
    "lib.dataset": "dposer_amd.dataset",
    "lib.dataset.AMASS": "dposer_amd.dataset.AMASS",
    "lib.dataset.EvaSampler": "dposer_amd.dataset.EvaSampler",
    "configs": "dposer_amd.configs",
    "configs.default_amass_configs": "dposer_amd.configs.default_amass_configs",
    "configs.subvp": "dposer_amd.configs.subvp",
    "configs.subvp.amass_scorefc_continuous": "dposer_amd.configs.subvp.amass_scorefc_continuous",
}


def install_reference_aliases():
    """Make ``import lib.algorithms.advanced.model`` (etc.) resolve to this package."""
    for ref_name, ours in _ALIASES.items():
        try:
            sys.modules.setdefault(ref_name, importlib.import_module(ours))
        except ModuleNotFoundError:
            pass
