import importlib
import importlib.util
import os


class ConfigDict(dict):
    """Attribute-style nested dict (``cfg.a.b = 1``); stands in for ml_collections.ConfigDict."""

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as e:
            raise AttributeError(key) from e

    def __setattr__(self, key, value):
        self[key] = value

    def __delattr__(self, key):
        del self[key]

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, ConfigDict) else v) for k, v in self.items()}


def load_config(spec):
    """Accepts what the reference's entry points accept:
    * a path to a config file, ``--config configs/subvp/amass_scorefc_continuous.py`` (run/train.py:39-41)
    * a dotted path to the factory, ``configs.subvp.amass_scorefc_continuous.get_config``
      (lib/utils/generic.py:51-56)."""
    if spec.endswith(".py") or os.sep in spec:
        path = spec
        if not os.path.isabs(path) and not os.path.exists(path):
            path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), spec)
        s = importlib.util.spec_from_file_location("_dposer_cfg_" + os.path.basename(path)[:-3], path)
        mod = importlib.util.module_from_spec(s)
        s.loader.exec_module(mod)
        return mod.get_config()
    module_name, fn = spec.rsplit(".", 1)
    try:
        mod = importlib.import_module(module_name)
    except ModuleNotFoundError:
        mod = importlib.import_module("dposer_amd." + module_name)
    return getattr(mod, fn)()
