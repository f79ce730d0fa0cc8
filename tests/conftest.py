import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def tuning_env(monkeypatch):
    """The library reads its A/B switches once per process.  ``tuning_env(NAME="1", ...)`` sets them for the running test and makes
    the library re-read them; the switches are restored (and re-read) when the test ends."""
    from dposer_amd import _C

    def reload():
        _C.lib().dposer_scorefc_tuning_reload()
        _C.lib().dposer_body_tuning_reload()

    def set_env(**kv):
        for k, v in kv.items():
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, str(v))
        reload()

    yield set_env
    monkeypatch.undo()
    reload()
