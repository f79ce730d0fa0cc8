"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def _log_measured(kind, value):
    """DPOSER_LOG_ERR=<file>: append every measured error with the id of the running test (how the tolerances were set)."""
    path = os.environ.get("DPOSER_LOG_ERR")
    if path:
        with open(path, "a") as f:
            f.write(f"{os.environ.get('PYTEST_CURRENT_TEST', '?')}\t{kind}\t{value:.3e}\n")


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    e = float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
    _log_measured("rel_err", e)
    return e


def max_rel(a, b, floor=1e-6):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / (np.abs(b) + floor)))


def probe(name, t):
    from weights import probe_indices
    idx = probe_indices(name, t.numel())
    flat = t.detach().reshape(-1).double().cpu().numpy()
    return np.concatenate([[np.sqrt((flat ** 2).sum())], flat[idx]])


def masks_from_keep(keep):
    return [torch.tensor(k.astype(np.float32)) for k in keep]
