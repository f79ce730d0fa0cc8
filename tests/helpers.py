"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


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
