"""Deterministic, torch-RNG-independent ScoreModelFC weights shared by the golden generator and
the tests.  ``np.random.RandomState`` streams are frozen by numpy's compatibility policy, so the
33 MB of H=1024 weights never have to be committed -- only the seed.

Values follow nn.Linear's default scale (U(-1/sqrt(fan_in), 1/sqrt(fan_in))) but GroupNorm
affine parameters are randomised (gamma in [0.5,1.5], beta in [-0.2,0.2]) so that a kernel that
drops or transposes them cannot pass.
"""
import numpy as np
import torch


def scorefc_shapes(D=63, H=1024, E=512, n_blocks=2, fourier=False):
    """state_dict order and shapes of ScoreModelFC (reference model.py:98-139)."""
    shp = [("pre_dense.weight", (H, D)), ("pre_dense.bias", (H,)),
           ("pre_dense_t.weight", (H, E)), ("pre_dense_t.bias", (H,)),
           ("pre_dense_cond.weight", (H, H)), ("pre_dense_cond.bias", (H,)),
           ("pre_gnorm.weight", (H,)), ("pre_gnorm.bias", (H,))]
    if fourier:
        shp.append(("gauss_proj.W", (E // 2,)))
    shp += [("shared_time_embed.0.weight", (E, E)), ("shared_time_embed.0.bias", (E,))]
    for k in range(1, n_blocks + 1):
        for j in (1, 2):
            shp += [(f"b{k}_dense{j}.weight", (H, H)), (f"b{k}_dense{j}.bias", (H,)),
                    (f"b{k}_dense{j}_t.weight", (H, E)), (f"b{k}_dense{j}_t.bias", (H,)),
                    (f"b{k}_gnorm{j}.weight", (H,)), (f"b{k}_gnorm{j}.bias", (H,))]
    shp += [("post_dense.weight", (D, H)), ("post_dense.bias", (D,))]
    return shp


def make_weights(seed, D=63, H=1024, E=512, n_blocks=2, fourier=False, fourier_scale=16.0,
                 dtype=torch.float32):
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in scorefc_shapes(D, H, E, n_blocks, fourier):
        if name == "gauss_proj.W":
            w = rs.standard_normal(shape) * fourier_scale
        elif "gnorm" in name and name.endswith("weight"):
            w = rs.uniform(0.5, 1.5, size=shape)
        elif "gnorm" in name:
            w = rs.uniform(-0.2, 0.2, size=shape)
        else:
            fan_in = shape[1] if len(shape) == 2 else {"pre_dense.bias": D, "post_dense.bias": H,
                                                      "shared_time_embed.0.bias": E}.get(
                name, E if name.endswith("_t.bias") else H)
            b = 1.0 / np.sqrt(fan_in)
            w = rs.uniform(-b, b, size=shape)
        out[name] = torch.tensor(w.astype(np.float32)).to(dtype)
    return out


def timemlps_shapes(D=63, H=1024, n_blocks=2):
    """state_dict order and shapes of TimeMLPs (reference model.py:69-88): net.<index of the Linear inside the Sequential>."""
    idx = [0] + [2 + 3 * k for k in range(n_blocks)] + [2 + 3 * n_blocks]
    dims = [(H, D + 1)] + [(H, H)] * n_blocks + [(D, H)]
    shp = []
    for i, (o, k) in zip(idx, dims):
        shp += [(f"net.{i}.weight", (o, k)), (f"net.{i}.bias", (o,))]
    return shp


def make_mlp_weights(seed, D=63, H=1024, n_blocks=2):
    """nn.Linear's default scale U(-1/sqrt(fan_in), 1/sqrt(fan_in)) from a frozen numpy stream."""
    rs = np.random.RandomState(seed)
    out, fan_in = {}, None
    for name, shape in timemlps_shapes(D, H, n_blocks):
        if len(shape) == 2:
            fan_in = shape[1]
        b = 1.0 / np.sqrt(fan_in)
        out[name] = torch.tensor(rs.uniform(-b, b, size=shape).astype(np.float32))
    return out


def probe_indices(name, numel, n=48):
    """Indices at which per-tensor goldens (grads, updated params ...) are sampled."""
    rs = np.random.RandomState(abs(hash_name(name)) % (2 ** 31))
    return np.sort(rs.randint(0, numel, size=min(n, numel)))


def hash_name(name):
    h = 2166136261
    for ch in name.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h
