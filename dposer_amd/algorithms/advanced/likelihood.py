"""Log-likelihood under the probability-flow ODE (reference lib/algorithms/advanced/likelihood.py:25-113).

``get_likelihood_fn(sde, inverse_scaler, hutchinson_type, rtol, atol, method, eps)(model, data) -> (bpd, z, nfe)``
exactly as ``run/train.py:235,279`` and ``run/demo.py:121,432`` call it.  The adaptive Runge-Kutta driver is
``scipy.integrate.solve_ivp`` on the host, as in the reference (its step-size control *is* the semantics: rtol /
atol / nfe are part of the result); every right-hand-side evaluation -- the probability-flow drift and its
Hutchinson divergence estimate -- is one HIP forward plus one HIP input-gradient (vector-Jacobian product) of the
score network through ``ScoreModelFC``'s autograd function; only the flattened state crosses PCIe per evaluation.
"""
import contextlib

import numpy as np
import torch
from scipy import integrate

from . import utils as mutils


def probability_flow_drift(sde, model, x, t):
    """Drift of the probability-flow ODE, d x / d t = f(x, t) - 1/2 g(t)^2 score(x, t) (likelihood.py:60-65)."""
    score_fn = mutils.get_score_fn(sde, model, train=False, continuous=True)
    return sde.reverse(score_fn, probability_flow=True).sde(x, t, condition=None, mask=None)[0]


def get_div_fn(fn):
    """Hutchinson-Skilling estimate of div fn: eps^T (d fn / d x) eps, one value per sample (likelihood.py:25-37)."""

    def div_fn(x, t, eps):
        with torch.enable_grad():
            xg = x.detach().requires_grad_(True)
            projected = (fn(xg, t) * eps).sum()
            vjp, = torch.autograd.grad(projected, xg)
        return (vjp * eps).flatten(1).sum(dim=1)

    return div_fn


def hutchinson_noise(data, kind):
    if kind == "Gaussian":
        return torch.randn_like(data)
    if kind == "Rademacher":
        return torch.randint_like(data, low=0, high=2).float() * 2 - 1.0
    raise NotImplementedError(f"Hutchinson type {kind} unknown.")


def get_likelihood_fn(sde, inverse_scaler, hutchinson_type="Rademacher", rtol=1e-5, atol=1e-5, method="RK45", eps=1e-5):
    """Returns ``likelihood_fn(model, data) -> (bpd [B], z like data, nfe)`` (likelihood.py:40-113)."""

    def likelihood_fn(model, data, *, epsilon=None):
        shape, B = tuple(data.shape), data.shape[0]
        dev = data.device
        n_state = int(np.prod(shape))
        with torch.no_grad():
            noise = hutchinson_noise(data, hutchinson_type) if epsilon is None else epsilon
        div = get_div_fn(lambda xx, tt: probability_flow_drift(sde, model, xx, tt))

        def rhs(t, state):
            x = torch.from_numpy(state[:n_state].reshape(shape)).to(dev, torch.float32)
            vec_t = torch.full((B,), float(t), device=dev, dtype=torch.float32)
            with torch.no_grad():
                drift = probability_flow_drift(sde, model, x, vec_t)
            with (model.input_grad_only() if hasattr(model, "input_grad_only") else contextlib.nullcontext()):
                dlogp = div(x, vec_t, noise)
            return np.concatenate([mutils.to_flattened_numpy(drift), mutils.to_flattened_numpy(dlogp)])

        init = np.concatenate([mutils.to_flattened_numpy(data), np.zeros((B,))])
        sol = integrate.solve_ivp(rhs, (eps, sde.T), init, rtol=rtol, atol=atol, method=method)
        end = sol.y[:, -1]
        with torch.no_grad():
            z = torch.from_numpy(end[:n_state].reshape(shape)).to(dev, torch.float32)
            delta_logp = torch.from_numpy(end[n_state:]).to(dev, torch.float32)
            bpd = -(sde.prior_logp(z) + delta_logp) / np.log(2)
            bpd = bpd / np.prod(shape[1:])
        return bpd, z, sol.nfev

    return likelihood_fn
