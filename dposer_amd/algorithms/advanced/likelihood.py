"""Log-likelihood under the probability-flow ODE (reference lib/algorithms/advanced/likelihood.py:25-113).

``get_likelihood_fn(sde, inverse_scaler, hutchinson_type, rtol, atol, method, eps)(model, data) -> (bpd, z, nfe)``
exactly as ``run/train.py:235,279`` and ``run/demo.py:121,432`` call it.  The adaptive Runge-Kutta driver is
``scipy.integrate.solve_ivp`` on the host, as in the reference (its step-size control *is* the semantics: rtol /
atol / nfe are part of the result); every right-hand-side evaluation -- the probability-flow drift and its
Hutchinson divergence estimate -- is one HIP forward plus one HIP input-gradient (vector-Jacobian product) of the
score network through ``ScoreModelFC``'s autograd function; only the flattened state crosses PCIe per evaluation.
"""
import contextlib
import os

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


FIXED_STEP_METHODS = ("rk4", "euler")


def get_likelihood_fn(sde, inverse_scaler, hutchinson_type="Rademacher", rtol=1e-5, atol=1e-5, method="RK45", eps=1e-5, driver=None,
                      n_steps=None):
    """Returns ``likelihood_fn(model, data) -> (bpd [B], z like data, nfe)`` (likelihood.py:40-113).

    ``driver``: 'device' (default for RK45) integrates with ``ode_device.solve_rk45`` -- scipy's RK45 controller with the state and
    all stage arithmetic resident on the GPU in float64, one scalar to the host per attempted step; 'scipy' is the reference's
    own driver (host float64 state, two state copies per right-hand-side evaluation; the only choice for other ``method``s).
    ``DPOSER_ODE_DRIVER`` overrides the default.
    ``method='rk4' | 'euler'`` with ``n_steps``: fixed-step integration on the device (``ode_device.solve_fixed``) -- no step-size
    control, hence no host synchronisation at all: the whole likelihood evaluation is queued asynchronously; nfe =
    4 n_steps / n_steps.  Not a reference mode (its driver is always solve_ivp): for throughput runs such as the validation bpd
    of run/train.py:279 on large batches.

    One right-hand side is ONE network evaluation: the forward keeps its activations, the drift is its output and the Hutchinson
    vector-Jacobian product eps^T d drift / d x is the input-gradient (dgrad only, no parameter gradients) of the same evaluation
    (round 2 ran a second, inference-mode forward for the drift)."""
    fixed = method in FIXED_STEP_METHODS
    if fixed and not n_steps:
        raise ValueError(f"method={method!r} is a fixed-step integrator: pass n_steps")
    driver = driver or os.environ.get("DPOSER_ODE_DRIVER") or ("device" if (method == "RK45" or fixed) else "scipy")
    if driver == "device" and not (method == "RK45" or fixed):
        raise NotImplementedError("the device-resident driver implements RK45 (the reference's default), rk4 and euler; use driver='scipy'")
    if fixed and driver != "device":
        raise NotImplementedError("fixed-step methods run on the device-resident driver")

    def likelihood_fn(model, data, *, epsilon=None):
        shape, B = tuple(data.shape), data.shape[0]
        dev = data.device
        n_state = int(np.prod(shape))
        with torch.no_grad():
            noise = hutchinson_noise(data, hutchinson_type) if epsilon is None else epsilon
        only_x = (lambda: model.input_grad_only()) if hasattr(model, "input_grad_only") else contextlib.nullcontext

        def rhs_dev(t, state):
            """state float64 [n_state + B] on the device -> d state / dt (likelihood.py:86-95): drift and divergence estimate from
            one differentiable evaluation."""
            vec_t = torch.full((B,), float(t), device=dev, dtype=torch.float32)
            with torch.enable_grad(), only_x():
                xg = state[:n_state].reshape(shape).float().requires_grad_(True)
                drift = probability_flow_drift(sde, model, xg, vec_t)
                vjp, = torch.autograd.grad((drift * noise).sum(), xg)
            dlogp = (vjp * noise).flatten(1).sum(dim=1)
            return torch.cat([drift.detach().reshape(-1).double(), dlogp.reshape(-1).double()])

        if driver == "device":
            from .ode_device import solve_fixed, solve_rk45
            init = torch.cat([data.detach().reshape(-1).double(), torch.zeros(B, dtype=torch.float64, device=dev)])
            if fixed:
                end, nfev = solve_fixed(rhs_dev, eps, sde.T, init, int(n_steps), method=method)
            else:
                end, nfev = solve_rk45(rhs_dev, eps, sde.T, init, rtol=rtol, atol=atol)
            z = end[:n_state].reshape(shape).float()
            delta_logp = end[n_state:].float()
        else:
            def rhs(t, state):
                out = rhs_dev(t, torch.from_numpy(state).to(dev))
                return out.cpu().numpy()

            init = np.concatenate([mutils.to_flattened_numpy(data), np.zeros((B,))])
            sol = integrate.solve_ivp(rhs, (eps, sde.T), init, rtol=rtol, atol=atol, method=method)
            nfev = sol.nfev
            end = sol.y[:, -1]
            z = torch.from_numpy(end[:n_state].reshape(shape)).to(dev, torch.float32)
            delta_logp = torch.from_numpy(end[n_state:]).to(dev, torch.float32)
        with torch.no_grad():
            bpd = -(sde.prior_logp(z) + delta_logp) / np.log(2)
            bpd = bpd / np.prod(shape[1:])
        return bpd, z, nfev

    return likelihood_fn
