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


class FusedPfRhs:
    """One right-hand side of the probability-flow ODE as four library calls (``dposer_pf_ode_rhs_begin`` -> score network forward
    [-> input-gradient] -> ``dposer_pf_ode_rhs_end``) instead of the ~60 small torch launches of the expression-by-expression path
    (``probability_flow_drift`` + autograd), which is bound by host enqueue time below ~16k poses.  Same fp32 operation order as that
    path (the reference's, likelihood.py:60-65 / sde_lib.py:100-104 / utils.py:152-162); only the 63-term Hutchinson sum per sample is
    formed in a different order.  Covers ScoreModelFC with a VP / sub-VP / (continuous) VE SDE; ``build`` returns None otherwise (or with
    ``DPOSER_ODE_FUSED_RHS=0``) and the caller keeps the generic path.

    ``noise`` given: state [B*D + B] -> [d x / dt, d logp / dt] (likelihood.py:86-95); ``noise=None``: state [B*D] -> drift
    (sampling.py:513-518).  The weights are packed once per solve (they cannot change inside one)."""

    @staticmethod
    def build(sde, model, shape, device, noise=None):
        from ... import _C
        from .model import ScoreModelFC
        from .sde_lib import sde_desc
        desc = sde_desc(sde)
        device = torch.device(device)
        if (os.environ.get("DPOSER_ODE_FUSED_RHS", "1") == "0" or desc is None or not isinstance(model, ScoreModelFC) or len(shape) != 2
                or device.type != "cuda" or shape[1] != model._engine().D):
            return None
        return FusedPfRhs(_C, desc, model, shape, device, noise)

    def __init__(self, _C, desc, model, shape, device, noise):
        self._C, self.desc, self.model = _C, desc, model
        self.B, self.D = int(shape[0]), int(shape[1])
        model.eval()                                       # get_score_fn(train=False) (utils.py:121-124)
        if model.sigmas.device != device or model._param_list[0].device != device:
            raise _C.DPoserHipError("ScoreModelFC parameters and the ODE state are on different devices")
        eng = self.eng = model._engine()
        self.flat = model.flat_params()
        self.with_grad = noise is not None
        self.packed = eng.packed(self.flat, with_backward=self.with_grad, force=not model.freeze_packed)
        self.freq = eng.freq(device, model._fourier_W())
        f32 = dict(dtype=torch.float32, device=device)
        self.x, self.out = torch.empty(self.B, self.D, **f32), torch.empty(self.B, self.D, **f32)
        self.labels = torch.empty(self.B, **f32)
        if self.with_grad:
            self.noise = noise.detach().reshape(self.B, self.D).contiguous().float()
            self.dout, self.dx = torch.empty(self.B, self.D, **f32), torch.empty(self.B, self.D, **f32)
            self.lease = eng.lease_train_workspace(self.B, device)
            self.ws = self.lease.ws
        else:
            self.noise = self.dout = self.dx = self.lease = None
            self.ws = eng.workspace(self.B, _C.WS_INFER, 0, device)
        self.n_out = self.B * self.D + (self.B if self.with_grad else 0)

    def __call__(self, t, state):
        _C, lib, m, B, D = self._C, self.eng.lib, self.model, self.B, self.D
        ptr, st, h = _C.ptr, _C.stream_ptr(), self.eng.h
        state = state.to(torch.float64).contiguous()
        dstate = torch.empty(self.n_out, dtype=torch.float64, device=state.device)
        t = float(t)
        _C.check(lib.dposer_pf_ode_rhs_begin(self.desc, t, ptr(state), ptr(self.noise), ptr(self.x), ptr(self.labels), ptr(self.dout), B, D, st),
                 "dposer_pf_ode_rhs_begin")
        if self.with_grad:
            m._rng_step += 1
            _C.check(lib.dposer_scorefc_forward_train(h, ptr(self.flat), ptr(self.packed), ptr(self.ws), ptr(self.x), ptr(self.labels),
                                                      ptr(self.freq), ptr(m.sigmas), ptr(self.out), B, 0, m._rng_seed, m._rng_step, st),
                     "dposer_scorefc_forward_train")
            _C.check(lib.dposer_scorefc_backward(h, ptr(self.flat), ptr(self.packed), ptr(self.ws), ptr(self.labels), ptr(m.sigmas),
                                                 ptr(self.dout), None, ptr(self.dx), B, 0, m._rng_seed, m._rng_step, st),
                     "dposer_scorefc_backward")
        else:
            _C.check(lib.dposer_scorefc_forward(h, ptr(self.flat), ptr(self.packed), ptr(self.ws), ptr(self.x), ptr(self.labels),
                                                ptr(self.freq), ptr(m.sigmas), ptr(self.out), B, st), "dposer_scorefc_forward")
        _C.check(lib.dposer_pf_ode_rhs_end(self.desc, t, ptr(self.x), ptr(self.out), ptr(self.dx), ptr(self.noise), ptr(dstate), B, D, st),
                 "dposer_pf_ode_rhs_end")
        return dstate


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

        def rhs_generic(t, state):
            """state float64 [n_state + B] on the device -> d state / dt (likelihood.py:86-95): drift and divergence estimate from
            one differentiable evaluation."""
            vec_t = torch.full((B,), float(t), device=dev, dtype=torch.float32)
            with torch.enable_grad(), only_x():
                xg = state[:n_state].reshape(shape).float().requires_grad_(True)
                drift = probability_flow_drift(sde, model, xg, vec_t)
                vjp, = torch.autograd.grad((drift * noise).sum(), xg)
            dlogp = (vjp * noise).flatten(1).sum(dim=1)
            return torch.cat([drift.detach().reshape(-1).double(), dlogp.reshape(-1).double()])

        rhs_dev = FusedPfRhs.build(sde, model, shape, dev, noise) or rhs_generic

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
