"""Samplers -- counterpart of the reference's lib/algorithms/advanced/sampling.py (registries
:33-77, get_sampling_fn :80-124, Predictor/Corrector :127-174, EulerMaruyamaPredictor :177-207,
ReverseDiffusion/Ancestral/None predictors :210-270, LangevinCorrector :273-302, ALD :305-339,
NoneCorrector :342-350, shared update fns :353-372, get_pc_sampler :375-468).

Fast path on MI355X: Euler-Maruyama predictor + 'none' corrector (the shipped configuration,
configs/subvp/amass_scorefc_continuous.py:30-32) on a sub-VP/VP SDE with a ScoreModelFC runs as
``dposer_em_sampler``: the whole N-step loop is enqueued by one C call, the time branch of the
network collapses into a per-step bias table, the predictor update / completion imputation /
re-tiling of x for the next step are fused into one elementwise kernel per step, and noise is
drawn in-kernel (Philox).  Any other predictor/corrector combination runs the generic loop below on
top of the HIP score function.
"""
import abc
import ctypes as C
import functools

import numpy as np
import torch

from ... import _C
from . import sde_lib
from . import utils as mutils
from .utils import get_score_fn

_CORRECTORS = {}
_PREDICTORS = {}


def _registrar(table):
    def register(cls=None, *, name=None):
        def _register(c):
            key = name if name is not None else c.__name__
            if key in table:
                raise ValueError(f"Already registered model with name: {key}")
            table[key] = c
            return c

        return _register if cls is None else _register(cls)

    return register


register_predictor = _registrar(_PREDICTORS)
register_corrector = _registrar(_CORRECTORS)


def get_predictor(name):
    return _PREDICTORS[name]


def get_corrector(name):
    return _CORRECTORS[name]


def get_sampling_fn(config, sde, shape, inverse_scaler, eps, device=None):
    """sampling.py:80-124."""
    if device is None:
        device = config.device
    name = config.sampling.method.lower()
    if name == "ode":
        return get_ode_sampler(sde=sde, shape=shape, inverse_scaler=inverse_scaler, denoise=config.sampling.noise_removal, eps=eps,
                               device=device)
    if name != "pc":
        raise ValueError(f"Sampler name {config.sampling.method} unknown.")
    return get_pc_sampler(sde=sde, shape=shape, predictor=get_predictor(config.sampling.predictor.lower()),
                          corrector=get_corrector(config.sampling.corrector.lower()), inverse_scaler=inverse_scaler,
                          snr=config.sampling.snr, n_steps=config.sampling.n_steps_each,
                          probability_flow=config.sampling.probability_flow, continuous=config.training.continuous,
                          denoise=config.sampling.noise_removal, eps=eps, device=device)


class Predictor(abc.ABC):
    def __init__(self, sde, score_fn, probability_flow=False):
        super().__init__()
        self.sde = sde
        self.rsde = sde.reverse(score_fn, probability_flow)
        self.score_fn = score_fn

    @abc.abstractmethod
    def update_fn(self, x, t, observation, mask):
        ...


class Corrector(abc.ABC):
    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__()
        self.sde, self.score_fn, self.snr, self.n_steps = sde, score_fn, snr, n_steps

    @abc.abstractmethod
    def update_fn(self, x, t, observation, mask):
        ...


@register_predictor(name="euler_maruyama")
class EulerMaruyamaPredictor(Predictor):
    """sampling.py:177-207."""

    def update_fn(self, x, t, observation, mask):
        dt = -1.0 / self.rsde.N
        z = torch.randn_like(x)
        drift, diffusion = self.rsde.sde(x, t)
        x_mean = x + drift * dt
        return x_mean + diffusion[:, None] * np.sqrt(-dt) * z, x_mean

    def update_fn_guide(self, x_t, t, observation, mask, condition=None, grad_step=1.0):
        """MCG / DPS style guided step (sampling.py:191-207); needs d score / d x -> differentiable HIP forward."""
        import contextlib
        x_t.requires_grad_()
        dt = -1.0 / self.rsde.N
        z = torch.randn_like(x_t)
        # only d score / d x is ever asked for here: inside input_grad_only() the backward skips the weight-gradient GEMMs
        model = getattr(self.score_fn, "model", None)
        only_x = model.input_grad_only() if hasattr(model, "input_grad_only") else contextlib.nullcontext()
        with only_x:
            drift, diffusion, alpha, sigma_2, score = self.rsde.sde(x_t, t, condition, mask, guide=True)
        y_mean = x_t.detach() + drift.detach() * dt
        y_hat = y_mean + diffusion[:, None] * np.sqrt(-dt) * z
        with torch.enable_grad():
            y0 = (x_t + sigma_2[:, None] * score) / alpha
            norm = torch.norm((observation * mask) - (y0 * mask))
            g = torch.autograd.grad(outputs=norm, inputs=x_t)[0]
            if torch.isnan(g).any():
                raise ValueError("Consider reduce the value of parameter: grad_step={}".format(grad_step))
            y_hat = y_hat - grad_step * g
        return y_hat, y_mean


@register_predictor(name="reverse_diffusion")
class ReverseDiffusionPredictor(Predictor):
    """sampling.py:210-220."""

    def update_fn(self, x, t, observation=None, mask=None):
        f, G = self.rsde.discretize(x, t)
        z = torch.randn_like(x)
        x_mean = x - f
        return x_mean + G[:, None] * z, x_mean


@register_predictor(name="ancestral_sampling")
class AncestralSamplingPredictor(Predictor):
    """sampling.py:223-259 (VE / VP only)."""

    def __init__(self, sde, score_fn, probability_flow=False):
        super().__init__(sde, score_fn, probability_flow)
        if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE)):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")
        assert not probability_flow, "Probability flow not supported by ancestral sampling"

    def update_fn(self, x, t, observation=None, mask=None):
        sde = self.sde
        timestep = (t * (sde.N - 1) / sde.T).long()
        score = self.score_fn(x, t)
        noise = torch.randn_like(x)
        if isinstance(sde, sde_lib.VESDE):
            sigma = sde.discrete_sigmas.to(t.device)[timestep]
            adj = torch.where(timestep == 0, torch.zeros_like(t), sde.discrete_sigmas.to(t.device)[timestep - 1])
            x_mean = x + score * (sigma ** 2 - adj ** 2)[:, None]
            std = torch.sqrt((adj ** 2 * (sigma ** 2 - adj ** 2)) / (sigma ** 2))
            return x_mean + std[:, None] * noise, x_mean
        beta = sde.discrete_betas.to(t.device)[timestep]
        x_mean = (x + beta[:, None] * score) / torch.sqrt(1.0 - beta)[:, None]
        return x_mean + torch.sqrt(beta)[:, None] * noise, x_mean


@register_predictor(name="none")
class NonePredictor(Predictor):
    def __init__(self, sde, score_fn, probability_flow=False):
        pass

    def update_fn(self, x, t, observation, mask):
        return x, x


def _langevin_alpha(sde, t):
    if isinstance(sde, (sde_lib.VPSDE, sde_lib.subVPSDE)):
        return sde.alphas.to(t.device)[(t * (sde.N - 1) / sde.T).long()]
    return torch.ones_like(t)


def _check_langevin_sde(sde):
    if not isinstance(sde, (sde_lib.VPSDE, sde_lib.VESDE, sde_lib.subVPSDE)):
        raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")


@register_corrector(name="langevin")
class LangevinCorrector(Corrector):
    """sampling.py:273-302.  The step size couples the samples through batch-mean norms (:296-297).  This class is the generic
    form (any score function, means over the tensor it is given); ``pc_sampler`` runs Langevin + Euler-Maruyama on the HIP
    path instead (``fused_pc_langevin_sample``), where the means are GLOBAL-batch means also under data parallelism: the two
    norm sums are all-reduced between the two phases of ``dposer_langevin_step``."""

    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        _check_langevin_sde(sde)

    def update_fn(self, x, t, observation, mask):
        alpha = _langevin_alpha(self.sde, t)
        x_mean = x
        for _ in range(self.n_steps):
            grad = self.score_fn(x, t, condition=None, mask=mask)
            noise = torch.randn_like(x)
            grad_norm = torch.norm(grad.reshape(grad.shape[0], -1), dim=-1).mean()
            noise_norm = torch.norm(noise.reshape(noise.shape[0], -1), dim=-1).mean()
            step = (self.snr * noise_norm / grad_norm) ** 2 * 2 * alpha
            x_mean = x + step[:, None] * grad
            x = x_mean + torch.sqrt(step * 2)[:, None] * noise
        return x, x_mean


@register_corrector(name="ald")
class AnnealedLangevinDynamics(Corrector):
    """sampling.py:305-339."""

    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__(sde, score_fn, snr, n_steps)
        _check_langevin_sde(sde)

    def update_fn(self, x, t, observation, mask):
        alpha = _langevin_alpha(self.sde, t)
        std = self.sde.marginal_prob(x, t)[1]
        x_mean = x
        for _ in range(self.n_steps):
            grad = self.score_fn(x, t, condition=None, mask=mask)
            noise = torch.randn_like(x)
            step = (self.snr * std) ** 2 * 2 * alpha
            x_mean = x + step[:, None] * grad
            x = x_mean + noise * torch.sqrt(step * 2)[:, None]
        return x, x_mean


@register_corrector(name="none")
class NoneCorrector(Corrector):
    def __init__(self, sde, score_fn, snr, n_steps):
        pass

    def update_fn(self, x, t, observation, mask):
        return x, x


def shared_predictor_update_fn(x, t, observation, mask, sde, model, predictor, probability_flow, continuous):
    """sampling.py:353-361."""
    score_fn = mutils.get_score_fn(sde, model, train=False, continuous=continuous)
    obj = NonePredictor(sde, score_fn, probability_flow) if predictor is None else predictor(sde, score_fn, probability_flow)
    return obj.update_fn(x, t, observation, mask)


def shared_corrector_update_fn(x, t, observation, mask, sde, model, corrector, continuous, snr, n_steps):
    """sampling.py:364-372."""
    score_fn = mutils.get_score_fn(sde, model, train=False, continuous=continuous)
    obj = NoneCorrector(sde, score_fn, snr, n_steps) if corrector is None else corrector(sde, score_fn, snr, n_steps)
    return obj.update_fn(x, t, observation, mask)


def fused_em_supported(sde, model, predictor, corrector, probability_flow, continuous):
    from .model import ScoreModelFC
    return (predictor is EulerMaruyamaPredictor and corrector in (None, NoneCorrector) and not probability_flow
            and sde_lib.sde_desc(sde, continuous) is not None
            and isinstance(model, ScoreModelFC))


def fused_em_sample(model, sde, x, timesteps, *, start_step=0, observation=None, mask=None, noise=None, seed=0,
                    traj_stride=0, continuous=True):
    """dposer_em_sampler.  x [B, D] initial state (consumed); returns (trajs or None, x, x_mean)."""
    _C.require_gpu(x, "sampler state")
    eng = model._engine()
    flat = model.flat_params()
    packed = eng.packed(flat, with_backward=False, force=not model.freeze_packed)
    B, D = x.shape
    N = int(sde.N)
    n_run = N - start_step
    if B == 0:                      # nothing to sample: the reference's loop runs on empty tensors and returns them
        x = x.contiguous().float().clone()
        traj = torch.empty((max(n_run, 0) // traj_stride, 0, D), dtype=torch.float32, device=x.device) if traj_stride and n_run > 0 else None
        return traj, x, x.clone()
    ws = eng.workspace(B, _C.WS_SHARED_T, max(n_run, 1), x.device)
    x = x.contiguous().float().clone()
    x_mean = x.clone()
    ts_host = timesteps.detach().to("cpu", torch.float32).contiguous()
    traj = None
    if traj_stride and n_run > 0:
        traj = torch.empty((n_run // traj_stride, B, D), dtype=torch.float32, device=x.device)
    desc = sde_lib.sde_desc(sde, continuous)
    obs = None if observation is None else observation.contiguous().float()
    msk = None if mask is None else mask.contiguous().float()
    nz = None if noise is None else noise.contiguous().float()
    _C.check(eng.lib.dposer_em_sampler(eng.h, _C.ptr(flat), _C.ptr(packed), _C.ptr(ws), C.byref(desc), _C.ptr(x), _C.ptr(x_mean),
                                       C.c_void_p(ts_host.data_ptr()), int(start_step), _C.ptr(obs), _C.ptr(msk), _C.ptr(nz),
                                       int(seed), _C.ptr(traj), int(traj_stride or 1), _C.ptr(eng.freq(x.device, model._fourier_W())),
                                       _C.ptr(model.sigmas), B, _C.stream_ptr()), "dposer_em_sampler")
    return traj, x, x_mean


def fused_langevin_supported(sde, model, predictor, corrector, probability_flow, continuous):
    from .model import ScoreModelFC
    return (predictor is EulerMaruyamaPredictor and corrector is LangevinCorrector and not probability_flow
            and sde_lib.sde_desc(sde, continuous) is not None
            and isinstance(model, ScoreModelFC))


def fused_pc_langevin_sample(model, sde, x, timesteps, *, snr, n_steps=1, start_step=0, observation=None, mask=None, noise=None,
                             seed=0, traj_stride=0, continuous=True):
    """Predictor-corrector loop of sampling.py:455-461 with the Langevin corrector (:282-302) and the Euler-Maruyama predictor
    on the HIP path.  Per outer step: ``n_steps`` x [``dposer_langevin_step`` phase 0 -> all-reduce of the two norm sums over
    the data-parallel ranks -> phase 1], then ``dposer_em_sampler_steps`` for the (imputation,) predictor (, imputation) of that
    step.  Nothing synchronises with the host inside the loop.
    ``noise`` [n_run, n_steps + (3 if completion else 1), B, D]: injected draws in the reference's order (tests)."""
    from ... import distributed as ddp
    _C.require_gpu(x, "sampler state")
    eng = model._engine()
    flat = model.flat_params()
    packed = eng.packed(flat, with_backward=False, force=not model.freeze_packed)
    B, D = x.shape
    N = int(sde.N)
    n_run = N - start_step
    ws = eng.workspace(B, _C.WS_SHARED_T, 1, x.device)
    x = x.contiguous().float().clone()
    x_mean = x.clone()
    ts_host = timesteps.detach().to("cpu", torch.float32).contiguous()
    desc = sde_lib.sde_desc(sde, continuous)
    obs = None if observation is None else observation.contiguous().float()
    msk = None if mask is None else mask.contiguous().float()
    nz = None if noise is None else noise.contiguous().float()
    k_pred = 3 if obs is not None else 1
    dp = ddp.dp_active()
    global_batch = B
    if dp:                                         # ragged shards: the global batch is the sum of the local ones (once per call)
        cnt = torch.tensor([float(B)], dtype=torch.float64, device=x.device)     # on the device: RCCL cannot reduce a CPU tensor
        ddp.all_reduce_sum_(cnt)
        global_batch = int(cnt.item())
    norms = torch.empty(2, dtype=torch.float32, device=x.device)
    alphas = sde.alphas.detach().to("cpu") if hasattr(sde, "alphas") else None
    traj = torch.empty((n_run // traj_stride, B, D), dtype=torch.float32, device=x.device) if (traj_stride and n_run > 0) else None
    freq, lib = eng.freq(x.device, model._fourier_W()), eng.lib
    for i in range(n_run):
        gi = start_step + i
        t = ts_host[gi]
        alpha = float(alphas[int((t * (sde.N - 1) / sde.T).long())]) if alphas is not None else 1.0      # sampling.py:290-294
        for k in range(n_steps):
            z = None if nz is None else nz[i, k]
            args = (eng.h, _C.ptr(flat), _C.ptr(packed), _C.ptr(ws), C.byref(desc), _C.ptr(x), _C.ptr(x_mean), float(t), alpha, float(snr),
                    _C.ptr(z), int(seed), (gi * n_steps + k) & 0xFFFFFFFF, _C.ptr(norms))
            _C.check(lib.dposer_langevin_step(*args, 0, 1.0 / global_batch, _C.ptr(freq), _C.ptr(model.sigmas), B, _C.stream_ptr()),
                     "dposer_langevin_step")
            if dp:
                ddp.all_reduce_sum_(norms)         # two floats: keeps the reference's global-batch means under DP
            _C.check(lib.dposer_langevin_step(*args, 1, 1.0 / global_batch, _C.ptr(freq), _C.ptr(model.sigmas), B, _C.stream_ptr()),
                     "dposer_langevin_step")
        zp = None if nz is None else nz[i, n_steps:n_steps + k_pred].contiguous()
        slot = traj[(i + 1) // traj_stride - 1] if (traj is not None and (i + 1) % traj_stride == 0) else None
        _C.check(lib.dposer_em_sampler_steps(eng.h, _C.ptr(flat), _C.ptr(packed), _C.ptr(ws), C.byref(desc), _C.ptr(x), _C.ptr(x_mean),
                                             C.c_void_p(ts_host.data_ptr()), int(gi), 1, _C.ptr(obs), _C.ptr(msk), _C.ptr(zp), int(seed),
                                             _C.ptr(slot), 1, _C.ptr(freq), _C.ptr(model.sigmas), B, _C.stream_ptr()),
                 "dposer_em_sampler_steps")
    return traj, x, x_mean


def get_pc_sampler(sde, shape, predictor, corrector, inverse_scaler, snr, n_steps=1, probability_flow=False,
                   continuous=False, denoise=True, eps=1e-3, device="cuda"):
    """Predictor-corrector sampler factory (sampling.py:375-468).

    The returned ``pc_sampler(model, observation, mask, z, start_step, args)`` has the reference's
    signature and return value ``(trajs [n, B, D], x_mean if denoise else x)``.  Extra keyword-only
    arguments: ``traj_stride`` (default 1 = every step as the reference; 0 = keep no trajectory --
    at B = 65536, N = 1000 the full tensor is 16.5 GB), ``noise`` (injected draws for tests),
    ``seed`` (Philox key of the in-kernel noise)."""
    predictor_update_fn = functools.partial(shared_predictor_update_fn, sde=sde, predictor=predictor,
                                            probability_flow=probability_flow, continuous=continuous)
    corrector_update_fn = functools.partial(shared_corrector_update_fn, sde=sde, corrector=corrector, continuous=continuous,
                                            snr=snr, n_steps=n_steps)

    def with_imputation(update_fn):
        def fn(x, vec_t, observation, mask, model, args):
            x, x_mean = update_fn(x, vec_t, observation, mask, model=model)
            if args is not None and args.task in ["completion"]:                 # sampling.py:416-420
                mean, std = sde.marginal_prob(observation, vec_t)
                x = x * (1 - mask) + (mean + torch.randn_like(x) * std[:, None]) * mask
            return x, x_mean

        return fn

    projector = with_imputation(predictor_update_fn)
    correct = with_imputation(corrector_update_fn)
    call_count = [0]

    def pc_sampler(model, observation=None, mask=None, z=None, start_step=0, args=None, *, traj_stride=1, noise=None,
                   seed=None):
        with torch.no_grad():
            # sampling.py:446: the prior is drawn by the SDE (VESDE scales by sigma_max) from torch's CPU generator, so a seeded
            # run starts from the reference's own x_T; callers that keep everything on the device pass z
            x = sde.prior_sampling(shape).to(device) if z is None else z
            timesteps = torch.linspace(sde.T, eps, sde.N, device=device)          # sampling.py:449
            start_t = start_step if (args is not None and args.task in ["denoise"]) else 0
            completion = args is not None and args.task in ["completion"]
            if fused_em_supported(sde, model, predictor, corrector, probability_flow, continuous):
                call_count[0] += 1
                if seed is None:
                    seed = (model._rng_seed * 7919 + call_count[0]) & 0xFFFFFFFFFFFF
                was_training = model.training
                model.eval()                                                       # utils.py:117-119
                trajs, x, x_mean = fused_em_sample(model, sde, x, torch.linspace(sde.T, eps, sde.N), start_step=start_t,
                                                   observation=observation if completion else None,
                                                   mask=mask if completion else None, noise=noise, seed=seed,
                                                   traj_stride=traj_stride, continuous=continuous)
                model.train(was_training)
                if trajs is None:
                    trajs = x.new_empty((0,) + tuple(x.shape))
                return trajs, (x_mean if denoise else x)
            if fused_langevin_supported(sde, model, predictor, corrector, probability_flow, continuous):
                call_count[0] += 1
                if seed is None:
                    seed = (model._rng_seed * 7919 + call_count[0]) & 0xFFFFFFFFFFFF
                was_training = model.training
                model.eval()
                trajs, x, x_mean = fused_pc_langevin_sample(model, sde, x, torch.linspace(sde.T, eps, sde.N), snr=snr, n_steps=n_steps,
                                                            start_step=start_t, observation=observation if completion else None,
                                                            mask=mask if completion else None, noise=noise, seed=seed,
                                                            traj_stride=traj_stride, continuous=continuous)
                model.train(was_training)
                if trajs is None:
                    trajs = x.new_empty((0,) + tuple(x.shape))
                return trajs, (x_mean if denoise else x)
            trajs = []
            x_mean = x
            for i in range(start_t, sde.N):
                vec_t = torch.ones(shape[0], device=device) * timesteps[i]
                x, x_mean = correct(x, vec_t, observation, mask, model=model, args=args)
                x, x_mean = projector(x, vec_t, observation, mask, model=model, args=args)
                if traj_stride and (i - start_t + 1) % traj_stride == 0:
                    trajs.append(x)
            trajs = torch.stack(trajs, dim=0) if trajs else x.new_empty((0,) + tuple(x.shape))
            return trajs, (x_mean if denoise else x)

    return pc_sampler


def get_ode_sampler(sde, shape, inverse_scaler, denoise=False, rtol=1e-5, atol=1e-5, method="RK45", eps=1e-3, device="cuda", driver=None,
                    n_steps=None):
    """Probability-flow ODE sampler (sampling.py:471-542): ``ode_sampler(model, z=None) -> (nfe, samples)``.

    The adaptive RK45 steps follow scipy's controller exactly like the reference (tolerances and ``nfe`` are part of the
    result); with ``driver='device'`` (default for RK45, ``likelihood.get_likelihood_fn`` has the details) the state never leaves
    the GPU, with ``driver='scipy'`` ``scipy.integrate.solve_ivp`` drives it from the host.  Each drift evaluation is one HIP
    forward of the score network.  ``denoise`` adds one noise-free reverse-diffusion predictor step at ``eps`` (:492-499)."""
    import os
    from .likelihood import FusedPfRhs, probability_flow_drift
    from scipy import integrate
    fixed = method in ("rk4", "euler")      # fixed-step, fully asynchronous device integration (likelihood.get_likelihood_fn has the details)
    if fixed and not n_steps:
        raise ValueError(f"method={method!r} is a fixed-step integrator: pass n_steps")
    driver = driver or os.environ.get("DPOSER_ODE_DRIVER") or ("device" if (method == "RK45" or fixed) else "scipy")
    if driver == "device" and not (method == "RK45" or fixed):
        raise NotImplementedError("the device-resident driver implements RK45 (the reference's default), rk4 and euler; use driver='scipy'")
    if fixed and driver != "device":
        raise NotImplementedError("fixed-step methods run on the device-resident driver")

    def ode_sampler(model, z=None):
        with torch.no_grad():
            x = sde.prior_sampling(shape).to(device) if z is None else z

            def rhs_generic(t, state):
                xt = state.reshape(shape).float()
                vec_t = torch.full((shape[0],), float(t), device=device, dtype=torch.float32)
                return probability_flow_drift(sde, model, xt, vec_t).reshape(-1).double()

            rhs_dev = FusedPfRhs.build(sde, model, tuple(x.shape), x.device) or rhs_generic      # three launches around the forward

            if driver == "device":
                from .ode_device import solve_fixed, solve_rk45
                if fixed:
                    end, nfev = solve_fixed(rhs_dev, sde.T, eps, x.reshape(-1).double(), int(n_steps), method=method)
                else:
                    end, nfev = solve_rk45(rhs_dev, sde.T, eps, x.reshape(-1).double(), rtol=rtol, atol=atol)
                x = end.reshape(shape).float()
            else:
                sol = integrate.solve_ivp(lambda t, s_: rhs_dev(t, torch.from_numpy(s_).to(device)).cpu().numpy(), (sde.T, eps),
                                          mutils.to_flattened_numpy(x), rtol=rtol, atol=atol, method=method)
                nfev = sol.nfev
                x = torch.from_numpy(sol.y[:, -1].reshape(shape)).to(device, torch.float32)
            if denoise:
                score_fn = get_score_fn(sde, model, train=False, continuous=True)
                vec_eps = torch.full((shape[0],), float(eps), device=device, dtype=torch.float32)
                x = ReverseDiffusionPredictor(sde, score_fn, probability_flow=False).update_fn(x, vec_eps)[1]
            return nfev, inverse_scaler(x)

    return ode_sampler
