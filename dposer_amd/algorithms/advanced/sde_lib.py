"""Forward SDEs and the reverse-time SDE factory -- host-side mirror of the reference's
lib/algorithms/advanced/sde_lib.py (SDE :7-119, VPSDE :122-181, subVPSDE :184-231, VESDE :234-292).

These are the *scalar* schedules (a handful of fp32 torch ops on ``t``); the per-element work that
consumes them in the hot loops (perturbation, Euler-Maruyama update, imputation, Tweedie denoise) is
fused into HIP kernels (dposer_amd/csrc/elementwise.hip) which restate the same formulas.  The
classes keep the reference's attribute names (``beta_0``, ``beta_1``, ``N`` mutable after
construction, ``discrete_betas``, ``alphas`` ...) because callers poke at them directly
(run/smplify.py:41, run/motion_denoising.py:91).
"""
import abc
import math

import numpy as np
import torch


class SDE(abc.ABC):
    """A forward SDE dx = f(x,t) dt + g(t) dw on a mini-batch.  sde_lib.py:7-119."""

    def __init__(self, N):
        super().__init__()
        self.N = N

    @property
    @abc.abstractmethod
    def T(self):
        ...

    @abc.abstractmethod
    def sde(self, x, t):
        ...

    @abc.abstractmethod
    def marginal_prob(self, x, t):
        ...

    @abc.abstractmethod
    def return_alpha_sigma(self, t):
        ...

    def prior_sampling(self, shape):
        return torch.randn(*shape)

    def prior_logp(self, z):
        n = np.prod(z.shape[1:])
        return -n / 2.0 * np.log(2 * np.pi) - torch.sum(z ** 2, dim=1) / 2.0

    def discretize(self, x, t):
        """Euler-Maruyama discretisation x_{i+1} = x_i + f_i + G_i z_i (sde_lib.py:52-69)."""
        dt = 1 / self.N
        drift, diffusion = self.sde(x, t)
        return drift * dt, diffusion * torch.sqrt(torch.tensor(dt, device=t.device))

    def reverse(self, score_fn, probability_flow=False):
        """Reverse-time SDE / probability-flow ODE around ``score_fn`` (sde_lib.py:75-119)."""
        return _ReverseSDE(self, score_fn, probability_flow)


class _ReverseSDE:
    """What the reference builds as the inner class ``RSDE`` (sde_lib.py:88-117)."""

    def __init__(self, fwd, score_fn, probability_flow):
        self._fwd = fwd
        self._score_fn = score_fn
        self.N = fwd.N
        self.probability_flow = probability_flow

    @property
    def T(self):
        return self._fwd.T

    def sde(self, x, t, condition=None, mask=None, guide=False):
        drift, diffusion = self._fwd.sde(x, t)
        score = self._score_fn(x, t, condition, mask)
        drift = drift - diffusion[:, None] ** 2 * score * (0.5 if self.probability_flow else 1.0)
        if self.probability_flow:
            diffusion = torch.zeros(1, device=drift.device)
        if not guide:
            return drift, diffusion
        alpha, sigma = self._fwd.return_alpha_sigma(t)
        return drift, diffusion, alpha, sigma ** 2, score

    def discretize(self, x, t, condition=None, mask=None):
        f, G = self._fwd.discretize(x, t)
        rev_f = f - G[:, None] ** 2 * self._score_fn(x, t, condition, mask)
        rev_G = torch.zeros_like(G) if self.probability_flow else G
        return rev_f, rev_G

    def __getattr__(self, name):          # beta_0, marginal_prob ... fall through to the forward SDE
        return getattr(self._fwd, name)


class _LinearBeta(SDE):
    """Shared linear-beta schedule of VP / sub-VP: beta(t) = beta_0 + t (beta_1 - beta_0)."""

    def __init__(self, beta_min=0.1, beta_max=20, N=1000, T=1):
        super().__init__(N)
        self.beta_0 = beta_min
        self.beta_1 = beta_max
        self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)
        self.alphas = 1.0 - self.discrete_betas
        self._T = T

    @property
    def T(self):
        return self._T

    def _beta(self, t):
        return self.beta_0 + t * (self.beta_1 - self.beta_0)

    def _log_mean_coeff(self, t):
        return -0.25 * t ** 2 * (self.beta_1 - self.beta_0) - 0.5 * t * self.beta_0


class VPSDE(_LinearBeta):
    """sde_lib.py:122-181."""

    def __init__(self, beta_min=0.1, beta_max=20, N=1000, T=1):
        super().__init__(beta_min, beta_max, N, T)
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.sqrt_alphas_cumprod = torch.sqrt(self.alphas_cumprod)
        self.sqrt_1m_alphas_cumprod = torch.sqrt(1.0 - self.alphas_cumprod)

    def sde(self, x, t):
        beta_t = self._beta(t)
        return -0.5 * beta_t[:, None] * x, torch.sqrt(beta_t)

    def marginal_prob(self, x, t):
        lmc = self._log_mean_coeff(t)
        return torch.exp(lmc[:, None]) * x, torch.sqrt(1.0 - torch.exp(2.0 * lmc))

    def return_alpha_sigma(self, t):
        lmc = self._log_mean_coeff(t)
        return torch.exp(lmc[:, None]), torch.sqrt(1.0 - torch.exp(2.0 * lmc))

    def discretize(self, x, t):
        """DDPM discretisation (sde_lib.py:167-175)."""
        timestep = (t * (self.N - 1) / self.T).long()
        beta = self.discrete_betas.to(x.device)[timestep]
        alpha = self.alphas.to(x.device)[timestep]
        return torch.sqrt(alpha)[:, None] * x - x, torch.sqrt(beta)


class subVPSDE(_LinearBeta):
    """sde_lib.py:184-231.  Note the reference's "std" is 1 - exp(2 lmc) (no square root) and is
    used consistently as a std; reproduced as is."""

    def sde(self, x, t):
        beta_t = self._beta(t)
        discount = 1.0 - torch.exp(-2 * self.beta_0 * t - (self.beta_1 - self.beta_0) * t ** 2)
        return -0.5 * beta_t[:, None] * x, torch.sqrt(beta_t * discount)

    def marginal_prob(self, x, t):
        lmc = self._log_mean_coeff(t)
        return torch.exp(lmc)[:, None] * x, 1 - torch.exp(2.0 * lmc)

    def return_alpha_sigma(self, t):
        lmc = self._log_mean_coeff(t)
        return torch.exp(lmc[:, None]), 1.0 - torch.exp(2.0 * lmc)


class VESDE(SDE):
    """sde_lib.py:234-292."""

    def __init__(self, sigma_min=0.01, sigma_max=50, N=1000, T=1):
        super().__init__(N)
        self.sigma_min = sigma_min
        self.sigma_max = sigma_max
        self.discrete_sigmas = torch.exp(torch.linspace(np.log(sigma_min), np.log(sigma_max), N))
        self._T = T

    @property
    def T(self):
        return self._T

    def _sigma(self, t):
        return self.sigma_min * (self.sigma_max / self.sigma_min) ** t

    def sde(self, x, t):
        g = self._sigma(t) * torch.sqrt(torch.tensor(2 * (np.log(self.sigma_max) - np.log(self.sigma_min)), device=t.device))
        return torch.zeros_like(x), g

    def marginal_prob(self, x, t):
        return x, self._sigma(t)

    def prior_sampling(self, shape):
        return torch.randn(*shape) * self.sigma_max

    def prior_logp(self, z):
        n = np.prod(z.shape[1:])
        return -n / 2.0 * np.log(2 * np.pi * self.sigma_max ** 2) - torch.sum(z ** 2, dim=1) / (2 * self.sigma_max ** 2)

    def discretize(self, x, t):
        """SMLD discretisation (sde_lib.py:278-287)."""
        timestep = (t * (self.N - 1) / self.T).long()
        sigma = self.discrete_sigmas.to(t.device)[timestep]
        adjacent = torch.where(timestep == 0, torch.zeros_like(t), self.discrete_sigmas[timestep - 1].to(t.device))
        return torch.zeros_like(x), torch.sqrt(sigma ** 2 - adjacent ** 2)

    def return_alpha_sigma(self, t):
        return torch.tensor([[1.0]]), self._sigma(t)


def sde_desc(sde, continuous=True):
    """(kind, N, beta_min, beta_max, T) for the C ABI; None when the fused kernels do not cover the SDE.  ``continuous=False`` selects the
    discrete score functions: VE (utils.py:175-178: the network conditioned on round((T - t)(N - 1))) and VP (utils.py:157-160: label
    t (N - 1), std from the DDPM table -- which the kernels rebuild for sde.N, so a VPSDE whose N was changed after construction, its table
    still the constructor's, returns None: step by step); it changes nothing for sub-VP."""
    from ... import _C
    if isinstance(sde, _ReverseSDE):
        sde = sde._fwd
    if isinstance(sde, subVPSDE):
        kind = _C.SDE_SUBVP
    elif isinstance(sde, VPSDE):
        kind = _C.SDE_VP
        if not continuous:
            if int(sde.sqrt_1m_alphas_cumprod.shape[0]) != int(sde.N):
                return None
            kind = _C.SDE_VP_DISCRETE
    elif isinstance(sde, VESDE):
        # the beta fields carry sigma_min / sigma_max (include/dposer_hip.h: DPOSER_SDE_VE / DPOSER_SDE_VE_DISCRETE)
        return _C.SdeDesc(_C.SDE_VE if continuous else _C.SDE_VE_DISCRETE, int(sde.N), float(sde.sigma_min), float(sde.sigma_max), float(sde.T))
    else:
        return None
    return _C.SdeDesc(kind, int(sde.N), float(sde.beta_0), float(sde.beta_1), float(sde.T))
