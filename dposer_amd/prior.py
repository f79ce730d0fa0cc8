"""The DPoser pose prior as an nn.Module -- counterpart of the reference's ``DPoser`` in
run/smplify.py:17-115 and of the shared ``one_step_denoise`` / ``loss`` maths of
run/completion.py:105-149 and run/motion_denoising.py:99-143.

One prior evaluation = ``dposer_prior_loss``: perturb x_0 at a shared time t, one forward-only
network evaluation (x0_hat is detached in the reference, so no gradient ever flows through the
network), Tweedie estimate, weighted L2 -- with the analytic gradient 2 w (x_0 - x0_hat)/n returned
to autograd.
"""
import ctypes as C

import torch
from torch import nn

from . import _C
from .algorithms.advanced import sde_lib
from .algorithms.advanced.model import ScoreModelFC
from .algorithms.ema import ExponentialMovingAverage
from .dataset.AMASS import N_POSES, Posenormalizer


class _PriorLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, model, sde, t, weighted, inv_n, z, seed, step, continuous=True):
        _C.require_gpu(x0, "prior-loss input")
        eng = model._engine()
        flat = model.flat_params()
        packed = eng.packed(flat, with_backward=False, force=not model.freeze_packed)
        B = x0.shape[0]
        ws = eng.workspace(B, _C.WS_SHARED_T, 1, x0.device)
        x = x0.detach().contiguous().float()
        grad = torch.empty_like(x)
        x0_hat = torch.empty_like(x)
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        desc = sde_lib.sde_desc(sde, continuous)
        zz = None if z is None else z.contiguous().float()
        _C.check(eng.lib.dposer_prior_loss(eng.h, _C.ptr(flat), _C.ptr(packed), _C.ptr(ws), C.byref(desc), _C.ptr(x), _C.ptr(zz), float(t),
                                           1 if weighted else 0, float(inv_n), _C.ptr(x0_hat), _C.ptr(grad), _C.ptr(loss), int(seed),
                                           int(step) & 0xFFFFFFFF, _C.ptr(eng.freq(x.device, model._fourier_W())), _C.ptr(model.sigmas), B,
                                           _C.stream_ptr()), "dposer_prior_loss")
        ctx.save_for_backward(grad)
        ctx.x0_hat = x0_hat
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return (grad * g, None, None, None, None, None, None, None, None, None)


def _prior_loss_unfused(model, sde, x0, t, weighted, inv_n, z, continuous=True):
    """smplify.py:93-107 / completion.py:131-149 step by step for an SDE the fused kernel does not cover (VE).  The one-step estimate is
    detached in the reference (smplify.py:73), so the gradient reaches x0 through the explicit (x0 - x0_hat) only."""
    from .algorithms.advanced import utils as mutils
    B = x0.shape[0]
    with torch.no_grad():
        vec_t = torch.full((B,), t, device=x0.device, dtype=torch.float32)
        zz = torch.randn_like(x0) if z is None else z.to(x0.device, torch.float32)
        mean, std = sde.marginal_prob(x0.detach().float(), vec_t)
        x_t = mean + std[:, None] * zz
        score = mutils.get_score_fn(sde, model, train=False, continuous=continuous)(x_t, vec_t, condition=None, mask=None)
        alpha, sigma = sde.return_alpha_sigma(vec_t)
        alpha, sigma = alpha.to(x0.device), sigma.to(x0.device)
        x0_hat = (x_t + (sigma ** 2)[:, None] * score) / alpha              # alpha: [1, 1] (VE) or [B, 1]
        snr = alpha / sigma[:, None]
        w = 0.5 * torch.sqrt(1 + snr) if weighted else torch.full_like(snr, 0.5)
    return (w * (x0.float() - x0_hat) ** 2).sum() * inv_n


def prior_loss(model, sde, x0, t, *, weighted=True, reduction="mean", batch_size=None, z=None, seed=0, step=0, continuous=True):
    """Weighted denoising loss at one shared time ``t`` (python float).
    reduction='mean' -> torch.mean over [B, D] (completion.py:147); 'sum_over_batch' -> sum / batch_size (smplify.py:105).
    ``continuous``: ``config.training.continuous`` as the reference hands it to ``get_score_fn`` (motion_denoising.py:94,
    completion.py:103).  Under the VE SDE it selects the label the network is conditioned on (utils.py:164-181: sigma(t), or
    round((T - t)(N - 1)) for a discrete model), under the VP SDE label and std of the score (utils.py:152-160) -- all on the fused kernel
    since round 6 (DPOSER_SDE_VE_DISCRETE / DPOSER_SDE_VP_DISCRETE)."""
    if x0.shape[0] == 0:
        raise ValueError("prior_loss: empty batch (the reference's torch.mean over no elements is NaN)")
    n = x0.numel() if reduction == "mean" else (batch_size if batch_size is not None else x0.shape[0])
    if sde_lib.sde_desc(sde, bool(continuous)) is None:   # not covered by the fused kernel: the HIP score function + the reference's few elementwise steps
        return _prior_loss_unfused(model, sde, x0, float(t), bool(weighted), 1.0 / float(n), z, continuous=bool(continuous))
    return _PriorLoss.apply(x0, model, sde, float(t), bool(weighted), 1.0 / float(n), z, seed, step, bool(continuous))


class DPoser(nn.Module):
    """run/smplify.py:17-115.  ``forward(poses, betas, quan_t)`` returns the prior loss."""

    def __init__(self, batch_size=32, config_path="", args=None, model=None, normalizer=None):
        super().__init__()
        from .utils.generic import import_configs
        self.device = args.device
        self.batch_size = batch_size
        config = import_configs(config_path)
        self.Normalizer = normalizer if normalizer is not None else Posenormalizer(
            data_path=f"{args.dataset_folder}/{args.version}/train", min_max=config.data.min_max,
            rot_rep=config.data.rot_rep, device=args.device)
        diffusion_model = model if model is not None else self.load_model(config, args)
        name = config.training.sde.lower()
        if name == "vpsde":
            sde = sde_lib.VPSDE(beta_min=config.model.beta_min, beta_max=config.model.beta_max, N=config.model.num_scales)
        elif name == "subvpsde":
            sde = sde_lib.subVPSDE(beta_min=config.model.beta_min, beta_max=config.model.beta_max, N=config.model.num_scales)
        elif name == "vesde":
            sde = sde_lib.VESDE(sigma_min=config.model.sigma_min, sigma_max=config.model.sigma_max, N=config.model.num_scales)
        else:
            raise NotImplementedError(f"SDE {config.training.sde} unknown.")
        sde.N = args.sde_N
        self.sde = sde
        self.continuous = bool(getattr(config.training, "continuous", True))      # smplify.py:44: the score function's flavour
        self.model = diffusion_model
        self.model.eval()
        self.model.freeze_packed = False
        self.timesteps = torch.linspace(self.sde.T, 1e-3, self.sde.N)       # host copy: t is a launch scalar
        self._calls = 0

    def load_model(self, config, args):
        pose_dim = 3 if config.data.rot_rep == "axis" else 6
        model = ScoreModelFC(config, n_poses=N_POSES, pose_dim=pose_dim, hidden_dim=config.model.HIDDEN_DIM,
                             embed_dim=config.model.EMBED_DIM, n_blocks=config.model.N_BLOCKS)
        model.to(self.device)
        model.eval()
        ckpt = torch.load(args.ckpt_path, map_location={"cuda:0": self.device})
        ema = ExponentialMovingAverage(model.parameters(), decay=config.model.ema_rate)
        model.load_state_dict(ckpt["model_state_dict"])
        ema.load_state_dict(ckpt["ema"])       # loaded but never copied into the model, as in smplify.py:62-67
        return model

    def DPoser_loss(self, x_0, t, z=None):
        self._calls += 1
        return prior_loss(self.model, self.sde, x_0, t, weighted=True, reduction="sum_over_batch", batch_size=self.batch_size,
                          z=z, seed=self.model._rng_seed + 17, step=self._calls, continuous=getattr(self, "continuous", True))

    def forward(self, poses, betas, quan_t, z=None):
        poses = self.Normalizer.offline_normalize(poses[:, :N_POSES * 3], from_axis=True)
        return self.DPoser_loss(poses, float(self.timesteps[int(quan_t)]), z=z)
