"""Loss, optimizer and the one-step train/eval function -- counterpart of the reference's
lib/algorithms/advanced/losses.py (get_optimizer :31-41, optimization_manager :44-58,
get_sde_loss_fn :61-137, get_smld_loss_fn :140-161, get_ddpm_loss_fn :164-184, get_step_fn :187-275).

Hot path on MI355X: ``step_fn`` for the shipped configuration (continuous sub-VP/VP DSM loss,
reduce_mean, no likelihood weighting, no auxiliary loss) is ONE fused pipeline --
``dposer_dsm_loss_fwd_bwd`` (perturbation, forward with dropout, loss, full backward into a flat
gradient) -> optional RCCL all-reduce of the flat gradient (dposer_amd.distributed) ->
``dposer_adam_ema_clip_step`` (global-norm clip + Adam + EMA in one pass).  Everything else
(likelihood weighting, VE, SMLD/DDPM legacy losses) composes the same differentiable HIP forward
with torch autograd.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.optim as optim

from ... import _C
from ..ema import ExponentialMovingAverage, flat_base
from . import utils as mutils
from .sde_lib import VESDE, VPSDE, sde_desc


class FusedAdam(optim.Adam):
    """torch.optim.Adam whose state lives in flat buffers and whose update is one HIP kernel.

    ``state_dict()`` / ``load_state_dict()`` keep torch's per-parameter layout
    ({'step', 'exp_avg', 'exp_avg_sq'}), so reference checkpoints (run/train.py:393-403) round-trip.
    Semantics follow torch.optim.Adam (amsgrad=False, maximize=False; weight_decay = the L2 form, grad += wd * param):
    parameters whose ``.grad`` is None are skipped."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self._flat_m = self._flat_v = self._flat_g = self._scratch = None
        self._flat_cache = self._state_views = None
        self._step_count = 0
        self._sharded = None
        self.world_size = 1

    def _params(self):
        return [p for g in self.param_groups for p in g["params"]]

    def _ensure_flat(self):
        params = self._params()
        # the layout check walks every parameter: cache it while the parameter views have not moved
        key = (len(params), params[0].data_ptr(), params[-1].data_ptr(), params[0].device) if params else None
        cached = getattr(self, "_flat_cache", None)
        if cached is not None and cached[0] == key and self._flat_m is not None:
            return cached[1], cached[2], params
        flat, offs = flat_base(params)
        if flat is None:
            raise _C.DPoserHipError("FusedAdam needs parameters that are views of one flat buffer (ScoreModelFC.flat_params())")
        _C.require_gpu(flat, "FusedAdam parameters")
        if self._flat_m is None or self._flat_m.numel() != flat.numel() or self._flat_m.device != flat.device:
            old = {p: self.state.get(p) for p in params}
            self._flat_m = torch.zeros_like(flat)
            self._flat_v = torch.zeros_like(flat)
            self._flat_g = torch.zeros_like(flat)
            self._scratch = torch.zeros(16384, dtype=torch.float32, device=flat.device)
            self._state_views = None
            for p, o in zip(params, offs):
                st = old.get(p)
                if st and "exp_avg" in st:          # state restored by load_state_dict: keep its values
                    self._flat_m[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
                    self._flat_v[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                    self._step_count = max(self._step_count, int(st["step"]))
        self._flat_cache = (key, flat, offs)
        return flat, offs, params

    def load_state_dict(self, state_dict):
        """torch's loader fills ``self.state[p]`` with fresh tensors; mirror them into the flat moment buffers and the step
        counter even when this optimizer has already stepped (in-process restore / rollback), and re-point the views."""
        super().load_state_dict(state_dict)
        params = self._params()
        steps = [int(self.state[p]["step"]) for p in params if p in self.state and "step" in self.state[p]]
        self._step_count = max(steps) if steps else 0
        if self._flat_m is not None:
            flat, offs = flat_base(params)
            if flat is not None and flat.numel() == self._flat_m.numel():
                self._flat_m.zero_()
                self._flat_v.zero_()
                for p, o in zip(params, offs):
                    st = self.state.get(p)
                    if st and "exp_avg" in st:
                        self._flat_m[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
                        self._flat_v[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
            else:
                self._flat_m = self._flat_v = None        # layout changed: rebuilt (from self.state) by _ensure_flat
        self._flat_cache = None
        self._state_views = None

    def nonfinite_steps(self) -> int:
        """How many optimisation steps the device dropped because the gradient contained NaN / Inf (the fused update checks the
        squared gradient norm and leaves parameters, moments and EMA untouched).  Reading it synchronises; call it at logging
        / checkpoint time, not per step."""
        return 0 if self._scratch is None else int(self._scratch[1].item())

    def flat_grad(self):
        """Flat gradient buffer the fused backward writes into (one element per flat parameter)."""
        self._ensure_flat()
        return self._flat_g

    @torch.no_grad()
    def fused_step(self, *, live, grad_clip=-1.0, grad_scale=1.0, ema: ExponentialMovingAverage = None, repack=None):
        """One optimizer update from ``self._flat_g``.  ``live[i]``: parameter i has a gradient.
        ``repack``: the model's ScoreEngine -- the update then also rewrites the engine's packed weight copies in the same pass
        (dposer_scorefc_adam_pack_step), and the next training step finds them current instead of re-packing 33 MB."""
        flat, offs, params = self._ensure_flat()
        skip = []
        for p, o, ok in zip(params, offs, live):
            if not ok:
                if skip and skip[-1][1] == o:
                    skip[-1][1] = o + p.numel()
                else:
                    skip.append([o, o + p.numel()])
        if len(skip) > 2:
            raise _C.DPoserHipError("FusedAdam: more than two disjoint parameter ranges without gradient")
        lo = (C.c_int64 * 2)(*([s[0] for s in skip] + [0, 0])[:2])
        hi = (C.c_int64 * 2)(*([s[1] for s in skip] + [0, 0])[:2])
        g = self.param_groups[0]
        self._step_count += 1
        ema_flat, omd = None, 0.0
        if ema is not None:
            ema_flat = ema.flat_shadow_for(flat)
            if ema_flat is None:
                raise _C.DPoserHipError("EMA shadow parameters are not flat-backed")
            omd = ema.next_one_minus_decay()
        target = None
        if repack is not None and os.environ.get("DPOSER_ADAM_REPACK", "1") != "0" and flat.numel() == repack.num_params:
            target = repack.repack_target(flat)
        if target is not None:
            _C.check(_C.lib().dposer_scorefc_adam_pack_step(repack.h, _C.ptr(flat), _C.ptr(self._flat_g), _C.ptr(self._flat_m),
                                                            _C.ptr(self._flat_v), _C.ptr(ema_flat), _C.ptr(target), lo, hi, len(skip),
                                                            float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                                            float(g.get("weight_decay", 0.0)), float(grad_clip), float(grad_scale),
                                                            self._step_count, float(omd), _C.ptr(self._scratch), 0, _C.stream_ptr()),
                     "dposer_scorefc_adam_pack_step")
        else:
            _C.check(_C.lib().dposer_adam_ema_clip_step_wd(_C.ptr(flat), _C.ptr(self._flat_g), _C.ptr(self._flat_m), _C.ptr(self._flat_v),
                                                           _C.ptr(ema_flat), flat.numel(), lo, hi, len(skip), float(g["lr"]),
                                                           float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                                           float(g.get("weight_decay", 0.0)), float(grad_clip), float(grad_scale),
                                                           self._step_count, float(omd), _C.ptr(self._scratch), 0, _C.stream_ptr()),
                     "dposer_adam_ema_clip_step_wd")
        torch.autograd.graph.increment_version(params)   # the kernel wrote the parameters through raw pointers: tell autograd
        if target is not None:                           # (after the version bump: the key describes the state the copies were made from)
            repack.mark_packed_by_optimizer(flat, params)
        self._publish_state(params, offs, live)

    def _publish_state(self, params, offs, live):
        """torch-compatible per-parameter state: {'step', 'exp_avg', 'exp_avg_sq'} are views of the flat buffers and ONE shared
        step tensor, rebuilt only when the set of live parameters (or the buffers) changes -- not 3 tensors per parameter per step."""
        live_key = (tuple(bool(ok) for ok in live), self._flat_m.data_ptr())
        if getattr(self, "_state_views", None) != live_key:
            self._step_t = torch.tensor(float(self._step_count))
            for p, o, ok in zip(params, offs, live):
                if ok:
                    st = self.state[p]
                    st["step"] = self._step_t
                    st["exp_avg"] = self._flat_m[o:o + p.numel()].view(p.shape)
                    st["exp_avg_sq"] = self._flat_v[o:o + p.numel()].view(p.shape)
            self._state_views = live_key
        self._step_t.fill_(float(self._step_count))

    @torch.no_grad()
    def gather_state(self):
        """After ZeRO-1 steps every rank holds the Adam moments of its OWN range only.  All-gather them (two collectives over the
        ranges of the last sharded step) and publish the torch-style per-parameter state, so that ``state_dict()`` of a sharded
        run is the state of the replicated run.  COLLECTIVE: every rank must call it, explicitly, before ``state_dict()``."""
        from ... import distributed as ddp
        sh = getattr(self, "_sharded", None)
        if sh is None:
            return
        bounds, live = sh
        flat, offs, params = self._ensure_flat()
        ddp.all_gather_flat_(self._flat_m, bounds)
        ddp.all_gather_flat_(self._flat_v, bounds)
        self._publish_state(params, offs, live)
        self._sharded = None

    def state_dict(self):
        """torch's layout.  After sharded (ZeRO-1) steps the moments live on their owning ranks: call ``gather_state()`` on EVERY
        rank first (a collective -- two all-gathers), then ``state_dict()`` wherever the checkpoint is written.  Calling it on
        ungathered state raises instead of entering a collective that the other ranks may never join (the reference's pattern is
        'rank 0 saves', run/train.py:393-403: an implicit collective there hangs RCCL / gloo without a diagnostic)."""
        if getattr(self, "_sharded", None) is not None:
            raise RuntimeError("FusedAdam.state_dict(): the Adam moments are sharded over the ranks (ZeRO-1 steps since the last gather). "
                               "Call optimizer.gather_state() on ALL ranks first (it all-gathers them), then state_dict() on the "
                               "rank that writes the checkpoint.")
        return super().state_dict()

    @torch.no_grad()
    def fused_step_sharded(self, *, live, bounds, rank, grad_clip=-1.0, grad_scale=1.0, ema: ExponentialMovingAverage = None):
        """ZeRO-1 style update: ``self._flat_g`` holds this rank's reduce-scattered gradient range ``bounds[rank]``;
        the global squared norm for the clip is one all-reduced float; Adam / EMA touch only the owned range (1/G of the
        work and of the moment traffic); the caller all-gathers the parameters afterwards.  Moments outside the owned range are
        stale on this rank until ``gather_state()`` all-gathers them (call it on EVERY rank before ``state_dict()``, which refuses
        sharded moments)."""
        from ... import distributed as ddp
        flat, offs, params = self._ensure_flat()
        lo, hi = bounds[rank]
        lib = _C.lib()
        g = self.param_groups[0]
        self._step_count += 1
        ema_flat, omd = None, 0.0
        if ema is not None:
            ema_flat = ema.flat_shadow_for(flat)
            if ema_flat is None:
                raise _C.DPoserHipError("EMA shadow parameters are not flat-backed")
            omd = ema.next_one_minus_decay()
        # squared norm of the owned range, excluding parameters without a gradient (their range holds zeros already)
        _C.check(lib.dposer_grad_sqnorm(_C.ptr(self._flat_g[lo:hi]), hi - lo, _C.ptr(self._scratch), _C.stream_ptr()), "dposer_grad_sqnorm")
        ddp.all_reduce_sum_(self._scratch[0:1])
        skip = []
        for p, o, ok in zip(params, offs, live):
            if not ok:
                a, b = max(o, lo) - lo, min(o + p.numel(), hi) - lo
                if b > a:
                    if skip and skip[-1][1] == a:
                        skip[-1][1] = b
                    else:
                        skip.append([a, b])
        if len(skip) > 2:
            raise _C.DPoserHipError("FusedAdam: more than two disjoint parameter ranges without gradient")
        slo = (C.c_int64 * 2)(*([s_[0] for s_ in skip] + [0, 0])[:2])
        shi = (C.c_int64 * 2)(*([s_[1] for s_ in skip] + [0, 0])[:2])
        if hi > lo:
            _C.check(lib.dposer_adam_ema_clip_step_wd(
                _C.ptr(flat[lo:hi]), _C.ptr(self._flat_g[lo:hi]), _C.ptr(self._flat_m[lo:hi]), _C.ptr(self._flat_v[lo:hi]),
                _C.ptr(None if ema_flat is None else ema_flat[lo:hi]), hi - lo, slo, shi, len(skip), float(g["lr"]), float(g["betas"][0]),
                float(g["betas"][1]), float(g["eps"]), float(g.get("weight_decay", 0.0)), float(grad_clip), float(grad_scale),
                self._step_count, float(omd), _C.ptr(self._scratch), 1, _C.stream_ptr()), "dposer_adam_ema_clip_step_wd")
        torch.autograd.graph.increment_version(params)
        self._sharded = (list(bounds), list(live))      # moments are complete on the owning ranks only: see gather_state()

    @torch.no_grad()
    def step(self, closure=None):
        """Generic entry (gradients in ``p.grad``, already clipped by the caller)."""
        loss = closure() if closure is not None else None
        flat, offs, params = self._ensure_flat()
        live = []
        for p, o in zip(params, offs):
            live.append(p.grad is not None)
            if p.grad is not None:
                self._flat_g[o:o + p.numel()].copy_(p.grad.reshape(-1))
        self.fused_step(live=live)
        return loss


def get_optimizer(config, params):
    """losses.py:31-41."""
    if config.optim.optimizer == "Adam":
        return FusedAdam(params, lr=config.optim.lr, betas=(config.optim.beta1, 0.999), eps=config.optim.eps,
                         weight_decay=config.optim.weight_decay)
    raise NotImplementedError(f"Optimizer {config.optim.optimizer} not supported yet!")


class OptimizeFn:
    """``optimize_fn(optimizer, params, step, ...)`` of losses.py:44-58: lr warm-up, global-norm clip, step."""

    def __init__(self, config):
        self.lr, self.warmup, self.grad_clip = config.optim.lr, config.optim.warmup, config.optim.grad_clip

    def warm_lr(self, optimizer, step, lr=None, warmup=None):
        lr = self.lr if lr is None else lr
        warmup = self.warmup if warmup is None else warmup
        if warmup > 0:
            for g in optimizer.param_groups:
                g["lr"] = lr * np.minimum(step / warmup, 1.0)

    def __call__(self, optimizer, params, step, lr=None, warmup=None, grad_clip=None):
        grad_clip = self.grad_clip if grad_clip is None else grad_clip
        self.warm_lr(optimizer, step, lr, warmup)
        if grad_clip >= 0:
            torch.nn.utils.clip_grad_norm_(params, max_norm=grad_clip)
        optimizer.step()


def optimization_manager(config):
    return OptimizeFn(config)


def get_sde_loss_fn(sde, train, reduce_mean=False, continuous=True, likelihood_weighting=False, eps=1e-5,
                    return_data=False, denoise_steps=5):
    """Denoising score-matching loss for an arbitrary SDE (losses.py:61-137), composed from the
    differentiable HIP forward and torch elementwise ops (generic path).

    ``return_data=True`` (the auxiliary-loss training of losses.py:91-119): the perturbed batch is denoised in ``denoise_steps``
    deterministic steps from t to t / (2 denoise_steps) -- every step one differentiable network evaluation whose activations
    stay leased until the backward pass -- and the loss function returns ``(loss, {'clean_sample', 'SNR', 't'})``; the score of
    the FIRST step carries the DSM term.  Keyword-only ``t=`` / ``z=`` inject the draws of losses.py:110-111 (tests)."""
    reduce_op = torch.mean if reduce_mean else (lambda *a, **k: 0.5 * torch.sum(*a, **k))

    def loss_fn(model, batch, condition, mask, *, t=None, z=None):
        score_fn = mutils.get_score_fn(sde, model, train=train, continuous=continuous)

        def multi_step_denoise(x_t, t0, t_end, N):                                  # losses.py:91-106
            from ...utils.misc import linear_interpolation
            time_traj = linear_interpolation(t0, t_end, N + 1)
            x_current, score_return = x_t, None
            for i in range(N):
                alpha_c, sigma_c = sde.return_alpha_sigma(time_traj[i])
                alpha_b, sigma_b = sde.return_alpha_sigma(time_traj[i + 1])
                score = score_fn(x_current, time_traj[i], condition, mask)
                if i == 0:
                    score_return = score
                noise = -score * sigma_c[:, None]                                   # score -> noise prediction
                x_current = alpha_b / alpha_c * (x_current - sigma_c[:, None] * noise) + sigma_b[:, None] * noise
            return score_return, x_current

        if t is None:
            t = torch.rand(batch.shape[0], device=batch.device) * (sde.T - eps) + eps
        if z is None:
            z = torch.randn_like(batch)
        mean, std = sde.marginal_prob(batch, t)
        perturbed = mean + std[:, None] * z
        if return_data:
            alpha, sigma = sde.return_alpha_sigma(t)
            SNR = alpha / sigma[:, None]
            score, estimated = multi_step_denoise(perturbed, t, t / (2 * denoise_steps), denoise_steps)
        else:
            score = score_fn(perturbed, t, condition, mask)
        if not likelihood_weighting:
            losses = reduce_op(torch.square(score * std[:, None] + z).reshape(batch.shape[0], -1), dim=-1)
        else:
            g2 = sde.sde(torch.zeros_like(batch), t)[1] ** 2
            losses = reduce_op(torch.square(score + z / std[:, None]).reshape(batch.shape[0], -1), dim=-1) * g2
        loss = torch.mean(losses)
        if return_data:
            return loss, {"clean_sample": estimated, "SNR": SNR, "t": t}
        return loss

    return loss_fn


def get_smld_loss_fn(vesde, train, reduce_mean=False):
    """Legacy SMLD loss (losses.py:140-161)."""
    assert isinstance(vesde, VESDE), "SMLD training only works for VESDEs."
    sigmas_desc = torch.flip(vesde.discrete_sigmas, dims=(0,))
    reduce_op = torch.mean if reduce_mean else (lambda *a, **k: 0.5 * torch.sum(*a, **k))

    def loss_fn(model, batch, condition, mask):
        model_fn = mutils.get_model_fn(model, train=train)
        labels = torch.randint(0, vesde.N, (batch.shape[0],), device=batch.device)
        sigmas = sigmas_desc.to(batch.device)[labels]
        noise = torch.randn_like(batch) * sigmas[:, None]
        score = model_fn(batch + noise, labels, condition, mask)
        target = -noise / (sigmas ** 2)[:, None]
        losses = reduce_op(torch.square(score - target).reshape(batch.shape[0], -1), dim=-1) * sigmas ** 2
        return torch.mean(losses)

    return loss_fn


def get_ddpm_loss_fn(vpsde, train, reduce_mean=True):
    """Legacy DDPM loss (losses.py:164-184)."""
    assert isinstance(vpsde, VPSDE), "DDPM training only works for VPSDEs."
    reduce_op = torch.mean if reduce_mean else (lambda *a, **k: 0.5 * torch.sum(*a, **k))

    def loss_fn(model, batch, condition, mask):
        model_fn = mutils.get_model_fn(model, train=train)
        labels = torch.randint(0, vpsde.N, (batch.shape[0],), device=batch.device)
        a = vpsde.sqrt_alphas_cumprod.to(batch.device)[labels, None]
        s = vpsde.sqrt_1m_alphas_cumprod.to(batch.device)[labels, None]
        noise = torch.randn_like(batch)
        score = model_fn(a * batch + s * noise, labels, condition, mask)
        losses = reduce_op(torch.square(score - noise).reshape(batch.shape[0], -1), dim=-1)
        return torch.mean(losses)

    return loss_fn


def fused_dsm_supported(sde, model, continuous, reduce_mean, likelihood_weighting, auxiliary_loss):
    from .model import ScoreModelFC
    return (continuous and reduce_mean and not likelihood_weighting and not auxiliary_loss and sde_desc(sde) is not None
            and isinstance(model, ScoreModelFC))


def fused_dsm_grad(model, sde, batch, *, flat_grad, t=None, z=None, eps=1e-5, seed=0, step=0, bucket_events=None, on_final=None):
    """dposer_dsm_loss_fwd_bwd[_bucketed | _notify]: loss (device scalar) and d loss/d params into ``flat_grad``.
    ``bucket_events`` (ScoreEngine.bucket_events()) are recorded as each gradient bucket becomes final; with ``on_final(ranges,
    event)`` the call itself announces every group of final buckets (one event + merged flat ranges) while it is still queueing
    the rest of the backward pass (distributed.StreamedAllReduce)."""
    _C.require_gpu(batch, "training batch")
    if batch.shape[0] == 0:
        # (the reference would take torch.mean of an empty tensor: a NaN loss and NaN gradients into Adam's moments -- refuse instead)
        raise ValueError("empty training batch")
    from ...engine import param_state_key
    eng = model._engine()
    flat = model.flat_params()
    # (a key equal to the one the fused optimizer step left behind: it wrote these copies itself, nothing to launch)
    packed = eng.packed(flat, with_backward=True, force=True, state_key=param_state_key(flat, model._param_list))
    B = batch.shape[0]
    ws = eng.workspace(B, _C.WS_TRAIN, 0, batch.device)
    loss = torch.empty(1, dtype=torch.float32, device=batch.device)
    desc = sde_desc(sde)
    x = batch.contiguous().float()
    if on_final is not None:
        cb_error = []

        def _cb(user, first_bucket, n_ranges, lo, hi, event):
            try:                                      # (an exception must not unwind through the C call: ctypes would swallow it)
                on_final([(lo[i], hi[i]) for i in range(n_ranges)], event)
            except BaseException as e:                # noqa: BLE001 -- re-raised below, after the call
                cb_error.append(e)
        cb = _C.RANGES_FINAL_FN(_cb)                  # (kept alive until the call has returned)
        _C.check(eng.lib.dposer_dsm_loss_fwd_bwd_notify(
            eng.h, _C.ptr(flat), _C.ptr(packed), _C.ptr(ws), C.byref(desc), _C.ptr(x), _C.ptr(t), _C.ptr(z), float(eps), int(seed),
            int(step) & 0xFFFFFFFF, _C.ptr(eng.freq(batch.device, model._fourier_W())), _C.ptr(model.sigmas), _C.ptr(flat_grad), _C.ptr(loss), B,
            bucket_events, len(bucket_events), cb, None, _C.stream_ptr()), "dposer_dsm_loss_fwd_bwd_notify")
        if cb_error:
            raise cb_error[0]
        return loss[0]
    _C.check(eng.lib.dposer_dsm_loss_fwd_bwd_bucketed(
        eng.h, _C.ptr(flat), _C.ptr(packed), _C.ptr(ws), C.byref(desc), _C.ptr(x), _C.ptr(t), _C.ptr(z), float(eps), int(seed),
        int(step) & 0xFFFFFFFF, _C.ptr(eng.freq(batch.device, model._fourier_W())), _C.ptr(model.sigmas), _C.ptr(flat_grad), _C.ptr(loss), B,
        bucket_events, 0 if bucket_events is None else len(bucket_events), _C.stream_ptr()), "dposer_dsm_loss_fwd_bwd_bucketed")
    return loss[0]


def _live_params(model):
    """Which parameters get a gradient (fixed per model unless requires_grad flags are flipped): cached on the model."""
    params = model._param_list
    rg = tuple(p.requires_grad for p in params)
    cache = getattr(model, "_live_cache", None)
    if cache is None or cache[0] != rg:
        cache = (rg, [not model._is_nograd(o) and r for r, o in zip(rg, model._offsets)])
        model._live_cache = cache
    return cache[1]


def _rot6d_to_axis_angle_autograd(rot6d):
    """Differentiable 6D -> axis-angle for the auxiliary loss with rot_rep='rot6d' (losses.py:247-249; the reference goes through
    torchgeometry: matrix -> quaternion -> angle-axis).  Plain torch ops on the device: the HIP conversion kernel
    (utils.transforms.rot6d_to_axis_angle) has no backward, and this path is neither the shipped configuration nor hot.
    Like the reference's route it is well-conditioned over the whole range: the angle comes from atan2(|skew| / 2, (trace - 1) / 2)
    (acos loses half its digits near 0 and pi), the axis from the skew part while sin(angle) is not small and, towards pi, from the
    diagonal of R (R + R^T = 2 cos(a) I + 2 (1 - cos a) n n^T), signed by the skew part; below 1e-4 rad the series of a / (2 sin a)."""
    a = rot6d.reshape(-1, 3, 2)
    b1 = torch.nn.functional.normalize(a[:, :, 0], dim=1)
    b2 = torch.nn.functional.normalize(a[:, :, 1] - (b1 * a[:, :, 1]).sum(dim=1, keepdim=True) * b1, dim=1)
    b3 = torch.cross(b1, b2, dim=1)
    R = torch.stack([b1, b2, b3], dim=-1)
    skew = torch.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], dim=1)       # 2 sin(a) n
    sin_a = 0.5 * torch.sqrt((skew * skew).sum(dim=1) + 1e-30)
    cos_a = 0.5 * ((R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]) - 1.0)
    angle = torch.atan2(sin_a, cos_a)
    # regular branch: angle / (2 sin angle) * skew, with the series for tiny angles
    small = angle < 1e-4
    scale = torch.where(small, 0.5 + angle * angle / 12.0, angle / (2.0 * torch.where(small, torch.ones_like(sin_a), sin_a).clamp_min(1e-12)))
    regular = skew * scale[:, None]
    # towards pi: |n_i| from the diagonal, signs from the largest component's row of the symmetric part and the skew part
    one_m_cos = (1.0 - cos_a).clamp_min(1e-12)
    diag = torch.stack([R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]], dim=1)
    n_abs = torch.sqrt(((diag - cos_a[:, None]) / one_m_cos[:, None]).clamp_min(0.0) + 1e-30)
    sym = 0.5 * (R + R.transpose(1, 2))                                                            # cos I + (1 - cos) n n^T
    k = torch.argmax(n_abs, dim=1)
    row = sym[torch.arange(R.shape[0], device=R.device), k]                                        # (1 - cos) n_k n  (+ cos e_k)
    row = row - cos_a[:, None] * torch.nn.functional.one_hot(k, 3).to(R.dtype)
    n = row / (one_m_cos * n_abs.gather(1, k[:, None]).squeeze(1)).clamp_min(1e-12)[:, None]
    sgn = torch.sign((n * skew).sum(dim=1))
    sgn = torch.where(sgn == 0, torch.ones_like(sgn), sgn)                                          # exactly pi: either sign is the same rotation
    near_pi = n * (sgn * angle)[:, None]
    return torch.where((cos_a < -0.99)[:, None], near_pi, regular)


_RANK_GENERATORS = {}


def _dp_draws(sde, batch, eps=1e-5):
    """Data-parallel runs of the autograd steps (auxiliary loss, generic fallback): t and z of losses.py:110-111 from a generator
    keyed per RANK -- identically seeded ranks must not perturb sample i of every shard with the same (t, z) (the fused step
    keys its Philox streams the same way)."""
    from ... import distributed as ddp
    key = (batch.device.type, batch.device.index, ddp.rank())
    seed0 = torch.initial_seed()
    ent = _RANK_GENERATORS.get(key)
    if ent is None or ent[0] != seed0:                # a later torch.manual_seed() (new experiment, resume, tests) re-keys the stream
        gen = torch.Generator(device=batch.device)
        gen.manual_seed((seed0 + 0x9E3779B1 * (ddp.rank() + 1)) & 0x7FFFFFFFFFFFFFFF)
        ent = _RANK_GENERATORS[key] = (seed0, gen)
    gen = ent[1]
    t = torch.rand(batch.shape[0], device=batch.device, generator=gen) * (sde.T - eps) + eps
    z = torch.randn(batch.shape, device=batch.device, dtype=batch.dtype, generator=gen)
    return t, z


@torch.no_grad()
def _dp_average_grads(params):
    """All-reduce (mean) of the ``.grad`` tensors of an autograd step under data parallelism: ONE collective over a flat copy
    (the reference runs these steps under nn.DataParallel, whose gradient covers the global batch).  Every rank must hold a
    gradient for the same parameters."""
    from ... import distributed as ddp
    live = [p for p in params if p.grad is not None]
    if not live:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in live])
    world = ddp.all_reduce_sum_(flat)
    flat.mul_(1.0 / world)
    o = 0
    for p in live:
        n = p.numel()
        p.grad.copy_(flat[o:o + n].view_as(p.grad))
        o += n


def _dp_model_seed(model):
    """(context) dropout of the differentiable HIP forward keyed per rank for the duration of an autograd step."""
    from ... import distributed as ddp

    class _Ctx:
        def __enter__(self_):
            self_.keep = getattr(model, "_rng_seed", None)
            if self_.keep is not None and ddp.dp_active():
                model._rng_seed = (self_.keep + 0x9E3779B1 * ddp.rank()) & 0xFFFFFFFFFFFFFFFF

        def __exit__(self_, *exc):
            if self_.keep is not None:
                model._rng_seed = self_.keep
            return False
    return _Ctx()


def get_step_fn(sde, train, optimize_fn=None, reduce_mean=False, continuous=True, likelihood_weighting=False,
                auxiliary_loss=False, denormalize=None, body_model=None, rot_rep="rot6d", denoise_steps=5):
    """One-step training / evaluation function (losses.py:187-275).

    ``step_fn(state, batch, condition=None, mask=None)`` with ``state = {model, optimizer, ema, step}``
    returns ``{'step_loss', 'score_loss'}`` exactly like the reference.  Extra keyword-only knobs for
    tests: ``t=``/``z=`` inject the random draws of losses.py:110-111."""
    if continuous:
        loss_fn = get_sde_loss_fn(sde, train, reduce_mean=reduce_mean, continuous=True, likelihood_weighting=likelihood_weighting,
                                  return_data=auxiliary_loss, denoise_steps=denoise_steps)
    else:
        assert not likelihood_weighting, "Likelihood weighting is not supported for original SMLD/DDPM training."
        if isinstance(sde, VESDE):
            loss_fn = get_smld_loss_fn(sde, train, reduce_mean=reduce_mean)
        elif isinstance(sde, VPSDE):
            loss_fn = get_ddpm_loss_fn(sde, train, reduce_mean=reduce_mean)
        else:
            raise ValueError(f"Discrete training for {sde.__class__.__name__} is not recommended.")

    if auxiliary_loss:
        assert denormalize is not None and body_model is not None                     # losses.py:213-214

    def aux_step(state, batch, condition, mask, t, z):
        """losses.py:242-258: DSM term + SNR-weighted vertex / joint errors between the body posed by the multi-step estimate
        and by the batch.  Score network and body model both run their HIP forward / backward through autograd: this is the one
        training step where the two meet (the reference calls the body model 'the bottleneck of training', :252)."""
        from ... import distributed as ddp
        model, optimizer = state["model"], state["optimizer"]
        optimizer.zero_grad()
        dp = ddp.dp_active()
        if dp and (t is None or z is None):
            t_r, z_r = _dp_draws(sde, batch)
            t, z = (t_r if t is None else t), (z_r if z is None else z)
        with _dp_model_seed(model):
            score_loss, data = loss_fn(model, batch, condition, mask, t=t, z=z)
        weight = torch.log(1.0 + data["SNR"])                                          # [b, 1]
        estimate, target = denormalize(data["clean_sample"]), denormalize(batch)
        if rot_rep == "rot6d":
            n_poses = batch.shape[1] // 6
            estimate = _rot6d_to_axis_angle_autograd(estimate.reshape(-1, 6)).reshape(-1, n_poses * 3)
            target = _rot6d_to_axis_angle_autograd(target.reshape(-1, 6)).reshape(-1, n_poses * 3)
        gt_body = body_model(pose_body=target)
        pred_body = body_model(pose_body=estimate)
        loss_v2v = torch.mean(weight * torch.square(gt_body.v - pred_body.v).sum(dim=-1))
        loss_j2j = torch.mean(weight * torch.square(gt_body.Jtr - pred_body.Jtr).sum(dim=-1))
        loss = score_loss + loss_v2v + loss_j2j
        loss.backward()
        if dp:                                                                     # gradient of the GLOBAL batch, then clip + step
            _dp_average_grads(list(model.parameters()))
        optimize_fn(optimizer, model.parameters(), step=state["step"])
        state["step"] += 1
        state["ema"].update(model.parameters())
        return {"step_loss": loss, "score_loss": score_loss, "v2v_loss": loss_v2v, "j2j_loss": loss_j2j}

    def step_fn(state, batch, condition=None, mask=None, *, t=None, z=None):
        model = state["model"]
        if train and auxiliary_loss:
            return aux_step(state, batch, condition, mask, t, z)
        if not train:
            with torch.no_grad():                                                  # losses.py:264-271
                ema = state["ema"]
                ema.store(model.parameters())
                ema.copy_to(model.parameters())
                loss = loss_fn(model, batch, condition, mask)
                if isinstance(loss, tuple):                                        # (auxiliary-loss loss_fn: the reference's eval step
                    loss = loss[0]                                                 #  would put the tuple into the dict, :264-271)
                ema.restore(model.parameters())
            return {"step_loss": loss, "score_loss": loss}
        optimizer = state["optimizer"]
        fused = (isinstance(optimizer, FusedAdam) and isinstance(optimize_fn, OptimizeFn)
                 and fused_dsm_supported(sde, model, continuous, reduce_mean, likelihood_weighting, auxiliary_loss))
        if fused:
            from ... import distributed as ddp
            if not model.training:
                model.train()
            flat_grad = optimizer.flat_grad()
            # Philox key of this rank's shard: sample i of every shard must NOT draw the same t / z / dropout mask
            seed = (model._rng_seed + 0x9E3779B1 * ddp.rank()) & 0xFFFFFFFFFFFFFFFF if ddp.dp_active() else model._rng_seed
            zero1 = ddp.dp_active() and (os.environ.get("DPOSER_ZERO1") == "1" or bool(getattr(optimizer, "zero1", False)))
            if zero1:
                # ZeRO-1 style step: reduce-scatter the gradient, update only the owned 1/G range of parameters,
                # moments and EMA, all-gather the parameters (and the EMA shadow, which every rank keeps whole for evaluation)
                world, rk = ddp.world_size(), ddp.rank()
                bounds = ddp.zero1_bounds(flat_grad.numel(), world)
                if os.environ.get("DPOSER_DP_NOTIFY", "1") != "0":
                    # the reduce-scatter bucket by bucket UNDER the backward pass: every announced group of final ranges is cut at the
                    # ownership boundaries and each piece reduced to its owner from the communication stream
                    eng = model._engine()
                    red = ddp.StreamedReduceToOwners(flat_grad, lambda stream, ev: _C.check(eng.lib.dposer_stream_wait_event(stream, ev),
                                                                                            "dposer_stream_wait_event"), bounds)
                    loss = fused_dsm_grad(model, sde, batch, flat_grad=flat_grad, t=t, z=z, seed=seed, step=state["step"],
                                          bucket_events=eng.bucket_events(), on_final=red.on_final)
                    red.finish(expect=eng.grad_buckets)
                else:                                                               # (A/B: one collective after the call)
                    loss = fused_dsm_grad(model, sde, batch, flat_grad=flat_grad, t=t, z=z, seed=seed, step=state["step"])
                    ddp.reduce_scatter_flat_(flat_grad, bounds)
                optimize_fn.warm_lr(optimizer, state["step"])
                live = _live_params(model)
                optimizer.fused_step_sharded(live=live, bounds=bounds, rank=rk, grad_clip=optimize_fn.grad_clip, grad_scale=1.0 / world,
                                             ema=state["ema"])
                ddp.all_gather_flat_(model.flat_params(), bounds)
                shadow = state["ema"].flat_shadow_for(model.flat_params()) if state["ema"] is not None else None
                if shadow is not None:
                    ddp.all_gather_flat_(shadow, bounds)
                state["step"] += 1
                return {"step_loss": loss, "score_loss": loss}
            if ddp.dp_active():
                # bucketed: each GN layer's gradient is all-reduced (RCCL over xGMI) on a side stream while the layers in
                # front of it are still being differentiated
                eng = model._engine()
                events = eng.bucket_events()
                if os.environ.get("DPOSER_DP_NOTIFY", "1") != "0":
                    # the backward pass announces final bucket groups from inside the call: collectives are issued at once
                    red = ddp.StreamedAllReduce(flat_grad, lambda stream, ev: _C.check(eng.lib.dposer_stream_wait_event(stream, ev),
                                                                                       "dposer_stream_wait_event"))
                    loss = fused_dsm_grad(model, sde, batch, flat_grad=flat_grad, t=t, z=z, seed=seed, step=state["step"],
                                          bucket_events=events, on_final=red.on_final)
                    world = red.finish(expect=eng.grad_buckets)
                else:                                                               # (A/B: all waits + collectives after the call)
                    loss = fused_dsm_grad(model, sde, batch, flat_grad=flat_grad, t=t, z=z, seed=seed, step=state["step"],
                                          bucket_events=events)
                    world = ddp.all_reduce_buckets_(flat_grad, eng.grad_buckets,
                                                    lambda i, stream: _C.check(eng.lib.dposer_stream_wait_event(stream, events[i]),
                                                                               "dposer_stream_wait_event"))
            else:
                # (DPOSER_FORCE_BUCKET_EVENTS=1: single-GPU A/B of the bucketed backward schedule -- the events are recorded, nobody waits)
                events = model._engine().bucket_events() if os.environ.get("DPOSER_FORCE_BUCKET_EVENTS") == "1" else None
                loss = fused_dsm_grad(model, sde, batch, flat_grad=flat_grad, t=t, z=z, seed=seed, step=state["step"], bucket_events=events)
                world = 1
            optimize_fn.warm_lr(optimizer, state["step"])                           # losses.py:51-53
            live = _live_params(model)
            optimizer.fused_step(live=live, grad_clip=optimize_fn.grad_clip, grad_scale=1.0 / world, ema=state["ema"], repack=model._engine())
            state["step"] += 1
            return {"step_loss": loss, "score_loss": loss}
        from ... import distributed as ddp
        optimizer.zero_grad()
        dp = ddp.dp_active()
        if dp and not continuous:
            # (SMLD / DDPM draw labels and noise inside their loss functions from the global generator: identically seeded ranks
            #  would train on perfectly correlated perturbations)
            raise NotImplementedError("data-parallel training of the legacy SMLD / DDPM losses is not supported: use continuous=True")
        with _dp_model_seed(model):
            if dp:
                t_r, z_r = _dp_draws(sde, batch)
                loss = loss_fn(model, batch, condition, mask, t=t_r if t is None else t, z=z_r if z is None else z)
            elif continuous and (t is not None or z is not None):
                loss = loss_fn(model, batch, condition, mask, t=t, z=z)
            else:
                loss = loss_fn(model, batch, condition, mask)
        loss.backward()
        if dp:
            _dp_average_grads(list(model.parameters()))
        optimize_fn(optimizer, model.parameters(), step=state["step"])
        state["step"] += 1
        state["ema"].update(model.parameters())
        return {"step_loss": loss, "score_loss": loss}

    return step_fn
