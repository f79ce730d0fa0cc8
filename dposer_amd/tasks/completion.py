"""Pose completion by optimisation under the DPoser prior -- counterpart of ``DPoserComp`` in the
reference's run/completion.py:95-207.

The whole optimisation loop is ONE call, ``dposer_completion_optimize``: the time-bias table of all 200 steps is built once,
the weights are packed once, and each step is perturb -> forward-only score network at the step's shared t -> one kernel for
the Tweedie estimate, the gradients of the weighted prior loss and of the masked MSE data term, and torch.optim.Adam's update
with per-sample moments.  The optimiser state is per sample, so the loop shards over GPUs with no collective
(``distributed.shard_bounds`` = the reference's DistributedEvalSampler arithmetic).  SDEs / models outside the fused path
(VE, Fourier embedding) run the same loop step by step through ``prior_loss`` and torch's Adam.
"""
import ctypes as C
import math

import numpy as np
import torch
from torch import nn

from .. import _C
from ..algorithms.advanced import sde_lib
from ..prior import prior_loss


class DPoserComp:
    def __init__(self, diffusion_model, sde, continuous, batch_size=1):
        self.batch_size = batch_size
        self.sde = sde
        self.model = diffusion_model
        self.continuous = continuous
        self.data_loss = nn.MSELoss(reduction="mean")
        self._calls = 0

    def loss(self, x_0, t, weighted=False, z=None):
        """completion.py:131-149 at one shared time ``t`` (python float): mean(weight * (x_0 - x0_hat)^2)."""
        self._calls += 1
        return prior_loss(self.model, self.sde, x_0, t, weighted=bool(weighted), reduction="mean", z=z,
                          seed=self.model._rng_seed + 29, step=self._calls, continuous=bool(getattr(self, "continuous", True)))

    def get_loss_weights(self):
        """completion.py:151-155."""
        return {"data": lambda cst, it: 100 * cst / (1 + it), "dposer": lambda cst, it: 0.1 * cst * (it + 1)}

    @staticmethod
    def quan_t(step, total_steps, N, sample_trun=5.0):
        """time strategy '3' (completion.py:189-190); the reference multiplies an int64 tensor by a python float,
        i.e. in fp32."""
        return int(N - math.floor(float(np.float32(total_steps - step - 1) * np.float32(N / (sample_trun * total_steps)))) - 2)

    def _fused_supported(self):
        from ..algorithms.advanced.model import ScoreModelFC
        # (sub-VP: one score function; VE / VP: continuous or discrete, all on the fused kernels)
        return sde_lib.sde_desc(self.sde, bool(getattr(self, "continuous", True))) is not None and isinstance(self.model, ScoreModelFC)

    def _schedule(self, time_strategy, total_steps, sample_trun, sample_time):
        """quan_t of every step (completion.py:183-192)."""
        if time_strategy == "1":
            return [int(torch.randint(self.sde.N, [1])) for _ in range(total_steps)]
        if time_strategy == "2":
            return [int(sample_time)] * total_steps
        if time_strategy == "3":
            return [self.quan_t(step, total_steps, self.sde.N, sample_trun) for step in range(total_steps)]
        raise NotImplementedError("unsupported time sampling strategy")

    def optimize(self, observation, mask, time_strategy="3", lr=0.1, sample_trun=5.0, sample_time=900, iterations=2,
                 steps_per_iter=100, noise=None):
        """completion.py:167-207.  ``noise`` [total_steps, B, D]: injected z of the prior loss (tests)."""
        total_steps = iterations * steps_per_iter
        weights = self.get_loss_weights()
        timesteps = torch.linspace(self.sde.T, 1e-3, self.sde.N)          # host copy: t enters the kernels as a scalar
        quan = self._schedule(time_strategy, total_steps, sample_trun, sample_time)
        if self._fused_supported():
            _C.require_gpu(observation, "completion observation")
            model = self.model
            eng = model._engine()
            flat = model.flat_params()
            packed = eng.packed(flat, with_backward=False, force=not model.freeze_packed)      # once per loop
            B, D = observation.shape
            ws = eng.workspace(B, _C.WS_SHARED_T, total_steps, observation.device)
            obs = observation.detach().contiguous().float()
            msk = mask.detach().contiguous().float()
            x = obs.clone()
            m, v = torch.zeros_like(x), torch.zeros_like(x)
            its = [s // steps_per_iter for s in range(total_steps)]
            t_host = (C.c_float * total_steps)(*[float(timesteps[q]) for q in quan])
            # the reference passes quan_t positionally into `weighted` (completion.py:196): weighted = bool(quan_t)
            wflag = (C.c_int32 * total_steps)(*[1 if q else 0 for q in quan])
            w_prior = (C.c_float * total_steps)(*[float(weights["dposer"](1.0, it)) for it in its])
            w_data = (C.c_float * total_steps)(*[float(weights["data"](1.0, it)) for it in its])
            nz = None if noise is None else noise.detach().contiguous().float()
            step0 = self._calls + 1
            self._calls += total_steps
            _C.check(eng.lib.dposer_completion_optimize(
                eng.h, _C.ptr(flat), _C.ptr(packed), _C.ptr(ws), C.byref(sde_lib.sde_desc(self.sde, bool(getattr(self, "continuous", True)))), _C.ptr(x), _C.ptr(obs), _C.ptr(msk),
                _C.ptr(m), _C.ptr(v), t_host, wflag, w_prior, w_data, total_steps, float(lr), 0.9, 0.999, 1e-8, _C.ptr(nz),
                int(model._rng_seed + 29), int(step0) & 0xFFFFFFFF, _C.ptr(eng.freq(x.device, model._fourier_W())), _C.ptr(model.sigmas), B, _C.stream_ptr()),
                "dposer_completion_optimize")
            return observation * mask + x * (1.0 - mask)
        x = observation.clone().detach().requires_grad_(True)
        optimizer = torch.optim.Adam([x], lr, betas=(0.9, 0.999))
        for step, q in enumerate(quan):
            it = step // steps_per_iter
            optimizer.zero_grad()
            l_prior = self.loss(x, float(timesteps[q]), weighted=bool(q), z=None if noise is None else noise[step])
            l_data = self.data_loss(x * mask, observation * mask)
            tot = weights["dposer"](l_prior, it) + weights["data"](l_data, it)
            tot.backward()
            optimizer.step()
        return observation * mask + x.detach() * (1.0 - mask)


def evaluate_completion(model, sde, normalizer, body_model, poses, *, part="legs", hypo=1, batch_size=16384, continuous=True,
                        num_replicas=None, rank=None, optimize_kwargs=None):
    """The evaluation loop of run/completion.py:215-323 (``inference``) for the poses of one test set, without its process / CLI
    shell: this rank's contiguous shard (DistributedEvalSampler arithmetic, EvaSampler.py:78-106) in batches of ``batch_size``
    with ``drop_last`` like the reference's DataLoader -> create_mask -> ``hypo`` completions -> de-normalise -> Evaler
    (min over hypotheses) -> mean of every metric over all samples of all ranks.

    Nothing synchronises with the host inside the loop: the per-sample metrics stay device tensors, and the end is ONE all-reduce
    of (sum, count) pairs (``distributed.reduce_metric_means``) instead of ``gather_object`` of every value (:300-305).
    ``poses`` [N, D] normalised test poses (the dataset's ``poses`` tensor); returns ``({metric: mean}, samples evaluated here)``."""
    from .. import distributed as ddp
    from ..dataset.AMASS import Evaler
    from ..utils.misc import create_mask
    world = ddp.world_size() if num_replicas is None else num_replicas
    rk = ddp.rank() if rank is None else rank
    lo, hi = ddp.shard_bounds(poses.shape[0], world, rk)
    dev = next(model.parameters()).device
    evaler = Evaler(body_model=body_model, part=part)
    comp = DPoserComp(model, sde, continuous, batch_size=batch_size)      # one object: its call counter keys the in-kernel noise
    results, done = [], 0
    for b0 in range(lo, hi - batch_size + 1, batch_size) if hi - lo >= batch_size else []:
        batch = poses[b0:b0 + batch_size].to(dev, non_blocking=True)
        mask, observation = create_mask(batch, part=part)
        outs = torch.stack([comp.optimize(observation, mask, **(optimize_kwargs or {})) for _ in range(hypo)], dim=1)
        preds = normalizer.offline_denormalize(outs, to_axis=True)
        gts = normalizer.offline_denormalize(batch, to_axis=True)
        results.append(evaler.multi_eval_bodys(preds, gts, as_tensors=True))
        done += batch_size
    # fixed name list: a rank whose shard is shorter than one batch (drop_last) has no results to learn the names from
    return ddp.reduce_metric_means(results, device=dev, names=("mpjpe_body", "mpvpe_all")), done
