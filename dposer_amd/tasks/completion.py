"""Pose completion by optimisation under the DPoser prior -- counterpart of ``DPoserComp`` in the
reference's run/completion.py:95-207.

Each of the 200 Adam steps is one ``dposer_prior_loss`` call (perturb -> forward-only score network at a
shared t -> Tweedie estimate -> weighted L2 with the analytic gradient) plus a masked MSE data term; the
optimiser state is per sample, so the loop shards over GPUs with no collective
(``distributed.shard_bounds`` = the reference's DistributedEvalSampler arithmetic).
"""
import math

import numpy as np
import torch
from torch import nn

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
                          seed=self.model._rng_seed + 29, step=self._calls)

    def get_loss_weights(self):
        """completion.py:151-155."""
        return {"data": lambda cst, it: 100 * cst / (1 + it), "dposer": lambda cst, it: 0.1 * cst * (it + 1)}

    @staticmethod
    def quan_t(step, total_steps, N, sample_trun=5.0):
        """time strategy '3' (completion.py:189-190); the reference multiplies an int64 tensor by a python float,
        i.e. in fp32."""
        return int(N - math.floor(float(np.float32(total_steps - step - 1) * np.float32(N / (sample_trun * total_steps)))) - 2)

    def optimize(self, observation, mask, time_strategy="3", lr=0.1, sample_trun=5.0, sample_time=900, iterations=2,
                 steps_per_iter=100, noise=None):
        """completion.py:167-207.  ``noise`` [total_steps, B, D]: injected z of the prior loss (tests)."""
        total_steps = iterations * steps_per_iter
        x = observation.clone().detach().requires_grad_(True)
        optimizer = torch.optim.Adam([x], lr, betas=(0.9, 0.999))
        weights = self.get_loss_weights()
        timesteps = torch.linspace(self.sde.T, 1e-3, self.sde.N)          # host copy: t enters the kernels as a scalar
        for it in range(iterations):
            for i in range(steps_per_iter):
                step = it * steps_per_iter + i
                optimizer.zero_grad()
                if time_strategy == "1":
                    q = int(torch.randint(self.sde.N, [1]))
                elif time_strategy == "2":
                    q = int(sample_time)
                elif time_strategy == "3":
                    q = self.quan_t(step, total_steps, self.sde.N, sample_trun)
                else:
                    raise NotImplementedError("unsupported time sampling strategy")
                # the reference passes quan_t positionally into `weighted` (completion.py:196): weighted = bool(quan_t)
                l_prior = self.loss(x, float(timesteps[q]), weighted=bool(q), z=None if noise is None else noise[step])
                l_data = self.data_loss(x * mask, observation * mask)
                tot = weights["dposer"](l_prior, it) + weights["data"](l_data, it)
                tot.backward()
                optimizer.step()
        return observation * mask + x.detach() * (1.0 - mask)
