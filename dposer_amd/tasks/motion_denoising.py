"""Motion denoising: DPoser prior + SMPL-X FK/LBS fitting of noisy 3D joints -- counterpart of
``MotionDenoise`` in the reference's run/motion_denoising.py:63-300.

Per Adam step: z-score normalise -> ``dposer_prior_loss`` -> ``BodyModel`` forward WITH gradient
(dposer_lbs_forward / dposer_lbs_backward) -> temporal term on vertices + data term on Jtr[:, :22].
The frames of one sequence are coupled by the temporal term, so data parallelism is over sequences.
"""
import math

import numpy as np
import torch

from ..algorithms.advanced import sde_lib
from ..prior import prior_loss
from ..utils.misc import gaussian_smoothing


class MotionDenoise:
    def __init__(self, config, args, diffusion_model, body_model, sde_N=1000, dposer_weight=1.0, out_path=None, debug=False,
                 batch_size=1, normalizer=None):
        from ..dataset.AMASS import Posenormalizer
        self.args, self.debug, self.device = args, debug, args.device
        self.body_model = body_model
        self.dposer_weight = dposer_weight
        self.out_path = out_path
        self.batch_size = batch_size
        self.betas = torch.zeros((batch_size, 10), device=self.device)
        self.poses = torch.randn((batch_size, 63), device=self.device) * 0.01
        self.Normalizer = normalizer if normalizer is not None else Posenormalizer(
            data_path=f"{args.dataset_folder}/{args.version}/train", normalize=config.data.normalize, min_max=config.data.min_max,
            rot_rep=config.data.rot_rep, device=args.device)
        name = config.training.sde.lower()
        if name == "vpsde":
            sde = sde_lib.VPSDE(beta_min=config.model.beta_min, beta_max=config.model.beta_max, N=config.model.num_scales)
        elif name == "subvpsde":
            sde = sde_lib.subVPSDE(beta_min=config.model.beta_min, beta_max=config.model.beta_max, N=config.model.num_scales)
        elif name == "vesde":
            sde = sde_lib.VESDE(sigma_min=config.model.sigma_min, sigma_max=config.model.sigma_max, N=config.model.num_scales)
        else:
            raise NotImplementedError(f"SDE {config.training.sde} unknown.")
        sde.N = sde_N
        self.sde = sde
        self.model = diffusion_model
        self._calls = 0

    def DPoser_loss(self, x_0, t, weighted=False, z=None):
        """motion_denoising.py:124-143: sum(weight * (x_0 - x0_hat)^2) / batch_size."""
        self._calls += 1
        return prior_loss(self.model, self.sde, x_0, t, weighted=weighted, reduction="sum_over_batch", batch_size=self.batch_size, z=z,
                          seed=self.model._rng_seed + 31, step=self._calls)

    def get_loss_weights(self):
        """motion_denoising.py:157-163."""
        return {"temp": lambda cst, it: 10.0 ** 1 * cst * (1 + it), "data": lambda cst, it: 10.0 ** 2 * cst / (1 + it * it),
                "dposer": lambda cst, it: 10.0 ** -1 * cst * (1 + it) * self.dposer_weight}

    def optimize(self, joints3d, gt_poses=None, time_strategy="1", sample_trun=2.0, sample_time=990, iterations=5, steps_per_iter=50,
                 verbose=False, vis=False, noise=None, init_poses=None):
        """motion_denoising.py:199-300 (visualisation dropped).  Returns {'init_MPJPE', 'MPJPE', 'MPVPE'} in cm per frame."""
        bm = self.body_model
        with torch.no_grad():
            gt = bm(betas=self.betas, pose_body=gt_poses)
            joint_error = joints3d - gt.Jtr[:, :22]
            init_mpjpe = torch.mean(torch.sqrt(torch.sum(joint_error * joint_error, dim=2)), dim=1) * 100.0
        init_joints = joints3d.detach()
        pose = (self.poses if init_poses is None else init_poses).clone().detach().requires_grad_(True)
        optimizer = torch.optim.Adam([pose], 0.03, betas=(0.9, 0.999))
        weights = self.get_loss_weights()
        timesteps = torch.linspace(self.sde.T, 1e-3, self.sde.N)
        total_steps = iterations * steps_per_iter
        for it in range(iterations):
            for i in range(steps_per_iter):
                step = it * steps_per_iter + i
                optimizer.zero_grad()
                poses_n = self.Normalizer.offline_normalize(pose, from_axis=True)
                if time_strategy == "1":
                    q = int(torch.randint(self.sde.N, [1]))
                elif time_strategy == "2":
                    q = int(sample_time)
                elif time_strategy == "3":
                    q = int(self.sde.N - math.floor(float(np.float32(total_steps - step - 1) * np.float32(self.sde.N / (sample_trun * total_steps)))) - 2)
                else:
                    raise NotImplementedError("unsupported time sampling strategy")
                losses = {"dposer": self.DPoser_loss(poses_n, float(timesteps[q]), z=None if noise is None else noise[step])}
                body = bm(betas=self.betas, pose_body=pose)                          # forward WITH gradient
                temp = body.v[:-1] - body.v[1:]
                losses["temp"] = torch.mean(torch.sqrt(torch.sum(temp * temp, dim=2)))
                data = body.Jtr[:, :22] - init_joints
                # motion_denoising.py:262 keeps the data term only `if data_term > 0` ("for nans"): a host sync per step whose
                # real job is the all-zero case (pose still equal to the observation: value 0, but sqrt'(0) = inf poisons the
                # backward pass).  Same effect without the sync: distances are clamped away from 0 before the sqrt, so a zero
                # residual contributes the value ~0 and a ZERO gradient, and a non-finite / non-positive term is dropped by value.
                dist = torch.sqrt(torch.sum(data * data, dim=2).clamp_min(1e-36))
                data_term = torch.mean(dist)
                losses["data"] = torch.where(torch.isfinite(data_term) & (data_term > 0), data_term, torch.zeros_like(data_term))
                tot = torch.stack([weights[k](v, it) for k, v in losses.items()]).sum()
                tot.backward()
                optimizer.step()
        with torch.no_grad():
            final = pose.detach()
            smooth = gaussian_smoothing(final, window_size=3, sigma=2)
            smooth[[0, -1]] = final[[0, -1]]
            out = bm(betas=self.betas, pose_body=smooth)
            je = out.Jtr[:, :22] - gt.Jtr[:, :22]
            ve = out.v - gt.v
            mpjpe = torch.mean(torch.sqrt(torch.sum(je * je, dim=2)), dim=1) * 100.0
            mpvpe = torch.mean(torch.sqrt(torch.sum(ve * ve, dim=2)), dim=1) * 100.0
        return {"init_MPJPE": init_mpjpe.cpu().numpy(), "MPJPE": mpjpe.cpu().numpy(), "MPVPE": mpvpe.cpu().numpy(), "pose_body": final}
