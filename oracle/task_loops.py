"""CPU restatement of the reference's task loops around the DPoser prior -- TEST INFRASTRUCTURE (see oracle/__init__.py).

* ``completion_optimize``      run/completion.py:167-207 (``DPoserComp.optimize``) with the loss of :131-149, the weights of
                               :151-155 and ``backward_step`` :157-165.
* ``motion_denoise_optimize``  run/motion_denoising.py:199-300 (``MotionDenoise.optimize``) with ``DPoser_loss`` :124-143, the
                               weights of :157-163, the ``if data_term > 0`` guard :261-263 and the final Gaussian smoothing
                               (lib/utils/misc.py:84-95).

PINNED: ``tests/golden/g14_completion_loop.npz`` and ``g15_motion_denoise_loop.npz`` hold outputs of the reference's OWN loops
(``run.completion.DPoserComp.optimize`` / ``run.motion_denoising.MotionDenoise.optimize`` imported in the build container,
``tests/golden/gen_golden.py g14 g15``), and ``tests/test_oracle_golden.py`` checks these restatements against them.  The body
model handed to the reference's motion-denoising loop is ``oracle.fk_torch`` on the synthetic SMPL-X-shaped asset (smplx itself
is absent), so g15 pins the LOOP -- weights, time schedule, guard, optimiser, smoothing, metrics -- not the LBS arithmetic.
"""
import math

import numpy as np
import torch

from . import fk_torch
from . import score_ref as R


def quan_t_strategy3(step: int, total_steps: int, N: int, sample_trun: float) -> int:
    """``N - floor(tensor(total - step - 1) * (N / (trun * total))) - 2`` (completion.py:189-190, motion_denoising.py:245): an int64
    tensor times a python float is evaluated in fp32."""
    return int(N - math.floor(float(np.float32(total_steps - step - 1) * np.float32(N / (sample_trun * total_steps)))) - 2)


def completion_optimize(p, sde, observation, mask, noise, *, iterations=2, steps_per_iter=100, lr=0.1, sample_trun=5.0):
    """run/completion.py:167-207, time strategy '3'.  ``noise[step]`` is the z of completion.py:133.
    The reference passes ``quan_t`` positionally into ``weighted`` (:196), i.e. weighted = bool(quan_t)."""
    obs = torch.as_tensor(observation, dtype=torch.float32)
    msk = torch.as_tensor(mask, dtype=torch.float32)
    x = obs.clone().requires_grad_(True)
    opt = torch.optim.Adam([x], lr, betas=(0.9, 0.999))
    ts = torch.linspace(1.0, 1e-3, sde.N)
    total = iterations * steps_per_iter
    for it in range(iterations):
        for i in range(steps_per_iter):
            step = it * steps_per_iter + i
            opt.zero_grad()
            q = quan_t_strategy3(step, total, sde.N, sample_trun)
            t = torch.ones(x.shape[0]) * ts[q]
            _, g = R.dposer_prior_loss(p, sde, x.detach(), t, torch.as_tensor(noise[step]), weighted=bool(q), reduction="mean")
            ld = torch.nn.functional.mse_loss(x * msk, obs * msk)                  # data_loss, :197
            (100 * ld / (1 + it)).backward()                                       # weights :151-155
            x.grad += 0.1 * (it + 1) * g
            opt.step()
    return (obs * msk + x.detach() * (1.0 - msk)).numpy()


def gaussian_smoothing(x, window_size=3, sigma=2.0):
    """lib/utils/misc.py:84-95: depth-wise 1-D Gaussian filter along the frame axis, zero padded."""
    k = torch.arange(window_size, dtype=x.dtype) - window_size // 2
    w = torch.exp(-0.5 * (k / sigma) ** 2)
    w = w / w.sum()
    xt = x.t()[None]                                                               # [1, D, T]
    pad = window_size // 2
    out = torch.nn.functional.conv1d(torch.nn.functional.pad(xt, (pad, pad)), w.view(1, 1, -1).repeat(x.shape[1], 1, 1), groups=x.shape[1])
    return out[0].t()


def motion_denoise_optimize(p, sde, asset, mean, std, joints3d, gt_poses, init_poses, noise, *, iterations=5, steps_per_iter=50,
                            sample_trun=2.0, dposer_weight=1.0, body_dtype=torch.float64):
    """run/motion_denoising.py:199-300, time strategy '3', z-score normaliser.  Returns (final pose_body before smoothing,
    {'init_MPJPE', 'MPJPE', 'MPVPE'}).  The body model is fk_torch (float64; ``body_dtype=torch.float32`` = the precision smplx runs in inside the reference); poses are an fp32 leaf like in the reference."""
    T = init_poses.shape[0]
    mean = torch.as_tensor(mean, dtype=torch.float32)
    std = torch.as_tensor(std, dtype=torch.float32)
    joints = torch.as_tensor(joints3d, dtype=body_dtype)
    bm = lambda pose: fk_torch.smplx_forward(asset, pose.to(body_dtype), dtype=body_dtype)
    with torch.no_grad():
        v_gt, j_gt = bm(torch.as_tensor(gt_poses, dtype=torch.float32))
    je = joints - j_gt[:, :22]
    init_mpjpe = torch.mean(torch.sqrt(torch.sum(je * je, dim=2)), dim=1) * 100.0
    pose = torch.as_tensor(init_poses, dtype=torch.float32).clone().requires_grad_(True)
    opt = torch.optim.Adam([pose], 0.03, betas=(0.9, 0.999))
    ts = torch.linspace(1.0, 1e-3, sde.N)
    total = iterations * steps_per_iter
    for it in range(iterations):
        for i in range(steps_per_iter):
            step = it * steps_per_iter + i
            opt.zero_grad()
            q = quan_t_strategy3(step, total, sde.N, sample_trun)
            x0 = ((pose - mean) / std).detach()                                    # offline_normalize(from_axis=True), z-score
            # DPoser_loss(x_0, vec_t, quan_t, weighted=False) (:124, called :250): unweighted here -- unlike completion.py:196,
            # whose loss() has no quan_t parameter, so quan_t lands in `weighted` there
            _, gprior = R.dposer_prior_loss(p, sde, x0, torch.ones(T) * ts[q], torch.as_tensor(noise[step]), weighted=False,
                                            reduction="sum_over_batch", batch_size=T)
            v, j = bm(pose)
            temp = v[:-1] - v[1:]
            l_temp = torch.mean(torch.sqrt(torch.sum(temp * temp, dim=2)))
            data = j[:, :22] - joints
            l_data = torch.mean(torch.sqrt(torch.sum(data * data, dim=2)))
            tot = 10.0 * l_temp * (1 + it)                                         # weights :157-163
            if bool(l_data > 0):                                                   # :261-263
                tot = tot + 100.0 * l_data / (1 + it * it)
            tot.backward()
            pose.grad += (0.1 * (1 + it) * dposer_weight) * (gprior / std)
            opt.step()
    final = pose.detach()
    smooth = gaussian_smoothing(final, 3, 2.0)
    smooth[[0, -1]] = final[[0, -1]]
    with torch.no_grad():
        v, j = bm(smooth)
    je = j[:, :22] - j_gt[:, :22]
    ve = v - v_gt
    res = {"init_MPJPE": init_mpjpe.numpy(), "MPJPE": (torch.mean(torch.sqrt(torch.sum(je * je, dim=2)), dim=1) * 100.0).numpy(),
           "MPVPE": (torch.mean(torch.sqrt(torch.sum(ve * ve, dim=2)), dim=1) * 100.0).numpy()}
    return final.numpy(), res
