"""Caller-side glue of the hot path -- counterpart of the reference's lib/utils/misc.py
(add_noise :11-24, create_mask :27-55, linear_interpolation :58-61, slerp :64-69,
moving_average :72-81, gaussian_smoothing :84-95).  Cheap torch ops; index sets are bit-exact."""
import torch
import torch.nn.functional as F

from ..body_model.utils import BodyPartIndices
from ..dataset.AMASS import N_POSES


def add_noise(gts, std=0.5, noise_type="gaussian"):
    if std == 0.0:
        return gts
    if noise_type == "gaussian":
        return gts + std * torch.randn(*gts.shape, device=gts.device)
    if noise_type == "uniform":
        return gts + std * (torch.rand(*gts.shape, device=gts.device) - 0.5)
    raise NotImplementedError


def mask_indices(part, rot_N):
    """Flat pose-vector indices of a body part: joint * rot_N + arange(rot_N) (misc.py:33-36)."""
    joints = torch.tensor(getattr(BodyPartIndices, part)).view(-1, 1)
    return (joints * rot_N + torch.arange(rot_N).view(1, -1)).flatten()


def create_mask(body_poses, part="legs", observation_type="noise"):
    """mask [B, D] (0 on the masked part) and the observation with the masked entries replaced by
    N(0, 1) noise, or by the SMPL mean pose for any other ``observation_type`` (misc.py:27-55)."""
    assert len(body_poses.shape) == 2 and body_poses.shape[1] % N_POSES == 0
    rot_N = body_poses.shape[1] // N_POSES
    assert rot_N in [3, 6]
    idx = mask_indices(part, rot_N).to(body_poses.device)
    mask = body_poses.new_ones(body_poses.shape)
    mask[:, idx] = 0
    observation = body_poses.clone()
    if observation_type == "noise":
        observation[:, idx] = torch.randn_like(observation[:, idx])
    else:
        # the mean pose as observation (misc.py:44-53): 'pose' of the SMPL mean-params file is [144] rot6d, body joints from 6 on
        import numpy as np
        from ..body_model import constants
        from .transforms import rot6d_to_axis_angle
        mean = np.load(constants.SMPL_MEAN_PATH)
        rot6d_body = torch.tensor(mean["pose"][6:], dtype=torch.float32, device=body_poses.device)       # [138]
        if rot_N == 3:
            fill = rot6d_to_axis_angle(rot6d_body.reshape(-1, 6)).reshape(-1)                          # [69]
        else:
            fill = rot6d_body
        observation[:, idx] = fill[idx][None].repeat(body_poses.shape[0], 1)
    return mask, observation


def linear_interpolation(A, B, frames):
    alpha = torch.linspace(0, 1, frames, device=A.device)[:, None]
    return (1 - alpha) * A + alpha * B


def slerp_interpolation(A, B, frames):
    omega = torch.acos((A * B).sum() / (torch.norm(A) * torch.norm(B)))
    alpha = torch.linspace(0, 1, frames, device=A.device)[:, None]
    return (torch.sin((1 - alpha) * omega) / torch.sin(omega)) * A + (torch.sin(alpha * omega) / torch.sin(omega)) * B


def _smooth(data, kernel):
    w = kernel.numel()
    x = data.transpose(0, 1).unsqueeze(1)
    y = F.conv1d(x, kernel.view(1, 1, -1).to(data.device), padding=w // 2)
    return y.squeeze(1).transpose(0, 1)


def moving_average(data, window_size):
    return _smooth(data, torch.ones(window_size) / window_size)


def gaussian_smoothing(data, window_size, sigma):
    k = torch.exp(-0.5 * ((torch.arange(window_size).float() - window_size // 2) / sigma) ** 2)
    return _smooth(data, k / k.sum())
