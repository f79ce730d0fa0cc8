"""Rotation-representation conversions -- counterpart of the rotation part of the reference's
lib/utils/transforms.py (rot6d_to_axis_angle :197-224, rot6d_to_mat3x3 :227-235,
axis_angle_to_rot6d :238-255, axis_angle_to_mat3x3 :258-261).

``rot6d_to_mat3x3`` and ``axis_angle_to_mat3x3`` are HIP kernels (dposer_rot6d_to_rotmat,
dposer_rodrigues).  The reference delegates the axis-angle <-> matrix directions to the un-vendored
``torchgeometry``; its published algorithms are restated here with torch ops (parity unpinned:
torchgeometry is absent and the reference holds no test for them).  Camera / Procrustes helpers of
the reference are host-side numpy outside the hot path and are not rebuilt.
"""
import torch
import torch.nn.functional as F

from .. import _C


def _launch_rot(fn_name, x, in_w):
    _C.require_gpu(x, fn_name + " input")
    x = x.reshape(-1, in_w).contiguous().float()
    out = torch.empty(x.shape[0], 3, 3, dtype=torch.float32, device=x.device)
    if x.shape[0] == 0:
        return out
    _C.check(getattr(_C.lib(), fn_name)(_C.ptr(x), _C.ptr(out), x.shape[0], _C.stream_ptr()), fn_name)
    return out


def rot6d_to_mat3x3(rot6d):
    """[n, 6] (row-major 3x2 = first two columns of R) -> [n, 3, 3] by Gram-Schmidt (transforms.py:227-235)."""
    return _launch_rot("dposer_rot6d_to_rotmat", rot6d, 6)


def batch_rodrigues(rot_vecs):
    """smplx.lbs.batch_rodrigues: [n, 3] axis-angle -> [n, 3, 3]."""
    return _launch_rot("dposer_rodrigues", rot_vecs, 3)


def axis_angle_to_mat3x3(angle_axis):
    """transforms.py:258-261 (tgm.angle_axis_to_rotation_matrix(...)[:, :3, :3]).  torchgeometry switches to a
    first-order Taylor form for theta^2 <= 1e-6; the Rodrigues kernel (angle = ||r + 1e-8||) agrees to fp32
    rounding there."""
    return batch_rodrigues(angle_axis)


def axis_angle_to_rot6d(angle_axis):
    """transforms.py:238-255: first two columns of the rotation matrix, row-major."""
    return axis_angle_to_mat3x3(angle_axis)[:, :3, :2].reshape(-1, 6)


def rotmat_to_axis_angle(R):
    """Rotation matrix [n,3,3] -> axis-angle [n,3] through the unit quaternion (the route
    torchgeometry.rotation_matrix_to_angle_axis takes: matrix -> quaternion -> angle-axis)."""
    m = R.reshape(-1, 3, 3)
    t = m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2]
    qw = torch.sqrt(torch.clamp(1.0 + t, min=1e-12)) * 0.5
    qx = torch.sqrt(torch.clamp(1.0 + m[:, 0, 0] - m[:, 1, 1] - m[:, 2, 2], min=1e-12)) * 0.5
    qy = torch.sqrt(torch.clamp(1.0 - m[:, 0, 0] + m[:, 1, 1] - m[:, 2, 2], min=1e-12)) * 0.5
    qz = torch.sqrt(torch.clamp(1.0 - m[:, 0, 0] - m[:, 1, 1] + m[:, 2, 2], min=1e-12)) * 0.5
    qx = torch.copysign(qx, m[:, 2, 1] - m[:, 1, 2])
    qy = torch.copysign(qy, m[:, 0, 2] - m[:, 2, 0])
    qz = torch.copysign(qz, m[:, 1, 0] - m[:, 0, 1])
    sin_half = torch.sqrt(qx * qx + qy * qy + qz * qz)
    angle = 2.0 * torch.atan2(sin_half, qw)
    k = torch.where(sin_half > 1e-8, angle / torch.clamp(sin_half, min=1e-8), torch.full_like(angle, 2.0))
    return torch.stack([qx * k, qy * k, qz * k], dim=1)


def rot6d_to_axis_angle(rot6d):
    """transforms.py:197-224."""
    aa = rotmat_to_axis_angle(rot6d_to_mat3x3(rot6d))
    aa[torch.isnan(aa)] = 0.0
    return aa
