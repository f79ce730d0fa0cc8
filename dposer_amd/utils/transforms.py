"""Rotation-representation conversions -- counterpart of the rotation part of the reference's
lib/utils/transforms.py (rot6d_to_axis_angle :197-224, rot6d_to_mat3x3 :227-235,
axis_angle_to_rot6d :238-255, axis_angle_to_mat3x3 :258-261).

All four directions are HIP kernels (dposer_rot6d_to_rotmat, dposer_rodrigues, dposer_rotmat_to_axis_angle,
dposer_rot6d_to_axis_angle).  The reference delegates the axis-angle <-> matrix directions to the un-vendored
``torchgeometry``; its published algorithms are restated (torchgeometry is absent and the reference holds no test for them:
parity unpinned against torchgeometry itself, but pinned against ``scipy.spatial.transform.Rotation`` -- an independent
implementation of the same maps -- in tests/test_gpu_fk.py).  Camera / Procrustes helpers of
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


def _launch_to_aa(fn_name, x, in_w):
    _C.require_gpu(x, fn_name + " input")
    x = x.reshape(-1, in_w).contiguous().float()
    out = torch.empty(x.shape[0], 3, dtype=torch.float32, device=x.device)
    if x.shape[0]:
        _C.check(getattr(_C.lib(), fn_name)(_C.ptr(x), _C.ptr(out), x.shape[0], _C.stream_ptr()), fn_name)
    return out


def rotmat_to_axis_angle(R):
    """Rotation matrix [n,3,3] -> axis-angle [n,3] through the unit quaternion (the route
    torchgeometry.rotation_matrix_to_angle_axis takes: matrix -> quaternion -> angle-axis); one HIP kernel."""
    return _launch_to_aa("dposer_rotmat_to_axis_angle", R, 9)


def rot6d_to_axis_angle(rot6d):
    """transforms.py:197-224 (Gram-Schmidt + matrix -> axis-angle + NaN -> 0) in one HIP kernel."""
    return _launch_to_aa("dposer_rot6d_to_axis_angle", rot6d, 6)
