"""SMPLX wrapper that remaps joints to 49 OpenPose + ground-truth joints -- counterpart of the
reference's lib/body_model/smpl.py:49-77."""
import numpy as np
import torch

from . import constants
from .body_model import _SMPLCore, Struct
from .synthetic import load_smplx_npz

SMPLOutput = Struct
JOINT_IDS = {constants.JOINT_NAMES[i]: i for i in range(len(constants.JOINT_NAMES))}


class SMPLX(torch.nn.Module):
    def __init__(self, model_path, **kwargs):
        super().__init__()
        nb, ne = kwargs.get("num_betas", 10), kwargs.get("num_expression_coeffs", 10)
        if isinstance(model_path, dict):
            asset = model_path
        else:                                                           # (file or directory: sizes clamped to the file's, as smplx does)
            asset = load_smplx_npz(model_path, nb, ne)
            nb, ne = asset["num_betas"], asset["num_expressions"]
        self.bm = _SMPLCore(asset, num_betas=nb, num_expression_coeffs=ne,
                            batch_size=kwargs.get("batch_size", 1), model_type="smplx")
        joints = [constants.JOINT_MAP[i] for i in constants.JOINT_NAMES]
        joints[:25] = constants.SMPLX_OPENPOSE_25                      # smpl.py:55-57
        self.joint_map = torch.tensor(joints, dtype=torch.long)
        self.faces = self.bm.faces_tensor.numpy()
        # smpl.py:59-62: initial pose / shape from the SMPL mean parameters (a user-supplied asset, like the body model itself).
        # The 6D -> axis-angle conversion is a HIP kernel (utils.transforms.rot6d_to_axis_angle): it runs as soon as the module
        # sits on a GPU -- at construction when one is visible, else at the first .to(device) -- and never silently yields zeros:
        # a missing file is the only case that leaves zeros behind (``mean_params_loaded`` says so), anything else raises.
        self.mean_params_loaded = False
        self._mean_converted = False
        self.register_buffer("mean_poses", torch.zeros(72))
        self.register_buffer("mean_shape", torch.zeros(10))
        self.register_buffer("mean_pose_rot6d", torch.zeros(24, 6))
        import os
        if os.path.exists(constants.SMPL_MEAN_PATH):
            mp = np.load(constants.SMPL_MEAN_PATH)
            self.mean_pose_rot6d = torch.tensor(mp["pose"], dtype=torch.float32).reshape(-1, 6)
            self.mean_poses = torch.zeros(self.mean_pose_rot6d.shape[0] * 3)
            self.mean_shape = torch.tensor(mp["shape"], dtype=torch.float32)
            self.mean_params_loaded = True
            if torch.cuda.is_available():
                self._convert_mean_pose(torch.device("cuda", torch.cuda.current_device()))

    def _convert_mean_pose(self, device):
        from ..utils.transforms import rot6d_to_axis_angle
        aa = rot6d_to_axis_angle(self.mean_pose_rot6d.to(device).contiguous()).reshape(-1)
        self.mean_poses = aa.to(self.mean_poses.device)
        self._mean_converted = True

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if self.mean_params_loaded and not self._mean_converted and self.mean_pose_rot6d.is_cuda:
            self._convert_mean_pose(self.mean_pose_rot6d.device)
        return out

    def forward(self, *args, **kwargs):
        kwargs.pop("get_skin", None)
        if not kwargs.pop("pose2rot", True):
            # smplx: the pose arguments are rotation matrices [B, n, 3, 3].  The FK kernels take axis-angle vectors (their
            # Rodrigues step is fused into the chain), so the matrices go through the HIP log map first; exp(log(R)) = R to
            # fp32 rounding (tests/test_gpu_fk.py: 1e-6 on the joints)
            from ..utils.transforms import rotmat_to_axis_angle
            for key in ("global_orient", "body_pose", "jaw_pose", "leye_pose", "reye_pose", "left_hand_pose", "right_hand_pose"):
                v = kwargs.get(key)
                if v is not None:
                    if v.shape[-2:] != (3, 3):
                        raise ValueError(f"pose2rot=False: {key} must hold rotation matrices [..., 3, 3], got {tuple(v.shape)}")
                    kwargs[key] = rotmat_to_axis_angle(v.reshape(-1, 3, 3).contiguous()).reshape(v.shape[0], -1)
        o = self.bm(*args, **kwargs)
        joints = o.joints[:, self.joint_map.to(o.joints.device), :]    # bit-exact gather (smpl.py:70)
        return SMPLOutput(vertices=o.vertices, global_orient=o.global_orient, body_pose=o.body_pose, joints=joints,
                          betas=o.betas, full_pose=o.full_pose)
