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
        asset = model_path if isinstance(model_path, dict) else load_smplx_npz(model_path)
        self.bm = _SMPLCore(asset, num_betas=kwargs.get("num_betas", 10), num_expression_coeffs=kwargs.get("num_expression_coeffs", 10),
                            batch_size=kwargs.get("batch_size", 1), model_type="smplx")
        joints = [constants.JOINT_MAP[i] for i in constants.JOINT_NAMES]
        joints[:25] = constants.SMPLX_OPENPOSE_25                      # smpl.py:55-57
        self.joint_map = torch.tensor(joints, dtype=torch.long)
        self.faces = self.bm.faces_tensor.numpy()
        try:       # smpl.py:59-62: initial pose/shape from SMPL mean params (user-supplied asset)
            from ..utils.transforms import rot6d_to_axis_angle
            mp = np.load(constants.SMPL_MEAN_PATH)
            self.register_buffer("mean_poses", rot6d_to_axis_angle(torch.tensor(mp["pose"], dtype=torch.float32).reshape(-1, 6)).reshape(-1))
            self.register_buffer("mean_shape", torch.tensor(mp["shape"], dtype=torch.float32))
        except Exception:
            self.register_buffer("mean_poses", torch.zeros(72))
            self.register_buffer("mean_shape", torch.zeros(10))

    def forward(self, *args, **kwargs):
        kwargs.pop("get_skin", None)
        if not kwargs.pop("pose2rot", True):
            raise NotImplementedError("pose2rot=False (rotation-matrix inputs) is not built: pass axis-angle poses "
                                      "(utils.transforms.rotmat_to_axis_angle converts)")
        o = self.bm(*args, **kwargs)
        joints = o.joints[:, self.joint_map.to(o.joints.device), :]    # bit-exact gather (smpl.py:70)
        return SMPLOutput(vertices=o.vertices, global_orient=o.global_orient, body_pose=o.body_pose, joints=joints,
                          betas=o.betas, full_pose=o.full_pose)
