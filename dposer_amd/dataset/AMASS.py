"""Pose normaliser -- counterpart of the hot-path surface of the reference's lib/dataset/AMASS.py
(N_POSES :9, Posenormalizer :187-259).  Dataset IO (AMASSDataset) is host-side and out of scope."""
import os

import torch

N_POSES = 21


class Posenormalizer:
    """z-score or min-max (de)normalisation with the statistics of ``{rot_rep}_normalize{1,2}.pt``
    (AMASS.py:187-259).  ``data_path`` may also be a dict with the four tensors."""

    def __init__(self, data_path, device="cuda:0", normalize=True, min_max=True, rot_rep=None):
        assert rot_rep in ["rot6d", "axis"]
        self.normalize, self.min_max, self.rot_rep = normalize, min_max, rot_rep
        if isinstance(data_path, dict):
            p1, p2 = data_path, data_path
        else:
            p1 = torch.load(os.path.join(data_path, "{}_normalize1.pt".format(rot_rep)))
            p2 = torch.load(os.path.join(data_path, "{}_normalize2.pt".format(rot_rep)))
        self.min_poses, self.max_poses = p1["min_poses"].to(device), p1["max_poses"].to(device)
        self.mean_poses, self.std_poses = p2["mean_poses"].to(device), p2["std_poses"].to(device)

    def _stats(self, a, b, poses):
        a, b = a.view(1, -1), b.view(1, -1)
        if len(poses.shape) == 3:
            a, b = a.unsqueeze(0), b.unsqueeze(0)
        return a, b

    def offline_normalize(self, poses, from_axis=False):
        assert len(poses.shape) in (2, 3)
        if from_axis and self.rot_rep == "rot6d":
            from ..utils.transforms import axis_angle_to_rot6d
            poses = axis_angle_to_rot6d(poses.reshape(-1, 3)).reshape(*poses.shape[:-1], -1)
        if not self.normalize:
            return poses
        if self.min_max:
            lo, hi = self._stats(self.min_poses, self.max_poses, poses)
            return 2 * (poses - lo) / (hi - lo) - 1
        mean, std = self._stats(self.mean_poses, self.std_poses, poses)
        return (poses - mean) / std

    def offline_denormalize(self, poses, to_axis=False):
        assert len(poses.shape) in (2, 3)
        out = poses
        if self.normalize:
            if self.min_max:
                lo, hi = self._stats(self.min_poses, self.max_poses, poses)
                out = 0.5 * ((poses + 1) * (hi - lo) + 2 * lo)
            else:
                mean, std = self._stats(self.mean_poses, self.std_poses, poses)
                out = poses * std + mean
        if to_axis and self.rot_rep == "rot6d":
            from ..utils.transforms import rot6d_to_axis_angle
            out = rot6d_to_axis_angle(out.reshape(-1, 6)).reshape(*out.shape[:-1], -1)
        return out


class Evaler:
    """Completion evaluator -- counterpart of lib/dataset/AMASS.py:263-324.  The reference copies the whole
    [B, 10475, 3] vertex tensor to the host inside a per-sample python loop; here the part-vertex / part-joint errors
    are reduced on the device and only the [B] metric vectors travel."""

    def __init__(self, body_model, part=None):
        import numpy as np
        from ..body_model.utils import BodyPartIndices, BodySegIndices
        self.body_model = body_model
        self.part = part
        if part is not None:
            self.joint_idx = torch.tensor(np.array(getattr(BodyPartIndices, part)) + 1)      # skip pelvis
            self.vert_idx = torch.tensor(np.array(getattr(BodySegIndices, part)))
        else:
            self.joint_idx = self.vert_idx = None

    def eval_bodys(self, outs, gts):
        """outs, gts [b, 63] axis-angle body poses -> {'mpvpe_all': [b], 'mpjpe_body': [b]} in mm (device tensors)."""
        with torch.no_grad():
            body_gt = self.body_model(pose_body=gts)
            body_out = self.body_model(pose_body=outs)
            dv, dj = body_out.v - body_gt.v, body_out.Jtr - body_gt.Jtr
            if self.vert_idx is not None:
                dv = dv[:, self.vert_idx.to(dv.device)]
                dj = dj[:, self.joint_idx.to(dj.device)]
            return {"mpvpe_all": torch.sqrt((dv ** 2).sum(-1)).mean(-1) * 1000, "mpjpe_body": torch.sqrt((dj ** 2).sum(-1)).mean(-1) * 1000}

    def multi_eval_bodys(self, outs, gts):
        """outs [b, hypo, 63]: minimum over hypotheses (AMASS.py:300-316); returns numpy vectors like the reference."""
        res = [self.eval_bodys(outs[:, h].contiguous(), gts) for h in range(outs.shape[1])]
        return {k: torch.stack([r[k] for r in res], 0).min(0).values.cpu().numpy() for k in ("mpvpe_all", "mpjpe_body")}

    def print_eval_result(self, eval_result):
        import numpy as np
        print("MPVPE (All): %.2f mm" % np.mean(eval_result["mpvpe_all"]))
        print("MPJPE (Body): %.2f mm" % np.mean(eval_result["mpjpe_body"]))

    def print_multi_eval_result(self, eval_result, hypo_num):
        import numpy as np
        print(f"multihypo {hypo_num} MPVPE (All): %.2f mm" % np.mean(eval_result["mpvpe_all"]))
        print(f"multihypo {hypo_num} MPJPE (Body): %.2f mm" % np.mean(eval_result["mpjpe_body"]))
