"""Counterpart of the reference's lib/dataset/AMASS.py: the AMASS ``.pt`` data format (AMASSDataset :12-182), the pose
normaliser (Posenormalizer :187-259) and the evaluation metrics (Evaler :262-330)."""
import os

import torch

N_POSES = 21


class AMASSDataset(torch.utils.data.Dataset):
    """``{root}/{version}/{subset}/pose_body.pt`` ([N, 63] axis-angle) and optionally ``betas.pt`` ([N, 10]) -- AMASS.py:12-182.

    Host-side data format: tensors stay where ``torch.load`` puts them (CPU); items are ``{'poses': [D] (, 'shapes': [10])}``.
    Normalisation statistics come from ``{root}/{version}/train/{rot_rep}_normalize{1: min-max, 2: z-score}.pt`` and are
    computed from this subset and written there when the file does not exist, exactly like the reference.  ``rot_rep='rot6d'``
    converts with the HIP rotation kernels (needs a GPU; the result is moved back to the poses' device)."""

    def __init__(self, root_path, version="version0", subset="train", sample_interval=None, rot_rep="rot6d", return_shape=False,
                 normalize=True, min_max=True):
        assert subset in ["train", "valid", "test"]
        assert rot_rep in ["axis", "rot6d"]
        self.root_path, self.version, self.subset = root_path, version, subset
        self.sample_interval, self.rot_rep, self.return_shape = sample_interval, rot_rep, return_shape
        self.normalize, self.min_max = normalize, min_max
        self.poses, self.shapes = self.read_data()
        if self.sample_interval:
            self._sample(sample_interval)
        if self.normalize:
            if self.min_max:
                self.min_poses, self.max_poses, self.min_shapes, self.max_shapes = self.Normalize()
            else:
                self.mean_poses, self.std_poses, self.mean_shapes, self.std_shapes = self.Normalize()
        self.real_data_len = len(self.poses)

    def __getitem__(self, idx):
        item = {"poses": self.poses[idx % self.real_data_len]}
        if self.return_shape:
            item["shapes"] = self.shapes[idx % self.real_data_len]
        return item

    def __len__(self):
        return len(self.poses)

    def _sample(self, sample_interval):
        print(f"Class AMASSDataset({self.subset}): sample dataset every {sample_interval} frame")
        self.poses = self.poses[::sample_interval]                 # (the reference leaves `shapes` un-sampled, AMASS.py:56-58)

    def read_data(self):
        data_path = os.path.join(self.root_path, self.version, self.subset)
        poses = torch.load(os.path.join(data_path, "pose_body.pt"))
        shapes = torch.load(os.path.join(data_path, "betas.pt")) if self.return_shape else None
        if self.rot_rep == "rot6d":
            from ..utils.transforms import axis_angle_to_rot6d
            n, home = len(poses), poses.device
            dev = home if poses.is_cuda else torch.device("cuda")
            poses = axis_angle_to_rot6d(poses.reshape(-1, 3).to(dev)).reshape(n, -1).to(home)
        return poses, shapes

    def _stat_path(self):
        return os.path.join(self.root_path, self.version, "train", self.rot_rep + ("_normalize1.pt" if self.min_max else "_normalize2.pt"))

    def Normalize(self):
        path = self._stat_path()
        keys = ("min_poses", "max_poses", "min_shapes", "max_shapes") if self.min_max else ("mean_poses", "std_poses", "mean_shapes", "std_shapes")
        if os.path.exists(path):
            saved = torch.load(path)
            a, b, sa, sb = (saved[k] for k in keys)
        else:
            lo_fn = (lambda x: torch.min(x, dim=0)[0]) if self.min_max else (lambda x: torch.mean(x, dim=0))
            hi_fn = (lambda x: torch.max(x, dim=0)[0]) if self.min_max else (lambda x: torch.std(x, dim=0))
            a, b = lo_fn(self.poses), hi_fn(self.poses)
            sa = lo_fn(self.shapes) if self.return_shape else None
            sb = hi_fn(self.shapes) if self.return_shape else None
            torch.save(dict(zip(keys, (a, b, sa, sb))), path)
        if self.min_max:
            self.poses = 2 * (self.poses - a) / (b - a) - 1
            if self.return_shape:
                self.shapes = 2 * (self.shapes - sa) / (sb - sa) - 1
        else:
            self.poses = (self.poses - a) / b
            if self.return_shape:
                self.shapes = (self.shapes - sa) / sb
        return a, b, sa, sb

    def Denormalize(self, poses, shapes=None):
        assert len(poses.shape) == 2 or len(poses.shape) == 3      # [b, data_dim] or [t, b, data_dim]

        def undo(x, a, b):
            a, b = a.view(1, -1).to(x.device), b.view(1, -1).to(x.device)
            if len(x.shape) == 3:
                a, b = a.unsqueeze(0), b.unsqueeze(0)
            return 0.5 * ((x + 1) * (b - a) + 2 * a) if self.min_max else x * b + a

        a, b, sa = (self.min_poses, self.max_poses, self.min_shapes) if self.min_max else (self.mean_poses, self.std_poses, self.mean_shapes)
        sb = self.max_shapes if self.min_max else self.std_shapes
        out = undo(poses, a, b)
        if shapes is not None and sa is not None:
            return out, undo(shapes, sa, sb)
        return out

    def eval(self, preds):
        pass


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

    def multi_eval_bodys(self, outs, gts, as_tensors=False):
        """outs [b, hypo, 63]: minimum over hypotheses (AMASS.py:300-316).  Returns numpy vectors like the reference, or -- with
        ``as_tensors`` -- device tensors, so a whole evaluation run never synchronises with the host until
        ``distributed.reduce_metric_means`` at the end."""
        res = [self.eval_bodys(outs[:, h].contiguous(), gts) for h in range(outs.shape[1])]
        out = {k: torch.stack([r[k] for r in res], 0).min(0).values for k in ("mpvpe_all", "mpjpe_body")}
        return out if as_tensors else {k: v.cpu().numpy() for k, v in out.items()}

    def print_eval_result(self, eval_result):
        import numpy as np
        print("MPVPE (All): %.2f mm" % np.mean(eval_result["mpvpe_all"]))
        print("MPJPE (Body): %.2f mm" % np.mean(eval_result["mpjpe_body"]))

    def print_multi_eval_result(self, eval_result, hypo_num):
        import numpy as np
        print(f"multihypo {hypo_num} MPVPE (All): %.2f mm" % np.mean(eval_result["mpvpe_all"]))
        print(f"multihypo {hypo_num} MPJPE (Body): %.2f mm" % np.mean(eval_result["mpjpe_body"]))
