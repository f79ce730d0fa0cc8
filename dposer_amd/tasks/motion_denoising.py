"""Motion denoising: DPoser prior + SMPL-X FK/LBS fitting of noisy 3D joints -- counterpart of
``MotionDenoise`` in the reference's run/motion_denoising.py:63-300.

Per Adam step: z-score normalise -> ``dposer_prior_loss`` -> ``BodyModel`` forward WITH gradient
(dposer_lbs_forward / dposer_lbs_backward) -> temporal term on vertices + data term on Jtr[:, :22].
The frames of one sequence are coupled by the temporal term, so data parallelism is over sequences.

The whole loop is ONE call, ``dposer_motion_denoise_optimize`` (axis-angle or 6-D rotation representation, sub-VP / VP SDE, positional or Fourier time embedding):
all steps are queued from C, the loss gradients and torch.optim.Adam's update are kernels, nothing returns to the host in
between.  Other configurations (and ``fused=False``) run the same step through autograd and torch's Adam.
"""
import ctypes as C
import math

import numpy as np
import torch

from .. import _C
from ..algorithms.advanced import sde_lib
from ..prior import prior_loss
from ..utils.misc import gaussian_smoothing


def _axis_angle_to_rot6d_autograd(aa):
    """Differentiable axis-angle -> 6-D (first two columns of the rotation matrix, row-major: lib/utils/transforms.py:238-255) for the
    step-by-step loop with rot_rep = 'rot6d' -- the HIP conversion kernel has no backward.  Rodrigues as smplx writes it
    (angle = ||r + 1e-8||), i.e. what the one-call loop's kernels evaluate."""
    angle = torch.norm(aa + 1e-8, dim=1, keepdim=True)
    k = aa / angle
    s, c = torch.sin(angle), torch.cos(angle)
    kx, ky, kz = k[:, 0:1], k[:, 1:2], k[:, 2:3]
    c1 = 1.0 - c
    R00 = 1.0 + c1 * (-(kz * kz) - ky * ky)
    R01 = s * (-kz) + c1 * (kx * ky)
    R10 = s * kz + c1 * (kx * ky)
    R11 = 1.0 + c1 * (-(kz * kz) - kx * kx)
    R20 = s * (-ky) + c1 * (kx * kz)
    R21 = s * kx + c1 * (ky * kz)
    return torch.cat([R00, R01, R10, R11, R20, R21], dim=1)


def _normalize_with_grad(nz, pose):
    """Posenormalizer.offline_normalize(pose, from_axis=True) with a gradient path to the axis-angle pose also for rot_rep = 'rot6d'."""
    if getattr(nz, "rot_rep", "axis") != "rot6d":
        return nz.offline_normalize(pose, from_axis=True)
    six = _axis_angle_to_rot6d_autograd(pose.reshape(-1, 3)).reshape(*pose.shape[:-1], -1)
    return nz.offline_normalize(six, from_axis=False)


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
        self.continuous = bool(getattr(config.training, "continuous", True))      # motion_denoising.py:94: the score function's flavour
        self.model = diffusion_model
        self._calls = 0

    def DPoser_loss(self, x_0, t, weighted=False, z=None):
        """motion_denoising.py:124-143: sum(weight * (x_0 - x0_hat)^2) / batch_size."""
        self._calls += 1
        return prior_loss(self.model, self.sde, x_0, t, weighted=weighted, reduction="sum_over_batch", batch_size=self.batch_size, z=z,
                          seed=self.model._rng_seed + 31, step=self._calls, continuous=getattr(self, "continuous", True))

    def get_loss_weights(self):
        """motion_denoising.py:157-163."""
        return {"temp": lambda cst, it: 10.0 ** 1 * cst * (1 + it), "data": lambda cst, it: 10.0 ** 2 * cst / (1 + it * it),
                "dposer": lambda cst, it: 10.0 ** -1 * cst * (1 + it) * self.dposer_weight}

    def _fused_supported(self):
        from ..algorithms.advanced.model import ScoreModelFC
        from ..body_model.body_model import BodyModel
        nz = self.Normalizer
        return (sde_lib.sde_desc(self.sde, bool(getattr(self, "continuous", True))) is not None and isinstance(self.model, ScoreModelFC)
                and isinstance(self.body_model, BodyModel)
                and getattr(nz, "rot_rep", None) in ("axis", "rot6d") and self.batch_size >= 2)

    def _optimize_fused(self, pose, init_joints, t_list, its, weights, noise, frames_per_sequence=0, betas=None):
        """All optimisation steps in one C call; ``pose`` [T, 63] is updated in place (T = sequences x frames_per_sequence when a
        batch of sequences is advanced together).  Returns the per-step loss log [steps, sequences, 3] (temp, data, prior)."""
        model, core, nz = self.model, self.body_model.bm, self.Normalizer
        dev = pose.device
        T, D = pose.shape
        n_steps = len(t_list)
        eng = model._engine()
        flat = model.flat_params()
        packed = eng.packed(flat, with_backward=False, force=not model.freeze_packed)      # once per loop
        ws = eng.workspace(T, _C.WS_SHARED_T, n_steps, dev)
        lib, h = _C.lib(), core._handle()
        F = int(frames_per_sequence) if frames_per_sequence else T
        n_seq = T // F
        if betas is None:
            betas = self.betas if self.betas.shape[0] == T else self.betas[:1]
        if betas.shape[0] > 1 and bool((betas == betas[:1]).all()):
            # one body shape for every frame (the module's default: motion_denoising.py:64 keeps zeros((batch_size, 10))): the rest
            # shape is formed once and shared -- the skinning kernels then read 126 KB from L2 instead of a [T, V, 3] copy from HBM
            # in every step (0.97 GB forward and again backward at 7680 frames); row b of the batched result carries the same bits
            betas = betas[:1]
        v_shaped, j_rest, batched = core.rest_shape(betas.contiguous(), None)
        v_shaped, j_rest = v_shaped.contiguous(), j_rest.contiguous()
        if batched and v_shaped.shape[0] == 1 and T > 1:
            batched = False                                        # [1, V, 3] / [1, J, 3] read as the shared [V, 3] / [J, 3]
        names = [name for name, _ in core.segments]
        segj = (C.c_int32 * len(names))(*[nj for _, nj in core.segments])
        jptr, jvidx, jw = core.joint_csr()
        u8 = lambda n: torch.empty(int(n), dtype=torch.uint8, device=dev)
        ws_f, ws_b = u8(lib.dposer_lbs_workspace_bytes(h, T)), u8(lib.dposer_lbs_backward_workspace_bytes(h, T))
        rows = core.J + core.n_extra + core.n_lmk
        scratch = u8(lib.dposer_motion_denoise_scratch_bytes(T, D, core.V, rows))
        m, v = torch.zeros_like(pose), torch.zeros_like(pose)
        log = torch.zeros(n_steps, n_seq, 3, dtype=torch.float32, device=dev)
        rot6d = getattr(nz, "rot_rep", "axis") == "rot6d"          # the network sees 6 J coordinates; the optimised pose stays axis-angle
        if rot6d and eng.D != 2 * D:
            raise _C.DPoserHipError(f"rot_rep='rot6d': the score network takes {eng.D} inputs, the pose has {D} axis-angle parameters")
        obs = init_joints.detach().contiguous().float()
        if not nz.normalize:
            mode, na, nb = 0, None, None
        elif nz.min_max:
            mode, (na, nb) = 2, nz._stats(nz.min_poses, nz.max_poses, pose)
        else:
            mode, (na, nb) = 1, nz._stats(nz.mean_poses, nz.std_poses, pose)
        if mode:
            na, nb = na.reshape(-1).contiguous().float(), nb.reshape(-1).contiguous().float()
        nzs = None if noise is None else noise.detach().contiguous().float()
        if nzs is not None and tuple(nzs.shape) != (n_steps, T, eng.D):
            # the C loop indexes the noise as noise + k * T * D_net: [steps, T, 6 J] with rot_rep = 'rot6d', [steps, T, 3 J] otherwise
            raise _C.DPoserHipError(f"noise must be [n_steps, frames, network inputs] = {(n_steps, T, eng.D)}, got {tuple(nzs.shape)}")
        fl = lambda xs: (C.c_float * n_steps)(*[float(x) for x in xs])
        step0 = self._calls + 1
        self._calls += n_steps
        desc = sde_lib.sde_desc(self.sde, bool(getattr(self, "continuous", True)))
        a = _C.MotionDenoiseArgs(
            net=eng.h, flat_params=_C.ptr(flat), packed=_C.ptr(packed), net_ws=_C.ptr(ws), sde=C.pointer(desc), freq=_C.ptr(eng.freq(dev, self.model._fourier_W())),
            sigmas=_C.ptr(model.sigmas), body=h, lbs_ws_fwd=_C.ptr(ws_f), lbs_ws_bwd=_C.ptr(ws_b), posedirs_packed=_C.ptr(core._packed_posedirs()),
            posedirs_bwd_packed=_C.ptr(core._packed_posedirs_bwd()), j_rest=_C.ptr(j_rest), v_shaped=_C.ptr(v_shaped),
            rest_batched=1 if batched else 0, skin_idx=_C.ptr(core.skin_idx), skin_w=_C.ptr(core.skin_w), skin_k=int(core.skin_idx.shape[1]),
            joint_ptr=_C.ptr(jptr), joint_vidx=_C.ptr(jvidx), joint_w=_C.ptr(jw), extra_vertex_ids=_C.ptr(core.extra_vertex_ids),
            lmk_tri=_C.ptr(core.lmk_tri), lmk_bary=_C.ptr(core.lmk_bary_coords), segment_joints_host=segj, num_segments=len(names),
            body_segment=names.index("body_pose"), num_vertices=core.V, num_joints=core.J, joint_rows=rows, frames=T, frames_per_sequence=F,
            pose=_C.ptr(pose),
            adam_m=_C.ptr(m), adam_v=_C.ptr(v), joints_obs=_C.ptr(obs), n_obs_joints=int(obs.shape[1]), norm_mode=mode, norm_a=_C.ptr(na),
            norm_b=_C.ptr(nb), n_steps=n_steps, weighted=0, t_host=fl(t_list), w_temp_host=fl(weights["temp"](1.0, it) for it in its),
            w_data_host=fl(weights["data"](1.0, it) for it in its), w_prior_host=fl(weights["dposer"](1.0, it) for it in its),
            lr=0.03, beta1=0.9, beta2=0.999, eps=1e-8, adam_step0=0, step0=int(step0) & 0xFFFFFFFF, seed=int(model._rng_seed + 31),
            noise=_C.ptr(nzs), scratch=_C.ptr(scratch), loss_log=_C.ptr(log), rot6d=1 if rot6d else 0)
        _C.check(lib.dposer_motion_denoise_optimize(C.byref(a), _C.stream_ptr()), "dposer_motion_denoise_optimize")
        return log

    def _quan_t(self, time_strategy, step, total_steps, sample_trun, sample_time):
        if time_strategy == "1":
            return int(torch.randint(self.sde.N, [1]))
        if time_strategy == "2":
            return int(sample_time)
        if time_strategy == "3":
            return int(self.sde.N - math.floor(float(np.float32(total_steps - step - 1) * np.float32(self.sde.N / (sample_trun * total_steps)))) - 2)
        raise NotImplementedError("unsupported time sampling strategy")

    def optimize_sequences(self, joints3d, gt_poses, time_strategy="3", sample_trun=2.0, sample_time=990, iterations=5, steps_per_iter=50,
                           noise=None, init_poses=None):
        """A BATCH of sequences advanced together by the one-call loop: joints3d [S, F, 22, 3], gt_poses / init_poses [S, F, 63],
        noise [steps, S * F, 63] or None.  Every sequence is the independent problem ``optimize`` solves (same schedule, same loss
        weights; temporal neighbours, data-term decision and loss means per sequence), but the S * F frames share every launch --
        one 60-frame sequence leaves most of an MI355X idle (GPU-bound small kernels), and under data parallelism
        each rank takes its shard of the sequences.  Returns the ``optimize`` metrics as [S, F] arrays and pose_body [S, F, 63]."""
        S, F = joints3d.shape[:2]
        if not self._fused_supported():
            # configurations outside the one-call loop (6D rotations, the VE SDE, a non-positional embedding): every sequence through
            # `optimize`'s step-by-step loop, one after the other -- same results layout, none of the batching
            if F != self.batch_size:
                raise ValueError(f"optimize_sequences: {F} frames per sequence, the module was built for batch_size = {self.batch_size}")
            res = [self.optimize(joints3d[i], gt_poses[i], time_strategy=time_strategy, sample_trun=sample_trun, sample_time=sample_time,
                                 iterations=iterations, steps_per_iter=steps_per_iter,
                                 noise=None if noise is None else noise[:, i * F:(i + 1) * F],
                                 init_poses=None if init_poses is None else init_poses[i], fused=False) for i in range(S)]
            out = {k: np.stack([np.asarray(r[k]) for r in res]) for k in ("init_MPJPE", "MPJPE", "MPVPE")}
            out["pose_body"] = torch.stack([r["pose_body"] for r in res])
            return out
        bm = self.body_model
        flat = lambda x: x.reshape(S * F, *x.shape[2:])
        betas = self.betas[:1].expand(S * F, -1).contiguous()
        with torch.no_grad():
            gt = bm(betas=betas, pose_body=flat(gt_poses))
            je = flat(joints3d) - gt.Jtr[:, :22]
            init_mpjpe = torch.mean(torch.sqrt(torch.sum(je * je, dim=2)), dim=1) * 100.0
        timesteps = torch.linspace(self.sde.T, 1e-3, self.sde.N)
        total_steps = iterations * steps_per_iter
        # (every sequence starts from the module's random initial poses, as `optimize` does: frames must differ -- two identical
        #  neighbouring frames give the temporal term's sqrt a zero argument and a NaN gradient, in the reference too)
        start = self.poses[:F].repeat(S, 1) if init_poses is None else flat(init_poses)
        pose = start.detach().clone().contiguous().float()
        quan = [self._quan_t(time_strategy, step, total_steps, sample_trun, sample_time) for step in range(total_steps)]
        # (the C entry takes at most 65535 frames per call -- one grid row per frame in its skinning kernels; sequences are independent
        #  problems, so a larger batch goes through in groups of whole sequences, each group with its rows of the injected noise)
        per_call = max(1, 65535 // F)
        joints_flat, t_steps = flat(joints3d).detach(), [float(timesteps[q]) for q in quan]
        iters_of_step, weights, logs = [s // steps_per_iter for s in range(total_steps)], self.get_loss_weights(), []
        for s0 in range(0, S, per_call):
            fr = slice(s0 * F, min(S, s0 + per_call) * F)
            logs.append(self._optimize_fused(pose[fr], joints_flat[fr], t_steps, iters_of_step, weights,
                                             None if noise is None else (noise if per_call >= S else noise[:, fr].contiguous()),
                                             frames_per_sequence=F,
                                             betas=self.betas[fr] if (per_call < S and self.betas.shape[0] == S * F) else None))
        self.loss_log = logs[0] if len(logs) == 1 else torch.cat(logs, dim=1)
        with torch.no_grad():
            final = pose.reshape(S, F, -1)
            smooth = torch.stack([gaussian_smoothing(final[i], window_size=3, sigma=2) for i in range(S)])
            smooth[:, [0, -1]] = final[:, [0, -1]]
            out = bm(betas=betas, pose_body=flat(smooth))
            je = out.Jtr[:, :22] - gt.Jtr[:, :22]
            ve = out.v - gt.v
            mpjpe = torch.mean(torch.sqrt(torch.sum(je * je, dim=2)), dim=1) * 100.0
            mpvpe = torch.mean(torch.sqrt(torch.sum(ve * ve, dim=2)), dim=1) * 100.0
        r = lambda x: x.reshape(S, F).cpu().numpy()
        return {"init_MPJPE": r(init_mpjpe), "MPJPE": r(mpjpe), "MPVPE": r(mpvpe), "pose_body": final}

    def optimize(self, joints3d, gt_poses=None, time_strategy="1", sample_trun=2.0, sample_time=990, iterations=5, steps_per_iter=50,
                 verbose=False, vis=False, noise=None, init_poses=None, fused=None):
        """motion_denoising.py:199-300 (visualisation dropped).  Returns {'init_MPJPE', 'MPJPE', 'MPVPE'} in cm per frame.
        ``fused``: None = the one-call loop when the configuration supports it, False = step by step through autograd."""
        bm = self.body_model
        with torch.no_grad():
            gt = bm(betas=self.betas, pose_body=gt_poses)
            joint_error = joints3d - gt.Jtr[:, :22]
            init_mpjpe = torch.mean(torch.sqrt(torch.sum(joint_error * joint_error, dim=2)), dim=1) * 100.0
        init_joints = joints3d.detach()
        weights = self.get_loss_weights()
        timesteps = torch.linspace(self.sde.T, 1e-3, self.sde.N)
        total_steps = iterations * steps_per_iter
        use_fused = self._fused_supported() if fused is None else bool(fused)
        if use_fused:
            if not self._fused_supported():
                raise NotImplementedError("the one-call motion-denoising loop covers axis-angle / rot6d poses and sub-VP / VP SDEs")
            pose = (self.poses if init_poses is None else init_poses).detach().clone().contiguous().float()
            quan = [self._quan_t(time_strategy, step, total_steps, sample_trun, sample_time) for step in range(total_steps)]
            self.loss_log = self._optimize_fused(pose, init_joints, [float(timesteps[q]) for q in quan],
                                                 [s // steps_per_iter for s in range(total_steps)], weights, noise)
        else:
            pose = (self.poses if init_poses is None else init_poses).clone().detach().requires_grad_(True)
            optimizer = torch.optim.Adam([pose], 0.03, betas=(0.9, 0.999))
            for it in range(iterations):
                for i in range(steps_per_iter):
                    step = it * steps_per_iter + i
                    optimizer.zero_grad()
                    poses_n = _normalize_with_grad(self.Normalizer, pose)
                    q = self._quan_t(time_strategy, step, total_steps, sample_trun, sample_time)
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


def evaluate_motion_denoising(md, joints3d, gt_poses, *, sequences_per_call=32, num_replicas=None, rank=None, **optimize_kwargs):
    """The dataset loop of run/motion_denoising.py (its ``__main__`` walks the test sequences one at a time, one process) for a set
    of equal-length sequences, data parallel over SEQUENCES -- the frames of one sequence are coupled by the temporal term, so that
    is the only axis that shards: this rank's contiguous shard (``distributed.shard_bounds``, the reference's
    DistributedEvalSampler arithmetic), ``sequences_per_call`` sequences advanced together by ``optimize_sequences``, per-frame
    metrics kept as device-side sums, ONE all-reduce of (sum, count) pairs at the end (``distributed.reduce_metric_means``).
    ``joints3d`` [S, F, 22, 3] noisy observations, ``gt_poses`` [S, F, 63].  Returns ({'init_MPJPE', 'MPJPE', 'MPVPE'} means in cm over
    every frame of every rank, sequences evaluated here)."""
    from .. import distributed as ddp
    world = ddp.world_size() if num_replicas is None else num_replicas
    rk = ddp.rank() if rank is None else rank
    lo, hi = ddp.shard_bounds(joints3d.shape[0], world, rk)
    results = []
    for s0 in range(lo, hi, sequences_per_call):
        s1 = min(s0 + sequences_per_call, hi)
        res = md.optimize_sequences(joints3d[s0:s1], gt_poses[s0:s1], **optimize_kwargs)
        results.append({k: torch.as_tensor(res[k]) for k in ("init_MPJPE", "MPJPE", "MPVPE")})
    return ddp.reduce_metric_means(results, device=joints3d.device, names=("MPJPE", "MPVPE", "init_MPJPE")), hi - lo

