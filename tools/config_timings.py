#!/usr/bin/env python3
"""Wall-clock of BASELINE.json's non-headline configurations on ONE MI355X (synthetic data, random-init weights):
  cfg2  train step at batch 8192                      cfg3  1000-step EM sampling of 500 poses
  cfg4  pose completion, one rank's share: batch 16384, `--part legs`, 2 x 100 optimisation steps per hypothesis
  cfg5  motion denoising of one 60-frame sequence: 5 x 50 optimisation steps (SMPL-X LBS forward + backward + prior)
  bpd   validation likelihood (bits/dim, run/train.py:279) of 8192 poses: adaptive RK45 and fixed-step RK4
Prints a markdown table (committed as profiles/rNN_configs.md).  Parity of each loop is covered by tests/test_gpu_tasks.py."""
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, warm=1, reps=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else None       # e.g. `cfg5`: that configuration only (for rocprofv3 runs)
    from dposer_amd.algorithms.advanced import losses, sampling, sde_lib
    from dposer_amd.algorithms.advanced.model import ScoreModelFC
    from dposer_amd.algorithms.ema import ExponentialMovingAverage
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    from dposer_amd.configs import load_config
    from dposer_amd.dataset.AMASS import Posenormalizer
    from dposer_amd.tasks.completion import DPoserComp
    from dposer_amd.tasks.motion_denoising import MotionDenoise
    from dposer_amd.utils.misc import create_mask
    dev = "cuda:0"
    cfg = load_config("configs.subvp.amass_scorefc_continuous.get_config")
    torch.manual_seed(42)
    model = ScoreModelFC(cfg, n_poses=21, pose_dim=3, hidden_dim=1024, embed_dim=512, n_blocks=2).to(dev)
    g = np.load(os.path.join(ROOT, "tests", "golden", "g10_normalizer.npz"))
    toy = torch.tensor(g["toy_pose_samples"])
    stats = {k: torch.tensor(g[f"stats/axis_normalize{n}/{k}"]) for n, ks in ((1, ("min_poses", "max_poses")), (2, ("mean_poses", "std_poses")))
             for k in ks}
    norm = Posenormalizer(stats, device=dev, normalize=True, min_max=False, rot_rep="axis")
    rows = []

    sde = sde_lib.subVPSDE(0.1, 20.0, 1000)
    if only in (None, 'cfg2'):
        # cfg2: training step, batch 8192
        state = dict(model=model, optimizer=losses.get_optimizer(cfg, model.parameters()),
                     ema=ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate), step=0)
        step_fn = losses.get_step_fn(sde, True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=True, continuous=True)
        batch = norm.offline_normalize(toy[torch.randint(0, 500, (8192,))].to(dev))
        s = timed(lambda: [step_fn(state, batch) for _ in range(20)], warm=1) / 20
        rows.append(("cfg2", "train step, batch 8192 (axis-angle)", f"{s * 1e3:.3f} ms / step", f"{8192 / s:,.0f} poses/s"))

    model.eval()
    if only in (None, 'cfg3'):
        # cfg3: 500 samples, 1000-step EM sampler
        fn = sampling.get_sampling_fn(cfg, sde, (500, 63), lambda v: v, 1e-3, device=dev)
        s = timed(lambda: fn(model, traj_stride=0), warm=1)
        rows.append(("cfg3", "1000-step EM sampling, 500 poses", f"{s:.3f} s", f"{500 / s:,.0f} samples/s"))

    if only in (None, 'cfg4'):
        # cfg4: completion, batch 16384, legs masked, 2 x 100 steps (one hypothesis)
        poses = norm.offline_normalize(toy[torch.randint(0, 500, (16384,))].to(dev))
        mask, obs = create_mask(poses, part="legs")
        comp = DPoserComp(model, sde, continuous=True, batch_size=16384)
        s = timed(lambda: comp.optimize(obs, mask, iterations=2, steps_per_iter=100), warm=1)
        rows.append(("cfg4", "completion (legs), batch 16384, 200 optimisation steps", f"{s:.3f} s / hypothesis", f"{16384 / s:,.0f} poses/s/hypothesis"))

    if only in (None, 'cfg5'):
        # cfg5: motion denoising, one 60-frame sequence, 5 x 50 steps
        bm = BodyModel(make_synthetic_smplx_asset(seed=0), batch_size=60).to(dev)
        args = types.SimpleNamespace(device=dev, dataset_folder="", version="", task="denoise")
        md = MotionDenoise(cfg, args, model, bm, sde_N=1000, batch_size=60, normalizer=norm)
        gt = toy[:60].to(dev)
        with torch.no_grad():
            joints = bm(pose_body=gt, betas=md.betas).Jtr[:, :22] + 0.04 * torch.randn(60, 22, 3, device=dev)
        s = timed(lambda: md.optimize(joints, gt_poses=gt, iterations=5, steps_per_iter=50), warm=1)
        rows.append(("cfg5", "motion denoising, 60 frames, 250 optimisation steps (LBS fwd+bwd + prior), one C call", f"{s:.3f} s / sequence", f"{s / 250 * 1e3:.2f} ms / step"))
        for S in (() if (len(sys.argv) > 2 and sys.argv[2] == "one") else (8, 32, 128)):      # (`cfg5 one`: the single sequence only -- for kernel traces)
            jb = joints[None].expand(S, -1, -1, -1).contiguous() + 0.01 * torch.randn(S, 60, 22, 3, device=dev)
            gb = gt[None].expand(S, -1, -1).contiguous()
            s = timed(lambda: md.optimize_sequences(jb, gb, time_strategy="1", iterations=5, steps_per_iter=50), warm=1)
            rows.append((f"cfg5 x {S}", f"{S} sequences of 60 frames advanced together (optimize_sequences), 250 steps", f"{s / S:.4f} s / sequence",
                         f"{s / 250 * 1e3:.2f} ms / step, {S / s:,.0f} sequences/s"))
        if only == "cfg5" and len(sys.argv) > 2 and sys.argv[2] in ("fused-only", "one"):
            rows.append(("", "", "", ""))
            s = 0.0
        else:
            s = timed(lambda: md.optimize(joints, gt_poses=gt, iterations=5, steps_per_iter=50, fused=False), warm=1)
        rows.append(("cfg5 (autograd loop)", "the same steps through autograd + torch.optim.Adam around the same kernels", f"{s:.3f} s / sequence", f"{s / 250 * 1e3:.2f} ms / step"))

    if only in (None, 'bpd'):
        # validation bits/dim of run/train.py:279 (likelihood.py:40-113) at batch 8192: adaptive RK45 (the reference's mode; device-
        # resident controller) and the fixed-step RK4 (no host synchronisation), one network evaluation per right-hand side
        from dposer_amd.algorithms.advanced import likelihood
        data = norm.offline_normalize(toy[torch.randint(0, 500, (8192,))].to(dev))
        eps = likelihood.hutchinson_noise(data, "Rademacher")
        res = {}
        for tag, kw in (("RK45 rtol=atol=1e-5 (reference default), device driver", dict(method="RK45")),
                        ("RK45, scipy driver (host state, as the reference)", dict(method="RK45", driver="scipy")),
                        ("fixed-step RK4, 100 steps", dict(method="rk4", n_steps=100)),
                        ("fixed-step RK4, 25 steps", dict(method="rk4", n_steps=25))):
            fn = likelihood.get_likelihood_fn(sde, lambda v: v, **kw)
            out = {}

            def run():
                out["r"] = fn(model, data, epsilon=eps)

            s = timed(run, warm=1)
            bpd, _, nfe = out["r"]
            res[tag] = float(bpd.mean())
            rows.append(("bpd", f"likelihood of 8192 poses, {tag}", f"{s:.3f} s ({nfe} evaluations, {s / nfe * 1e3:.3f} ms each)",
                         f"{8192 / s:,.0f} poses/s, mean bpd {float(bpd.mean()):.4f}"))

    print("| config | workload | time | rate |")
    print("|---|---|---:|---:|")
    for r in rows:
        print("| " + " | ".join(r) + " |")


if __name__ == "__main__":
    main()
