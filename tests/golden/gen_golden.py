#!/usr/bin/env python3
"""Generate golden input/output vectors by IMPORTING the reference (read-only, /root/reference).

Run in the build container only:   python tests/golden/gen_golden.py
Writes small ``.npz`` fixtures next to this file.  Nothing of the reference's source travels:
only inputs, outputs and seeds (weights are regenerated from ``weights.make_weights(seed)``).

Absent third-party modules the reference imports at module scope are stubbed (MagicMock) --
none of them is exercised by the captured paths except where noted:
  torchgeometry (transforms.py tgm-backed conversions: NOT captured), ml_collections (attr-dict),
  smplx / cv2 / pyrender / ... (import-only).
Randomness inside the reference (torch.rand / torch.randn_like / F.dropout) is replaced by a
recorded numpy RandomState stream so every draw is part of the fixture.
"""
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
REF = "/root/reference"
sys.path.insert(0, REF)

for _name in ["torchgeometry", "cv2", "smplx", "smplx.utils", "smplx.body_models", "tensorboardX",
              "absl", "absl.flags", "absl.app", "absl.flags.argparse_flags",
              "ml_collections.config_flags", "pyrender", "trimesh", "pytorch3d", "plyfile",
              "pymeshlab", "chumpy", "yacs"]:
    sys.modules[_name] = mock.MagicMock()


class _ConfigDict(dict):
    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


_mlc = types.ModuleType("ml_collections")
_mlc.ConfigDict = _ConfigDict
sys.modules["ml_collections"] = _mlc

from weights import make_weights, make_mlp_weights, probe_indices  # noqa: E402

import lib.algorithms.advanced.losses as ref_losses  # noqa: E402
import lib.algorithms.advanced.sampling as ref_sampling  # noqa: E402
import lib.algorithms.advanced.sde_lib as ref_sde  # noqa: E402
import lib.algorithms.advanced.utils as ref_mutils  # noqa: E402
from lib.algorithms.advanced.model import ScoreModelFC, TimeMLPs  # noqa: E402
from lib.algorithms.ema import ExponentialMovingAverage  # noqa: E402
import lib.utils.misc as ref_misc  # noqa: E402
import lib.utils.transforms as ref_tf  # noqa: E402
import lib.dataset.AMASS as ref_amass  # noqa: E402
import lib.body_model.constants as ref_const  # noqa: E402
import lib.body_model.utils as ref_bmu  # noqa: E402
import lib.dataset.EvaSampler as ref_eva  # noqa: E402
import run.completion as ref_completion  # noqa: E402
from configs.subvp.amass_scorefc_continuous import get_config  # noqa: E402

torch.set_num_threads(8)


class Recorder:
    """Replaces torch.rand / torch.randn_like / F.dropout with a recorded numpy stream."""

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.draws = []

    def rand(self, *shape, **kw):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)) else shape
        a = self.rs.random_sample(size=tuple(shape)).astype(np.float32)
        self.draws.append(("rand", a))
        return torch.tensor(a)

    def randn_like(self, x, **kw):
        a = self.rs.standard_normal(size=tuple(x.shape)).astype(np.float32)
        self.draws.append(("randn", a))
        return torch.tensor(a).to(x.dtype)

    def dropout(self, x, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return x
        keep = (self.rs.random_sample(size=tuple(x.shape)) >= p).astype(np.float32)
        self.draws.append(("keep", keep))
        return x * torch.tensor(keep) / (1.0 - p)

    def __enter__(self):
        self._p = [mock.patch.object(torch, "rand", self.rand),
                   mock.patch.object(torch, "randn_like", self.randn_like),
                   mock.patch.object(torch.nn.functional, "dropout", self.dropout)]
        for p in self._p:
            p.start()
        return self

    def __exit__(self, *a):
        for p in self._p:
            p.stop()

    def by_kind(self, kind):
        return [a for k, a in self.draws if k == kind]


def build_model(seed, D, embedding="positional", dropout=0.1):
    cfg = get_config()
    cfg.model.embedding_type = embedding
    cfg.model.dropout = dropout
    pose_dim = D // 21
    m = ScoreModelFC(cfg, n_poses=21, pose_dim=pose_dim, hidden_dim=cfg.model.HIDDEN_DIM,
                     embed_dim=cfg.model.EMBED_DIM, n_blocks=cfg.model.N_BLOCKS)
    w = make_weights(seed, D=D, fourier=(embedding == "fourier"),
                     fourier_scale=cfg.model.fourier_scale)
    sd = m.state_dict()
    for k, v in w.items():
        assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
        sd[k] = v
    m.load_state_dict(sd)
    return cfg, m


def toy_batch(n, rot="axis", seed=42):
    poses = np.load(os.path.join(REF, "examples/toy_data.npz"))["pose_samples"]
    idx = np.random.RandomState(seed).randint(0, poses.shape[0], size=n)
    x = torch.tensor(poses[idx])
    stats = torch.load(os.path.join(REF, "data/AMASS/amass_processed/version1/train/axis_normalize2.pt"))
    return (x - stats["mean_poses"]) / stats["std_poses"], x


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def sample_tensor(name, t):
    idx = probe_indices(name, t.numel())
    flat = t.detach().reshape(-1).double().numpy()
    return np.concatenate([[np.sqrt((flat ** 2).sum())], flat[idx]])


# ---------------------------------------------------------------------------------------------
def g1_forward():
    """G1/G2: ScoreModelFC eval forward + get_score_fn (subVP continuous, VP continuous, VE)."""
    out = {}
    for tag, D, emb, seed in (("axis_pos", 63, "positional", 1), ("rot6d_pos", 126, "positional", 2),
                              ("axis_fourier", 63, "fourier", 3)):
        cfg, m = build_model(seed, D, emb)
        m.eval()
        rs = np.random.RandomState(100 + seed)
        x = torch.tensor(rs.standard_normal((64, D)).astype(np.float32))
        t = np.concatenate([[1e-5, 1e-3, 0.25, 0.5, 1.0], rs.uniform(1e-5, 1, 59)]).astype(np.float32)
        t = torch.tensor(t)
        with torch.no_grad():
            if emb == "positional":
                out[f"{tag}_model"] = m(x, t * 999).numpy()
                for sname, sde in (("subvp", ref_sde.subVPSDE(0.1, 20.0, 1000)),
                                   ("vp", ref_sde.VPSDE(0.1, 20.0, 1000))):
                    fn = ref_mutils.get_score_fn(sde, m, train=False, continuous=True)
                    out[f"{tag}_score_{sname}"] = fn(x, t, None, None).numpy()
            else:
                ve = ref_sde.VESDE(0.01, 50.0, 1000)
                fn = ref_mutils.get_score_fn(ve, m, train=False, continuous=True)
                out[f"{tag}_score_ve"] = fn(x, t, None, None).numpy()
        out[f"{tag}_x"] = x.numpy()
        out[f"{tag}_t"] = t.numpy()
        out[f"{tag}_seed"] = np.int64(seed)
    save("g1_forward", **out)


def g3_loss_grads():
    """G3: DSM loss + all parameter grads, injected (t, z); dropout off and on (injected masks)."""
    out = {}
    for tag, drop in (("nodrop", 0.0), ("drop", 0.1)):
        cfg, m = build_model(5, 63, dropout=drop)
        sde = ref_sde.subVPSDE(0.1, 20.0, 1000)
        batch, _ = toy_batch(32)
        loss_fn = ref_losses.get_sde_loss_fn(sde, train=True, reduce_mean=True, continuous=True)
        with Recorder(7) as rec:
            loss = loss_fn(m, batch, None, None)
        loss.backward()
        out[f"{tag}_loss"] = np.float64(loss.item())
        out[f"{tag}_u"] = rec.by_kind("rand")[0]
        out[f"{tag}_z"] = rec.by_kind("randn")[0]
        if drop > 0:
            out[f"{tag}_keep"] = np.stack(rec.by_kind("keep")).astype(np.uint8)
        for n, p in m.named_parameters():
            out[f"{tag}_grad/{n}"] = (np.zeros(1) if p.grad is None else sample_tensor(n, p.grad))
        out[f"{tag}_batch"] = batch.numpy()
    # likelihood-weighted / sum-reduced variant (losses.py:127-129), loss only
    cfg, m = build_model(5, 63, dropout=0.0)
    sde = ref_sde.subVPSDE(0.1, 20.0, 1000)
    batch, _ = toy_batch(32)
    loss_fn = ref_losses.get_sde_loss_fn(sde, train=True, reduce_mean=False, continuous=True,
                                         likelihood_weighting=True)
    with Recorder(7):
        out["lw_loss"] = np.float64(loss_fn(m, batch, None, None).item())
    out["seed"] = np.int64(5)
    save("g3_loss_grads", **out)


def g4_train_steps():
    """G4: step_fn updates (Adam + warm-up + clip + EMA); steps 0,1,2 then 4999,5000; dropout 0.1
    with injected masks.  Stores loss, lr, and probes of params / Adam state / EMA shadows."""
    cfg, m = build_model(6, 63, dropout=0.1)
    sde = ref_sde.subVPSDE(0.1, 20.0, 1000)
    opt = ref_losses.get_optimizer(cfg, m.parameters())
    ema = ExponentialMovingAverage(m.parameters(), decay=cfg.model.ema_rate)
    state = dict(optimizer=opt, model=m, ema=ema, step=0)
    optimize_fn = ref_losses.optimization_manager(cfg)
    step_fn = ref_losses.get_step_fn(sde, train=True, optimize_fn=optimize_fn, reduce_mean=True,
                                     continuous=True, likelihood_weighting=False)
    batch, _ = toy_batch(32, seed=43)
    out = {"batch": batch.numpy(), "seed": np.int64(6)}
    names = [n for n, _ in m.named_parameters()]
    schedule = [0, 1, 2, 4999, 5000]
    for i, s in enumerate(schedule):
        state["step"] = s
        with Recorder(1000 + i) as rec:
            ld = step_fn(state, batch)
        out[f"s{i}_step"] = np.int64(s)
        out[f"s{i}_loss"] = np.float64(ld["step_loss"].item())
        out[f"s{i}_lr"] = np.float64(opt.param_groups[0]["lr"])
        out[f"s{i}_u"] = rec.by_kind("rand")[0]
        out[f"s{i}_z"] = rec.by_kind("randn")[0]
        out[f"s{i}_keep"] = np.stack(rec.by_kind("keep")).astype(np.uint8)
        for j, (n, p) in enumerate(m.named_parameters()):
            out[f"s{i}_param/{n}"] = sample_tensor(n, p)
            out[f"s{i}_ema/{n}"] = sample_tensor(n, ema.shadow_params[j])
            st = opt.state.get(p, {})
            if "exp_avg" in st:
                out[f"s{i}_m/{n}"] = sample_tensor(n, st["exp_avg"])
                out[f"s{i}_v/{n}"] = sample_tensor(n, st["exp_avg_sq"])
    out["ema_num_updates"] = np.int64(ema.num_updates)
    save("g4_train_steps", **out)


def g5_sampler():
    """G5/G6: pc_sampler trajectories with injected noise: EM only (N=8), EM + completion
    imputation (legs, N=8), EM + Langevin corrector (N=4), and the full N=1000 run (B=8)."""
    cfg, m = build_model(8, 63)
    m.eval()
    out = {"seed": np.int64(8)}

    class Args:
        task = None

    def run(tag, N, B, corrector, task, start_step=0):
        sde = ref_sde.subVPSDE(0.1, 20.0, N)
        cfg.sampling.corrector = corrector
        fn = ref_sampling.get_sampling_fn(cfg, sde, (B, 63), lambda x: x, 1e-3, device="cpu")
        rs = np.random.RandomState(900 + N + B)
        scale = 0.05 if start_step > 900 else 1.0
        z0 = torch.tensor((scale * rs.standard_normal((B, 63))).astype(np.float32))
        obs = mask = None
        args = None
        if task is not None:
            args = Args()
            args.task = task
        if task == "completion":
            poses, _ = toy_batch(B, seed=44)
            with Recorder(55) as r0:
                mask, obs = ref_misc.create_mask(poses, part="legs")
            out[f"{tag}_mask"] = mask.numpy()
            out[f"{tag}_obs"] = obs.numpy()
        with Recorder(77) as rec:
            trajs, x = fn(m, observation=obs, mask=mask, z=z0, start_step=start_step, args=args)
        out[f"{tag}_z0"] = z0.numpy()
        noise = np.stack(rec.by_kind("randn"))
        if noise.shape[0] <= 64:
            out[f"{tag}_noise"] = noise
        else:   # regenerate in the test: RandomState(77).standard_normal((B,63)) per draw, in order
            out[f"{tag}_noise_seed"] = np.int64(77)
            out[f"{tag}_noise_count"] = np.int64(noise.shape[0])
        out[f"{tag}_final"] = x.numpy()
        tr = trajs.numpy()
        out[f"{tag}_trajs"] = tr if tr.shape[0] <= 16 else tr[99::100]
        return tr

    run("em8", 8, 16, "none", None)
    run("comp8", 8, 16, "none", "completion")
    run("lang4", 1000, 16, "langevin", "denoise", start_step=996)
    run("den8", 8, 16, "none", "denoise", start_step=3)
    run("em1000", 1000, 8, "none", None)
    save("g5_sampler", **out)


def g16_guided_step():
    """G16: EulerMaruyamaPredictor.update_fn_guide (sampling.py:191-207, MCG / DPS style guidance: the EM step minus
    grad_step * d||obs * mask - y0_hat * mask|| / d x_t, which differentiates THROUGH the score network w.r.t. its input),
    sub-VP and VP, injected z; legs masked."""
    cfg, m = build_model(16, 63)
    m.eval()
    out = {"seed": np.int64(16)}
    B = 12
    poses, _ = toy_batch(B, seed=46)
    with Recorder(56):
        mask, obs = ref_misc.create_mask(poses, part="legs")
    out["mask"], out["obs"] = mask.numpy(), obs.numpy()
    rs = np.random.RandomState(160)
    x_t = torch.tensor((poses.numpy() + 0.3 * rs.standard_normal((B, 63))).astype(np.float32))
    out["x_t"] = x_t.numpy()
    for name, sde in (("subvp", ref_sde.subVPSDE(0.1, 20.0, 1000)), ("vp", ref_sde.VPSDE(0.1, 20.0, 1000))):
        score_fn = ref_mutils.get_score_fn(sde, m, train=False, continuous=True)
        pred = ref_sampling.EulerMaruyamaPredictor(sde, score_fn, probability_flow=False)
        for t_val in (0.9, 0.3):
            t = torch.ones(B) * t_val
            with Recorder(161) as rec:
                y_hat, y_mean = pred.update_fn_guide(x_t.clone(), t, obs, mask, grad_step=0.7)
            tag = f"{name}_t{int(t_val * 10)}"
            out[f"{tag}_z"] = rec.by_kind("randn")[0]
            out[f"{tag}_y_hat"] = y_hat.detach().numpy()
            out[f"{tag}_y_mean"] = y_mean.detach().numpy()
    save("g16_guided_step", **out)


def g7_prior_loss():
    """G7: DPoserComp.loss (run/completion.py:131-149) value + autograd grad wrt x_0 at the
    quan_t schedule of completion.py:189-190 for steps {0,99,100,199}; injected z."""
    cfg, m = build_model(9, 63)
    m.eval()
    sde = ref_sde.subVPSDE(0.1, 20.0, 1000)
    B = 16
    comp = ref_completion.DPoserComp(m, sde, continuous=True, batch_size=B)
    x0, _ = toy_batch(B, seed=45)
    timesteps = torch.linspace(sde.T, 1e-3, sde.N)
    out = {"x0": x0.numpy(), "seed": np.int64(9)}
    import math
    total = 200
    for step in (0, 99, 100, 199):
        quan_t = sde.N - math.floor(torch.tensor(total - step - 1) * (sde.N / (5.0 * total))) - 2
        t = timesteps[quan_t]
        vec_t = torch.ones(B) * t
        xv = x0.clone().requires_grad_(True)
        with Recorder(300 + step) as rec:
            loss = comp.loss(xv, vec_t, quan_t)          # quan_t lands in `weighted` (quirk)
        loss.backward()
        out[f"s{step}_quan_t"] = np.int64(quan_t)
        out[f"s{step}_t"] = np.float32(t.item())
        out[f"s{step}_z"] = rec.by_kind("randn")[0]
        out[f"s{step}_loss"] = np.float64(loss.item())
        out[f"s{step}_grad"] = xv.grad.numpy()
        with torch.no_grad(), Recorder(300 + step):
            out[f"s{step}_loss_unweighted"] = np.float64(comp.loss(x0, vec_t, False).item())
    save("g7_prior_loss", **out)


def _stub_finder():
    """run/motion_denoising.py imports the rendering stack at module scope (pytorch3d.renderer, ...): any module below these
    packages resolves to a MagicMock.  None of it is called by the captured loop."""
    import importlib.abc
    import importlib.machinery

    class StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        PREFIX = ("pytorch3d", "pyrender", "trimesh", "cv2", "smplx", "imageio", "matplotlib", "tqdm", "open3d")

        def find_spec(self, name, path, target=None):
            return importlib.machinery.ModuleSpec(name, self) if name.split(".")[0] in self.PREFIX else None

        def create_module(self, spec):
            m = mock.MagicMock()
            m.__path__, m.__name__, m.__spec__ = [], spec.name, spec
            return m

        def exec_module(self, module):
            pass

    for k in list(sys.modules):
        if k.split(".")[0] in StubFinder.PREFIX:
            del sys.modules[k]
    sys.meta_path.insert(0, StubFinder())


def g14_completion_loop():
    """G14: the reference's own DPoserComp.optimize (run/completion.py:167-207): B = 16, 2 x 4 steps, part = legs, time
    strategy '3', every z of completion.py:133 recorded."""
    cfg, m = build_model(61, 63)
    m.eval()
    sde = ref_sde.subVPSDE(0.1, 20.0, 1000)
    B, iters, spi = 16, 2, 4
    _, raw = toy_batch(B, seed=47)
    nz = ref_amass.Posenormalizer(os.path.join(REF, "data/AMASS/amass_processed/version1/train"), device="cpu", normalize=True,
                                  min_max=False, rot_rep="axis")
    poses = nz.offline_normalize(raw)
    torch.manual_seed(0)
    mask, obs = ref_misc.create_mask(poses, part="legs")
    comp = ref_completion.DPoserComp(m, sde, continuous=True, batch_size=B)
    with Recorder(1400) as rec:
        out = comp.optimize(obs, mask, iterations=iters, steps_per_iter=spi)
    noise = np.stack(rec.by_kind("randn"))
    assert noise.shape == (iters * spi, B, 63)
    save("g14_completion_loop", seed=np.int64(61), poses=poses.numpy(), observation=obs.numpy(), mask=mask.numpy(), noise=noise,
         out=out.detach().numpy(), iterations=np.int64(iters), steps_per_iter=np.int64(spi))


def g15_motion_denoise_loop():
    """G15: the reference's own MotionDenoise.optimize (run/motion_denoising.py:199-300) driving a torch body model that stands in
    for smplx (oracle.fk_torch on the synthetic SMPL-X-shaped asset): pins the LOOP -- loss weights, time schedule '3', the
    `data_term > 0` guard, Adam, Gaussian smoothing, metrics.  Case 'a': noisy joints.  Case 'b': the observed joints are exactly
    the joints of the initial pose, so the guard drops the data term at step 0."""
    _stub_finder()
    import run.motion_denoising as ref_md
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle import fk_torch
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    asset = make_synthetic_smplx_asset(seed=0)
    cfg, m = build_model(63, 63)
    m.eval()

    class TorchBM:
        def __init__(self):
            self.calls = []

        def __call__(self, betas=None, pose_body=None, **kw):
            v, j = fk_torch.smplx_forward(asset, pose_body.double())
            self.calls.append(pose_body.detach().clone())
            return types.SimpleNamespace(v=v, Jtr=j, f=None, betas=betas, pose_body=pose_body)

    class Args:
        device = "cpu"
        dataset_folder = os.path.join(REF, "data/AMASS/amass_processed")
        version = "version1"

    T, iters, spi, N = 12, 2, 3, 500
    _, raw = toy_batch(T, seed=48)
    gt = raw.numpy().astype(np.float32)
    rs = np.random.RandomState(7)
    init = (gt + rs.standard_normal(gt.shape) * 0.05).astype(np.float32)
    with torch.no_grad():
        _, jgt = fk_torch.smplx_forward(asset, torch.tensor(gt).double())
        _, jinit = fk_torch.smplx_forward(asset, torch.tensor(init).double())
    out = {"gt": gt, "init": init, "T": np.int64(T), "iterations": np.int64(iters), "steps_per_iter": np.int64(spi), "sde_N": np.int64(N),
           "seed": np.int64(63)}
    for tag, joints3d in (("a", jgt[:, :22] + torch.tensor(rs.standard_normal((T, 22, 3)) * 0.04)), ("b", jinit[:, :22].clone())):
        bm = TorchBM()
        md = ref_md.MotionDenoise(cfg, Args(), m, bm, sde_N=N, batch_size=T)
        md.poses = torch.tensor(init)
        with Recorder(1500 + ord(tag)) as rec:
            res = md.optimize(joints3d, gt_poses=torch.tensor(gt), time_strategy="3", iterations=iters, steps_per_iter=spi)
        out[f"{tag}_joints3d"] = joints3d.numpy()
        out[f"{tag}_noise"] = np.stack(rec.by_kind("randn"))
        out[f"{tag}_pose_final"] = md.poses.detach().numpy()      # the optimised leaf (Adam updates it in place)
        out[f"{tag}_pose_before_last_step"] = bm.calls[-2].numpy()
        out[f"{tag}_pose_smooth"] = bm.calls[-1].numpy()
        for k, v in res.items():
            out[f"{tag}_{k}"] = np.asarray(v)
    save("g15_motion_denoise_loop", **out)



def g17_aux_loss():
    """G17: the reference's own step_fn with auxiliary_loss=True (losses.py:91-119 multi-step denoise, :242-258 v2v / j2j body
    terms) driving a torch body model that stands in for smplx (oracle.fk_torch on the synthetic SMPL-X-shaped asset, as in
    G15) and the z-score de-normaliser of the shipped statistics.  dropout = 0 (a kernel cannot reproduce torch's masks),
    injected (t, z); full learning rate (step 5000).  Stores the four loss values, probes of every clipped gradient and of every
    updated parameter."""
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle import fk_torch
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    asset = make_synthetic_smplx_asset(seed=0)
    cfg, m = build_model(17, 63, dropout=0.0)
    sde = ref_sde.subVPSDE(0.1, 20.0, 1000)
    stats = torch.load(os.path.join(REF, "data/AMASS/amass_processed/version1/train/axis_normalize2.pt"))
    mean, std = stats["mean_poses"].float(), stats["std_poses"].float()

    class TorchBM:
        def __call__(self, pose_body=None, **kw):
            v, j = fk_torch.smplx_forward(asset, pose_body.double())
            return types.SimpleNamespace(v=v.float(), Jtr=j.float())

    B, steps = 8, 3
    batch, _ = toy_batch(B, seed=49)
    opt = ref_losses.get_optimizer(cfg, m.parameters())
    ema = ExponentialMovingAverage(m.parameters(), decay=cfg.model.ema_rate)
    state = dict(optimizer=opt, model=m, ema=ema, step=5000)
    step_fn = ref_losses.get_step_fn(sde, train=True, optimize_fn=ref_losses.optimization_manager(cfg), reduce_mean=True, continuous=True,
                                     likelihood_weighting=False, auxiliary_loss=True, denormalize=lambda x: x * std + mean,
                                     body_model=TorchBM(), rot_rep="axis", denoise_steps=steps)
    with Recorder(1700) as rec:
        ld = step_fn(state, batch)
    out = {"batch": batch.numpy(), "seed": np.int64(17), "denoise_steps": np.int64(steps), "mean": mean.numpy(), "std": std.numpy(),
           "u": rec.by_kind("rand")[0], "z": rec.by_kind("randn")[0], "step": np.int64(5000), "lr": np.float64(opt.param_groups[0]["lr"])}
    for k, v in ld.items():
        out[k] = np.float64(v.item())
    for n, p in m.named_parameters():
        out[f"grad/{n}"] = np.zeros(1) if p.grad is None else sample_tensor(n, p.grad)
        out[f"param/{n}"] = sample_tensor(n, p)
    save("g17_aux_loss", **out)


def g18_evaler():
    """G18: the reference's own completion evaluator (lib/dataset/AMASS.py:263-316: part vertex / joint index sets, MPVPE / MPJPE in
    mm, minimum over hypotheses) with oracle.fk_torch on the synthetic asset standing in for smplx.  Pins the index arithmetic and
    the reductions, for every body part and for part=None."""
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    from oracle import fk_torch
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    asset = make_synthetic_smplx_asset(seed=0)

    class TorchBM:
        def __call__(self, pose_body=None, **kw):
            v, j = fk_torch.smplx_forward(asset, pose_body.double())
            return types.SimpleNamespace(v=v.float(), Jtr=j.float())

    B, H = 6, 3
    _, raw = toy_batch(B, seed=50)
    rs = np.random.RandomState(18)
    outs = (raw.numpy()[:, None, :] + rs.standard_normal((B, H, 63)) * 0.1).astype(np.float32)
    out = {"gts": raw.numpy().astype(np.float32), "outs": outs}
    for part in ("left_leg", "right_leg", "left_arm", "right_arm", "trunk", "hands", "legs", "arms", None):
        ev = ref_amass.Evaler(TorchBM(), part=part)
        r = ev.multi_eval_bodys(torch.tensor(outs), torch.tensor(out["gts"]))
        r1 = ev.eval_bodys(torch.tensor(outs[:, 0]), torch.tensor(out["gts"]))
        tag = part or "all"
        out[f"{tag}/mpvpe_all"], out[f"{tag}/mpjpe_body"] = np.asarray(r["mpvpe_all"], np.float64), np.asarray(r["mpjpe_body"], np.float64)
        out[f"{tag}/h0_mpvpe_all"], out[f"{tag}/h0_mpjpe_body"] = np.asarray(r1["mpvpe_all"], np.float64), np.asarray(r1["mpjpe_body"], np.float64)
    save("g18_evaler", **out)


def g19_activations():
    """G19: ScoreModelFC forward + DSM loss and gradient probes for config.model.nonlinearity in {elu, relu, lrelu} (model.py:54-66),
    dropout off, injected (t, z)."""
    out = {"seed": np.int64(19)}
    batch, _ = toy_batch(24, seed=51)
    out["batch"] = batch.numpy()
    sde = ref_sde.subVPSDE(0.1, 20.0, 1000)
    for act in ("elu", "relu", "lrelu"):
        cfg = get_config()
        cfg.model.nonlinearity = act
        cfg.model.dropout = 0.0
        m = ScoreModelFC(cfg, n_poses=21, pose_dim=3, hidden_dim=cfg.model.HIDDEN_DIM, embed_dim=cfg.model.EMBED_DIM, n_blocks=cfg.model.N_BLOCKS)
        sd = m.state_dict()
        for k, v in make_weights(19, D=63).items():
            sd[k] = v
        m.load_state_dict(sd)
        m.eval()
        rs = np.random.RandomState(190)
        t = torch.tensor(rs.uniform(1e-3, 1.0, 24).astype(np.float32))
        out["t"] = t.numpy()
        with torch.no_grad():
            out[f"{act}_model"] = m(batch, t * 999).numpy()
        m.train()
        loss_fn = ref_losses.get_sde_loss_fn(sde, train=True, reduce_mean=True, continuous=True)
        with Recorder(191) as rec:
            loss = loss_fn(m, batch, None, None)
        loss.backward()
        out[f"{act}_loss"] = np.float64(loss.item())
        out["u"], out["z"] = rec.by_kind("rand")[0], rec.by_kind("randn")[0]
        for n, p in m.named_parameters():
            out[f"{act}_grad/{n}"] = np.zeros(1) if p.grad is None else sample_tensor(n, p.grad)
    save("g19_activations", **out)


def g20_fourier_paths():
    """G20: the reference's own training loss / gradients, EM sampler (plain and with completion imputation), prior loss and
    completion loop with `config.model.embedding_type = 'fourier'` (model.py:117-118,152-155: GaussianFourierProjection of
    log(labels), output divided by the labels themselves) -- the shipped config's documented alternative
    (configs/subvp/amass_scorefc_continuous.py:45).  Same recording scheme as g3 / g5 / g7 / g14."""
    out = {"seed": np.int64(23)}
    # DSM loss + gradients (g3 scheme, dropout off)
    cfg, m = build_model(23, 63, embedding="fourier", dropout=0.0)
    sde = ref_sde.subVPSDE(0.1, 20.0, 1000)
    batch, _ = toy_batch(32)
    loss_fn = ref_losses.get_sde_loss_fn(sde, train=True, reduce_mean=True, continuous=True)
    with Recorder(7) as rec:
        loss = loss_fn(m, batch, None, None)
    loss.backward()
    out["dsm_loss"] = np.float64(loss.item())
    out["dsm_u"] = rec.by_kind("rand")[0]
    out["dsm_z"] = rec.by_kind("randn")[0]
    out["dsm_batch"] = batch.numpy()
    for n, p in m.named_parameters():
        out[f"dsm_grad/{n}"] = (np.zeros(1) if p.grad is None else sample_tensor(n, p.grad))
    # EM sampler, N = 8 (g5 scheme)
    cfg, m = build_model(23, 63, embedding="fourier")
    m.eval()

    class Args:
        task = None

    for tag, task in (("em8", None), ("comp8", "completion")):
        N, B = 8, 16
        sde = ref_sde.subVPSDE(0.1, 20.0, N)
        cfg.sampling.corrector = "none"
        fn = ref_sampling.get_sampling_fn(cfg, sde, (B, 63), lambda x: x, 1e-3, device="cpu")
        z0 = torch.tensor(np.random.RandomState(950 + len(tag)).standard_normal((B, 63)).astype(np.float32))
        obs = mask = args = None
        if task is not None:
            args = Args()
            args.task = task
            poses, _ = toy_batch(B, seed=44)
            with Recorder(55):
                mask, obs = ref_misc.create_mask(poses, part="legs")
            out[f"{tag}_mask"] = mask.numpy()
            out[f"{tag}_obs"] = obs.numpy()
        with Recorder(77) as rec:
            trajs, x = fn(m, observation=obs, mask=mask, z=z0, start_step=0, args=args)
        out[f"{tag}_z0"] = z0.numpy()
        out[f"{tag}_noise"] = np.stack(rec.by_kind("randn"))
        out[f"{tag}_final"] = x.numpy()
        out[f"{tag}_trajs"] = trajs.numpy()
    # prior loss (g7 scheme), two steps of the completion schedule
    sde = ref_sde.subVPSDE(0.1, 20.0, 1000)
    B = 16
    comp = ref_completion.DPoserComp(m, sde, continuous=True, batch_size=B)
    x0, _ = toy_batch(B, seed=45)
    timesteps = torch.linspace(sde.T, 1e-3, sde.N)
    out["prior_x0"] = x0.numpy()
    import math
    total = 200
    for step in (0, 199):
        quan_t = sde.N - math.floor(torch.tensor(total - step - 1) * (sde.N / (5.0 * total))) - 2
        t = timesteps[quan_t]
        vec_t = torch.ones(B) * t
        xv = x0.clone().requires_grad_(True)
        with Recorder(300 + step) as rec:
            loss = comp.loss(xv, vec_t, quan_t)
        loss.backward()
        out[f"prior_s{step}_quan_t"] = np.int64(quan_t)
        out[f"prior_s{step}_t"] = np.float32(t.item())
        out[f"prior_s{step}_z"] = rec.by_kind("randn")[0]
        out[f"prior_s{step}_loss"] = np.float64(loss.item())
        out[f"prior_s{step}_grad"] = xv.grad.numpy()
    # completion loop (g14 scheme)
    iters, spi = 2, 4
    _, raw = toy_batch(B, seed=47)
    nz = ref_amass.Posenormalizer(os.path.join(REF, "data/AMASS/amass_processed/version1/train"), device="cpu", normalize=True,
                                  min_max=False, rot_rep="axis")
    poses = nz.offline_normalize(raw)
    torch.manual_seed(0)
    mask, obs = ref_misc.create_mask(poses, part="legs")
    comp = ref_completion.DPoserComp(m, sde, continuous=True, batch_size=B)
    with Recorder(1400) as rec:
        res = comp.optimize(obs, mask, iterations=iters, steps_per_iter=spi)
    out["loop_observation"] = obs.numpy()
    out["loop_mask"] = mask.numpy()
    out["loop_noise"] = np.stack(rec.by_kind("randn"))
    out["loop_out"] = res.detach().numpy()
    out["loop_iterations"] = np.int64(iters)
    out["loop_steps_per_iter"] = np.int64(spi)
    save("g20_fourier_paths", **out)


def g21_ve_paths():
    """G21: the reference's own outputs under the VARIANCE-EXPLODING SDE (sde_lib.py:234-292; continuous score function utils.py:164-181:
    the network is conditioned on sigma(t) and its output IS the score) on the paths this repository fuses: DSM loss + gradients, EM
    sampler (plain and with completion imputation), prior loss and the completion loop.  Recording scheme of g20 (positional embedding)."""
    out = {"seed": np.int64(29), "sigma_min": np.float64(0.01), "sigma_max": np.float64(50.0)}
    mk = lambda N: ref_sde.VESDE(sigma_min=0.01, sigma_max=50.0, N=N)
    cfg, m = build_model(29, 63, dropout=0.0)
    sde = mk(1000)
    batch, _ = toy_batch(32)
    loss_fn = ref_losses.get_sde_loss_fn(sde, train=True, reduce_mean=True, continuous=True)
    with Recorder(7) as rec:
        loss = loss_fn(m, batch, None, None)
    loss.backward()
    out["dsm_loss"] = np.float64(loss.item())
    out["dsm_u"] = rec.by_kind("rand")[0]
    out["dsm_z"] = rec.by_kind("randn")[0]
    out["dsm_batch"] = batch.numpy()
    for n, p in m.named_parameters():
        out[f"dsm_grad/{n}"] = (np.zeros(1) if p.grad is None else sample_tensor(n, p.grad))
    cfg, m = build_model(29, 63)
    m.eval()

    class Args:
        task = None

    for tag, task in (("em8", None), ("comp8", "completion")):
        N, B = 8, 16
        sde = mk(N)
        cfg.sampling.corrector = "none"
        fn = ref_sampling.get_sampling_fn(cfg, sde, (B, 63), lambda x: x, 1e-3, device="cpu")
        z0 = torch.tensor((np.random.RandomState(960 + len(tag)).standard_normal((B, 63)) * 50.0).astype(np.float32))      # prior_sampling: N(0, sigma_max^2)
        obs = mask = args = None
        if task is not None:
            args = Args()
            args.task = task
            poses, _ = toy_batch(B, seed=44)
            with Recorder(55):
                mask, obs = ref_misc.create_mask(poses, part="legs")
            out[f"{tag}_mask"] = mask.numpy()
            out[f"{tag}_obs"] = obs.numpy()
        with Recorder(77) as rec:
            trajs, x = fn(m, observation=obs, mask=mask, z=z0, start_step=0, args=args)
        out[f"{tag}_z0"] = z0.numpy()
        out[f"{tag}_noise"] = np.stack(rec.by_kind("randn"))
        out[f"{tag}_final"] = x.numpy()
        out[f"{tag}_trajs"] = trajs.numpy()
    sde = mk(1000)
    B = 16
    comp = ref_completion.DPoserComp(m, sde, continuous=True, batch_size=B)
    x0, _ = toy_batch(B, seed=45)
    timesteps = torch.linspace(sde.T, 1e-3, sde.N)
    out["prior_x0"] = x0.numpy()
    import math
    total = 200
    for step in (0, 199):
        quan_t = sde.N - math.floor(torch.tensor(total - step - 1) * (sde.N / (5.0 * total))) - 2
        t = timesteps[quan_t]
        vec_t = torch.ones(B) * t
        xv = x0.clone().requires_grad_(True)
        with Recorder(300 + step) as rec:
            loss = comp.loss(xv, vec_t, quan_t)
        loss.backward()
        out[f"prior_s{step}_quan_t"] = np.int64(quan_t)
        out[f"prior_s{step}_t"] = np.float32(t.item())
        out[f"prior_s{step}_z"] = rec.by_kind("randn")[0]
        out[f"prior_s{step}_loss"] = np.float64(loss.item())
        out[f"prior_s{step}_grad"] = xv.grad.numpy()
    iters, spi = 2, 4
    _, raw = toy_batch(B, seed=47)
    nz = ref_amass.Posenormalizer(os.path.join(REF, "data/AMASS/amass_processed/version1/train"), device="cpu", normalize=True,
                                  min_max=False, rot_rep="axis")
    poses = nz.offline_normalize(raw)
    torch.manual_seed(0)
    mask, obs = ref_misc.create_mask(poses, part="legs")
    comp = ref_completion.DPoserComp(m, sde, continuous=True, batch_size=B)
    with Recorder(1400) as rec:
        res = comp.optimize(obs, mask, iterations=iters, steps_per_iter=spi)
    out["loop_observation"] = obs.numpy()
    out["loop_mask"] = mask.numpy()
    out["loop_noise"] = np.stack(rec.by_kind("randn"))
    out["loop_out"] = res.detach().numpy()
    out["loop_iterations"] = np.int64(iters)
    out["loop_steps_per_iter"] = np.int64(spi)
    save("g21_ve_paths", **out)


def g25_ve_discrete_paths():
    """G25: the reference's own outputs under the VE SDE with the DISCRETE score function (get_score_fn(..., continuous=False), utils.py:175-181:
    the network is conditioned on the label round((T - t)(N - 1)); config.training.continuous = False) on the shared-t paths this repository
    fuses: EM sampler (plain and with completion imputation), prior loss and the completion loop.  Recording scheme of g21."""
    out = {"seed": np.int64(31), "sigma_min": np.float64(0.01), "sigma_max": np.float64(50.0)}
    mk = lambda N: ref_sde.VESDE(sigma_min=0.01, sigma_max=50.0, N=N)
    cfg, m = build_model(31, 63)
    cfg.training.continuous = False
    m.eval()

    class Args:
        task = None

    for tag, task in (("em8", None), ("comp8", "completion")):
        N, B = 8, 16
        sde = mk(N)
        cfg.sampling.corrector = "none"
        fn = ref_sampling.get_sampling_fn(cfg, sde, (B, 63), lambda x: x, 1e-3, device="cpu")
        z0 = torch.tensor((np.random.RandomState(970 + len(tag)).standard_normal((B, 63)) * 50.0).astype(np.float32))
        obs = mask = args = None
        if task is not None:
            args = Args()
            args.task = task
            poses, _ = toy_batch(B, seed=48)
            with Recorder(56):
                mask, obs = ref_misc.create_mask(poses, part="legs")
            out[f"{tag}_mask"] = mask.numpy()
            out[f"{tag}_obs"] = obs.numpy()
        with Recorder(78) as rec:
            trajs, x = fn(m, observation=obs, mask=mask, z=z0, start_step=0, args=args)
        out[f"{tag}_z0"] = z0.numpy()
        out[f"{tag}_noise"] = np.stack(rec.by_kind("randn"))
        out[f"{tag}_final"] = x.numpy()
        out[f"{tag}_trajs"] = trajs.numpy()
    sde = mk(1000)
    B = 16
    comp = ref_completion.DPoserComp(m, sde, continuous=False, batch_size=B)
    x0, _ = toy_batch(B, seed=49)
    timesteps = torch.linspace(sde.T, 1e-3, sde.N)
    out["prior_x0"] = x0.numpy()
    import math
    total = 200
    for step in (0, 100, 199):
        quan_t = sde.N - math.floor(torch.tensor(total - step - 1) * (sde.N / (5.0 * total))) - 2
        t = timesteps[quan_t]
        vec_t = torch.ones(B) * t
        xv = x0.clone().requires_grad_(True)
        with Recorder(400 + step) as rec:
            loss = comp.loss(xv, vec_t, quan_t)
        loss.backward()
        out[f"prior_s{step}_quan_t"] = np.int64(quan_t)
        out[f"prior_s{step}_t"] = np.float32(t.item())
        out[f"prior_s{step}_z"] = rec.by_kind("randn")[0]
        out[f"prior_s{step}_loss"] = np.float64(loss.item())
        out[f"prior_s{step}_grad"] = xv.grad.numpy()
    iters, spi = 2, 4
    _, raw = toy_batch(B, seed=50)
    nz = ref_amass.Posenormalizer(os.path.join(REF, "data/AMASS/amass_processed/version1/train"), device="cpu", normalize=True,
                                  min_max=False, rot_rep="axis")
    poses = nz.offline_normalize(raw)
    torch.manual_seed(0)
    mask, obs = ref_misc.create_mask(poses, part="legs")
    comp = ref_completion.DPoserComp(m, sde, continuous=False, batch_size=B)
    with Recorder(1500) as rec:
        res = comp.optimize(obs, mask, iterations=iters, steps_per_iter=spi)
    out["loop_observation"] = obs.numpy()
    out["loop_mask"] = mask.numpy()
    out["loop_noise"] = np.stack(rec.by_kind("randn"))
    out["loop_out"] = res.detach().numpy()
    out["loop_iterations"] = np.int64(iters)
    out["loop_steps_per_iter"] = np.int64(spi)
    save("g25_ve_discrete_paths", **out)


def g26_vp_discrete_paths():
    """G26: the reference's own outputs under the VP SDE with the DISCRETE score function (get_score_fn(..., continuous=False), utils.py:157-162:
    label t (N - 1), score = -model / sqrt_1m_alphas_cumprod[label.long()]; config.training.continuous = False) on the shared-t paths this
    repository fuses: EM sampler (plain and with completion imputation), prior loss and the completion loop.  Recording scheme of G21 / G25."""
    out = {"seed": np.int64(33), "beta_min": np.float64(0.1), "beta_max": np.float64(20.0)}
    mk = lambda N: ref_sde.VPSDE(beta_min=0.1, beta_max=20.0, N=N)
    cfg, m = build_model(33, 63)
    cfg.training.continuous = False
    m.eval()

    class Args:
        task = None

    for tag, task in (("em8", None), ("comp8", "completion")):
        N, B = 8, 16
        sde = mk(N)
        cfg.sampling.corrector = "none"
        fn = ref_sampling.get_sampling_fn(cfg, sde, (B, 63), lambda x: x, 1e-3, device="cpu")
        z0 = torch.tensor((np.random.RandomState(980 + len(tag)).standard_normal((B, 63))).astype(np.float32))
        obs = mask = args = None
        if task is not None:
            args = Args()
            args.task = task
            poses, _ = toy_batch(B, seed=48)
            with Recorder(56):
                mask, obs = ref_misc.create_mask(poses, part="legs")
            out[f"{tag}_mask"] = mask.numpy()
            out[f"{tag}_obs"] = obs.numpy()
        with Recorder(78) as rec:
            trajs, x = fn(m, observation=obs, mask=mask, z=z0, start_step=0, args=args)
        out[f"{tag}_z0"] = z0.numpy()
        out[f"{tag}_noise"] = np.stack(rec.by_kind("randn"))
        out[f"{tag}_final"] = x.numpy()
        out[f"{tag}_trajs"] = trajs.numpy()
    sde = mk(1000)
    B = 16
    comp = ref_completion.DPoserComp(m, sde, continuous=False, batch_size=B)
    x0, _ = toy_batch(B, seed=49)
    timesteps = torch.linspace(sde.T, 1e-3, sde.N)
    out["prior_x0"] = x0.numpy()
    import math
    total = 200
    for step in (0, 100, 199):
        quan_t = sde.N - math.floor(torch.tensor(total - step - 1) * (sde.N / (5.0 * total))) - 2
        t = timesteps[quan_t]
        vec_t = torch.ones(B) * t
        xv = x0.clone().requires_grad_(True)
        with Recorder(400 + step) as rec:
            loss = comp.loss(xv, vec_t, quan_t)
        loss.backward()
        out[f"prior_s{step}_quan_t"] = np.int64(quan_t)
        out[f"prior_s{step}_t"] = np.float32(t.item())
        out[f"prior_s{step}_z"] = rec.by_kind("randn")[0]
        out[f"prior_s{step}_loss"] = np.float64(loss.item())
        out[f"prior_s{step}_grad"] = xv.grad.numpy()
    iters, spi = 2, 4
    _, raw = toy_batch(B, seed=50)
    nz = ref_amass.Posenormalizer(os.path.join(REF, "data/AMASS/amass_processed/version1/train"), device="cpu", normalize=True,
                                  min_max=False, rot_rep="axis")
    poses = nz.offline_normalize(raw)
    torch.manual_seed(0)
    mask, obs = ref_misc.create_mask(poses, part="legs")
    comp = ref_completion.DPoserComp(m, sde, continuous=False, batch_size=B)
    with Recorder(1500) as rec:
        res = comp.optimize(obs, mask, iterations=iters, steps_per_iter=spi)
    out["loop_observation"] = obs.numpy()
    out["loop_mask"] = mask.numpy()
    out["loop_noise"] = np.stack(rec.by_kind("randn"))
    out["loop_out"] = res.detach().numpy()
    out["loop_iterations"] = np.int64(iters)
    out["loop_steps_per_iter"] = np.int64(spi)
    save("g26_vp_discrete_paths", **out)


def g8_scalars():
    """G8: marginal_prob / sde / return_alpha_sigma / discretize tables on linspace(1,1e-3,1000)."""
    t = torch.linspace(1.0, 1e-3, 1000)
    x = torch.ones(1000, 1)
    out = {"t": t.numpy()}
    for name, sde in (("subvp", ref_sde.subVPSDE(0.1, 20.0, 1000)), ("vp", ref_sde.VPSDE(0.1, 20.0, 1000)),
                      ("ve", ref_sde.VESDE(0.01, 50.0, 1000))):
        mean, std = sde.marginal_prob(x, t)
        drift, diff = sde.sde(x, t)
        a, s = sde.return_alpha_sigma(t)
        f, G = sde.discretize(x, t)
        out[f"{name}_mean"] = mean.numpy()
        out[f"{name}_std"] = std.numpy()
        out[f"{name}_drift"] = drift.numpy()
        out[f"{name}_diffusion"] = diff.numpy()
        out[f"{name}_alpha"] = a.numpy()
        out[f"{name}_sigma"] = s.numpy()
        out[f"{name}_disc_f"] = f.numpy()
        out[f"{name}_disc_G"] = G.numpy()
        out[f"{name}_prior_logp"] = sde.prior_logp(torch.linspace(-2, 2, 63 * 4).reshape(4, 63)).numpy()
    cfg, m = build_model(1, 63)
    out["sigmas_buffer"] = m.sigmas.numpy()
    out["temb_labels"] = (t * 999).numpy()
    from lib.algorithms.advanced.model import get_timestep_embedding
    out["temb"] = get_timestep_embedding(t[::50] * 999, 512).numpy()
    save("g8_scalars", **out)


def g24_openpose_maps():
    """G24: smpl_to_openpose (lib/body_model/utils.py:68-177) for EVERY argument combination it accepts:
    model_type x use_hands x use_face x use_face_contour x {coco25, coco19} (bit-exact index tables)."""
    out = {}
    for fmt in ("coco25", "coco19"):
        for mt in ("smpl", "smplh", "smplx"):
            for hands in (False, True):
                for face in (False, True):
                    for contour in (False, True):
                        m = ref_bmu.smpl_to_openpose(mt, use_hands=hands, use_face=face, use_face_contour=contour, openpose_format=fmt)
                        out[f"{fmt}/{mt}/{int(hands)}{int(face)}{int(contour)}"] = np.asarray(m).astype(np.int64)
    save("g24_openpose_maps", **out)


def g9_tables():
    """G9: integer index tables (bit-exact): create_mask, BodyPartIndices, BodySegIndices lengths
    + checksums, JOINT_NAMES/JOINT_MAP, smpl.py joint_map, smpl_to_openpose, 22-joint parents,
    DistributedEvalSampler shard indices."""
    out = {}
    parts = ["left_leg", "right_leg", "left_arm", "right_arm", "trunk", "hands", "legs", "arms"]
    for part in parts:
        out[f"part/{part}"] = np.array(getattr(ref_bmu.BodyPartIndices, part), dtype=np.int64)
        seg = np.array(getattr(ref_bmu.BodySegIndices, part), dtype=np.int64)
        out[f"seg/{part}"] = seg
        for rot_n in (3, 6):
            poses = torch.zeros(4, 21 * rot_n)
            mask, _ = ref_misc.create_mask(poses, part=part)
            out[f"mask/{part}/{rot_n}"] = mask[0].numpy().astype(np.uint8)
    out["joint_names"] = np.array(ref_const.JOINT_NAMES)
    out["joint_map"] = np.array([ref_const.JOINT_MAP[n] for n in ref_const.JOINT_NAMES], dtype=np.int64)
    jm = [ref_const.JOINT_MAP[i] for i in ref_const.JOINT_NAMES]
    jm[:25] = [55, 12, 17, 19, 21, 16, 18, 20, 0, 2, 5, 8, 1, 4, 7, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65]
    # ^ the literal list of lib/body_model/smpl.py:55-57 (smpl.py imports smplx at module scope;
    #   it is imported above with smplx stubbed, the list itself is data)
    out["smplx_joint_map"] = np.array(jm, dtype=np.int64)
    for mt in ("smpl", "smplh", "smplx"):
        out[f"openpose/{mt}"] = ref_bmu.smpl_to_openpose(mt).astype(np.int64)
    skel = ref_bmu.get_smpl_skeleton()
    parents = -np.ones(22, dtype=np.int64)
    for a, b in skel:
        parents[b] = a
    out["parents22"] = parents
    out["skeleton"] = np.asarray(skel, dtype=np.int64)
    for perm in ("SMPL_JOINTS_FLIP_PERM", "SMPL_POSE_FLIP_PERM", "J24_FLIP_PERM", "J49_FLIP_PERM",
                 "H36M_TO_J17", "H36M_TO_J14", "J24_TO_J17", "J24_TO_J14"):
        out[f"const/{perm}"] = np.array(getattr(ref_const, perm), dtype=np.int64)

    class DS:
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

    for total, world in ((103, 4), (16, 8), (7, 2), (100, 1)):
        for rank in range(world):
            s = ref_eva.DistributedEvalSampler(DS(total), num_replicas=world, rank=rank, shuffle=False)
            out[f"eva/{total}/{world}/{rank}"] = np.array(list(iter(s)), dtype=np.int64)
    save("g9_tables", **out)


def g10_normalizer():
    """G10: Posenormalizer z-score and min-max round trips on toy_data (axis)."""
    out = {}
    path = os.path.join(REF, "data/AMASS/amass_processed/version1/train")
    _, raw = toy_batch(64, seed=46)
    out["raw"] = raw.numpy()
    for mm in (False, True):
        nz = ref_amass.Posenormalizer(path, device="cpu", normalize=True, min_max=mm, rot_rep="axis")
        n = nz.offline_normalize(raw)
        out[f"norm_minmax{int(mm)}"] = n.numpy()
        out[f"denorm_minmax{int(mm)}"] = nz.offline_denormalize(n).numpy()
        out[f"norm3d_minmax{int(mm)}"] = nz.offline_normalize(raw.reshape(4, 16, 63)).numpy()
    for f in ("axis_normalize1", "axis_normalize2", "rot6d_normalize1", "rot6d_normalize2"):
        d = torch.load(os.path.join(path, f + ".pt"))
        for k, v in d.items():
            if v is not None:
                out[f"stats/{f}/{k}"] = v.numpy()
    out["toy_pose_samples"] = np.load(os.path.join(REF, "examples/toy_data.npz"))["pose_samples"]
    save("g10_normalizer", **out)


def g11_rot6d():
    rs = np.random.RandomState(11)
    x = torch.tensor(rs.standard_normal((257, 6)).astype(np.float32))
    out = {"rot6d": x.numpy(), "rotmat": ref_tf.rot6d_to_mat3x3(x).numpy()}
    ts = torch.linspace(0.1, 0.9, 7)
    out["lin_interp"] = ref_misc.linear_interpolation(ts, ts / 10, 6).numpy()
    data = torch.tensor(rs.standard_normal((60, 63)).astype(np.float32))
    out["smooth_in"] = data.numpy()
    out["smooth_out"] = ref_misc.gaussian_smoothing(data, 5, 2.0).numpy()
    save("g11_rot6d", **out)


def g12_likelihood_ode():
    """Probability-flow ODE: likelihood (likelihood.py:40-113, called as in run/train.py:235) and the black-box ODE
    sampler (sampling.py:471-542), plus the APD metric (lib/utils/metric.py:8-37)."""
    import lib.algorithms.advanced.likelihood as ref_lik
    import lib.utils.metric as ref_metric
    out = {"seed": np.int64(21)}
    cfg, m = build_model(21, 63)
    m.eval()
    sde = ref_sde.subVPSDE(beta_min=0.1, beta_max=20.0, N=1000)
    data, _ = toy_batch(6, seed=3)
    out["data"] = data.numpy()
    rs = np.random.RandomState(77)
    for kind in ("Rademacher", "Gaussian"):
        eps_np = (rs.randint(0, 2, size=data.shape).astype(np.float32) * 2 - 1) if kind == "Rademacher" \
            else rs.standard_normal(size=data.shape).astype(np.float32)
        with mock.patch.object(torch, "randint_like", lambda x, low=0, high=2: torch.tensor((eps_np + 1) / 2)), \
                mock.patch.object(torch, "randn_like", lambda x: torch.tensor(eps_np)):
            fn = ref_lik.get_likelihood_fn(sde, lambda v: v, hutchinson_type=kind, rtol=1e-4, atol=1e-4, eps=1e-4)
            bpd, z, nfe = fn(m, data.clone())
        out[f"lik_{kind}/eps"] = eps_np
        out[f"lik_{kind}/bpd"] = bpd.numpy()
        out[f"lik_{kind}/z"] = z.numpy()
        out[f"lik_{kind}/nfe"] = np.int64(nfe)
        print(kind, "bpd", bpd.numpy(), "nfe", nfe)
    z0 = rs.standard_normal(size=(6, 63)).astype(np.float32)
    out["ode/z"] = z0
    for denoise in (False, True):
        sampler = ref_sampling.get_ode_sampler(sde, (6, 63), lambda v: v, denoise=denoise, rtol=1e-4, atol=1e-4, eps=1e-3, device="cpu")
        with Recorder(5):
            nfe, x = sampler(m, z=torch.tensor(z0))
        out[f"ode/x_denoise{int(denoise)}"] = x.numpy()
        out[f"ode/nfe_denoise{int(denoise)}"] = np.int64(nfe)
        print("ode denoise", denoise, "nfe", nfe, float(x.abs().max()))
    joints = rs.standard_normal(size=(7, 22, 3)).astype(np.float32)
    out["apd/joints"] = joints
    out["apd/value"] = ref_metric.average_pairwise_distance(torch.tensor(joints)).numpy()
    save("g12_likelihood_ode", **out)


def g22_timemlps():
    """G22: the reference's secondary score model TimeMLPs (model.py:69-90; run/train.py:163-170): eval forward, gradients of a fixed
    linear functional w.r.t. every parameter and the input, and the sub-VP DSM loss + gradients with the model in train mode
    (dropout 0: the Dropout modules draw nothing), for three shapes -- the shipped widths (swish, H = 1024, 2 blocks), the constructor's
    default width (H = 64: padded inside the library) on the 6-D representation with leaky ReLU, and one block of ELU at H = 256."""
    out = {}
    for tag, D, H, nb, act, seed in (("swish1024", 63, 1024, 2, "swish", 31), ("lrelu64", 126, 64, 2, "lrelu", 32), ("elu256", 63, 256, 1, "elu", 33)):
        cfg = get_config()
        cfg.model.nonlinearity = act
        cfg.model.dropout = 0.0
        m = TimeMLPs(cfg, n_poses=21, pose_dim=D // 21, hidden_dim=H, n_blocks=nb)
        m.load_state_dict(make_mlp_weights(seed, D, H, nb))
        m.eval()
        rs = np.random.RandomState(seed + 100)
        B = 48
        x = torch.tensor(rs.standard_normal((B, D)).astype(np.float32), requires_grad=True)
        t = torch.tensor((rs.random_sample(B) * 999.0).astype(np.float32))
        c = torch.tensor(rs.standard_normal((B, D)).astype(np.float32))
        y = m(x, t)
        (y * c).sum().backward()
        out[f"{tag}/x"], out[f"{tag}/t"], out[f"{tag}/c"] = x.detach().numpy(), t.numpy(), c.numpy()
        out[f"{tag}/y"] = y.detach().numpy()
        out[f"{tag}/dx"] = x.grad.numpy()
        for n, p in m.named_parameters():
            out[f"{tag}/grad/{n}"] = sample_tensor(n, p.grad)
            p.grad = None
        # DSM training loss through the reference's loss function (losses.py:93-133), model.train() with p = 0
        sde = ref_sde.subVPSDE(beta_min=0.1, beta_max=20.0, N=1000)
        loss_fn = ref_losses.get_sde_loss_fn(sde, train=True, reduce_mean=True, continuous=True)
        batch = torch.tensor(rs.standard_normal((B, D)).astype(np.float32))
        with Recorder(seed + 200) as rec:
            loss = loss_fn(m, batch, None, None)
        loss.backward()
        out[f"{tag}/dsm_batch"] = batch.numpy()
        out[f"{tag}/dsm_u"] = rec.by_kind("rand")[0]
        out[f"{tag}/dsm_z"] = rec.by_kind("randn")[0]
        out[f"{tag}/dsm_loss"] = np.float64(loss.item())
        for n, p in m.named_parameters():
            out[f"{tag}/dsm_grad/{n}"] = sample_tensor(n, p.grad)
    save("g22_timemlps", **out)


def g13_dataset():
    """AMASSDataset (lib/dataset/AMASS.py:12-182) on a temporary root holding the toy poses: sampling, both normalisations
    (statistics computed and written, then re-read), Denormalize."""
    import tempfile
    poses = torch.tensor(np.load(os.path.join(REF, "examples/toy_data.npz"))["pose_samples"])          # [500, 63]
    betas = torch.tensor(np.random.RandomState(3).standard_normal((500, 10)).astype(np.float32))
    out = {"betas": betas.numpy()}
    with tempfile.TemporaryDirectory() as root:
        for sub in ("train", "valid"):
            os.makedirs(os.path.join(root, "v", sub))
            torch.save(poses if sub == "train" else poses[:100] * 0.5, os.path.join(root, "v", sub, "pose_body.pt"))
            torch.save(betas if sub == "train" else betas[:100], os.path.join(root, "v", sub, "betas.pt"))
        for tag, mm in (("minmax", True), ("zscore", False)):
            ds = ref_amass.AMASSDataset(root, version="v", subset="train", sample_interval=3, rot_rep="axis", return_shape=True,
                                        normalize=True, min_max=mm)
            out[f"{tag}/len"] = np.int64(len(ds))
            out[f"{tag}/poses"] = ds.poses.numpy()
            out[f"{tag}/shapes"] = ds.shapes.numpy()
            out[f"{tag}/item7_poses"] = ds[7]["poses"].numpy()
            out[f"{tag}/denorm"] = ds.Denormalize(ds.poses[:5]).numpy()
            dp, dsh = ds.Denormalize(ds.poses[None, :5], ds.shapes[None, :5])
            out[f"{tag}/denorm3_poses"], out[f"{tag}/denorm3_shapes"] = dp.numpy(), dsh.numpy()
            # the valid subset re-reads the statistics the train subset just wrote
            dv = ref_amass.AMASSDataset(root, version="v", subset="valid", sample_interval=None, rot_rep="axis", return_shape=False,
                                        normalize=True, min_max=mm)
            out[f"{tag}/valid_poses"] = dv.poses.numpy()
    save("g13_dataset", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g3", "g4", "g5", "g7", "g8", "g9", "g10", "g11", "g12", "g13", "g14", "g15", "g16", "g17", "g18", "g19", "g20", "g21", "g22", "g24", "g25", "g26"]
    fns = dict(g26=g26_vp_discrete_paths, g25=g25_ve_discrete_paths, g22=g22_timemlps, g21=g21_ve_paths, g20=g20_fourier_paths, g19=g19_activations, g18=g18_evaler, g17=g17_aux_loss, g16=g16_guided_step, g14=g14_completion_loop, g15=g15_motion_denoise_loop, g1=g1_forward, g3=g3_loss_grads, g4=g4_train_steps, g5=g5_sampler, g7=g7_prior_loss,
               g8=g8_scalars, g9=g9_tables, g24=g24_openpose_maps, g10=g10_normalizer, g11=g11_rot6d, g12=g12_likelihood_ode, g13=g13_dataset)
    for w in which:
        fns[w]()
