"""Index ranges: the entry points at batches whose element counts cross 2^31 (byte counts cross 2^32 / 2^33), checked through size-independent
properties -- a row's result does not depend on the batch it sits in; a replicated batch has the loss and the mean gradients of one copy.
(The reference has no such sizes; these are the "maximum sizes" cases of the parity plan.  ~40 GB of HBM at most, a few seconds each.)"""
import pytest
import torch

from gpu_common import make_model, DEV

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def _body():
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    return BodyModel(make_synthetic_smplx_asset(seed=0)).to(DEV)


def test_score_forward_at_two_million_poses():
    """2^21 poses x 1024 channels = 2^31 activation elements per layer."""
    B = 1 << 21
    cfg, m, p = make_model(12, precision="bf16")
    gen = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(B, 63, device=DEV, generator=gen)
    t = torch.rand(B, device=DEV, generator=gen) * 0.999 + 1e-3
    with torch.no_grad():
        big = m(x, t * 999)
        assert torch.isfinite(big).all()
        for s0 in (0, B // 2 + 77, B - 48):
            sl = slice(s0, s0 + 48)
            assert torch.equal(big[sl], m(x[sl].contiguous(), (t[sl] * 999).contiguous())), s0
    del big, x, t
    torch.cuda.empty_cache()


def test_training_gradients_at_half_a_million_poses():
    """65536 samples replicated 8 x (dropout off): the loss is a mean, so loss and gradients are those of one copy up to summation order."""
    from dposer_amd.algorithms.advanced import sde_lib
    from dposer_amd.algorithms.advanced.losses import fused_dsm_grad
    B0, rep = 65536, 8
    cfg, m, p = make_model(5, precision="bf16", dropout=0.0)
    gen = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randn(B0, 63, device=DEV, generator=gen) * 0.5
    t = torch.rand(B0, device=DEV, generator=gen) * (1 - 1e-5) + 1e-5
    z = torch.randn(B0, 63, device=DEV, generator=gen)
    out = []
    for r in (1, rep):
        fg = torch.zeros(m._num_flat, device=DEV)
        loss = fused_dsm_grad(m, sde_lib.subVPSDE(0.1, 20.0, 1000), x.repeat(r, 1), flat_grad=fg, t=t.repeat(r), z=z.repeat(r, 1), seed=m._rng_seed, step=0)
        out.append((float(loss), fg))
    assert abs(out[0][0] - out[1][0]) / abs(out[0][0]) < 1e-5
    assert rel(out[1][1], out[0][1]) < 1e-4
    torch.cuda.empty_cache()


def test_fk_joints_at_eight_million_poses():
    B = 1 << 23
    bm = _body()
    gen = torch.Generator(device=DEV).manual_seed(3)
    pb = torch.randn(B, 63, device=DEV, generator=gen) * 0.3
    with torch.no_grad():
        j = bm.fk_joints(pb)
        assert torch.isfinite(j).all()
        for s0 in (0, B // 2 + 5, B - 40):
            sl = slice(s0, s0 + 40)
            assert rel(j[sl], bm.fk_joints(pb[sl].contiguous())) < 1e-6, s0
    del j, pb
    torch.cuda.empty_cache()


def test_lbs_forward_and_backward_past_two_billion_vertex_coordinates():
    """73728 poses x 31425 vertex coordinates = 2.3e9 elements (9.3 GB per vertex-sized array): vertices, joints and the pose gradient of
    sum |v|^2 / 2 of a row are those of the row by itself."""
    B = 73728
    bm = _body()
    gen = torch.Generator(device=DEV).manual_seed(4)
    pb = (torch.randn(B, 63, device=DEV, generator=gen) * 0.3).requires_grad_(True)
    out = bm(pose_body=pb)
    v = out.v
    assert v.numel() > 2 ** 31
    (v.square().sum() * 0.5).backward()
    g = pb.grad.clone()
    for s0 in (0, B // 2 + 3, B - 24):
        sl = slice(s0, s0 + 24)
        ps = pb.detach()[sl].clone().requires_grad_(True)
        o = bm(pose_body=ps)
        assert rel(v[sl].detach(), o.v.detach()) < 1e-6 and rel(out.Jtr[sl].detach(), o.Jtr.detach()) < 1e-6, s0
        (o.v.square().sum() * 0.5).backward()
        assert rel(g[sl], ps.grad) < 1e-4, s0
    del out, v, g, pb
    torch.cuda.empty_cache()


def test_motion_denoising_loop_past_two_billion_vertex_coordinates():
    """1230 sequences of 60 frames = 73800 frames x 31425 vertex coordinates = 2.3e9 in one `optimize_sequences` call (three optimiser steps; the C entry takes 65535 frames, the
    host mirror hands it groups of whole sequences):
    the first, a middle and the last 16 sequences come out as the same 16 sequences advanced by themselves (sequences are independent
    problems; both sizes take the temporal gradient inside the skinning backward, the big one with five poses per workgroup)."""
    import numpy as np
    from helpers import load
    from dposer_amd.dataset.AMASS import Posenormalizer
    from dposer_amd.tasks.motion_denoising import MotionDenoise
    S, F, steps, sub = 1230, 60, 3, 16
    T = S * F
    cfg, m, p = make_model(63, precision="fp32")
    bm = _body()
    g = load("g10_normalizer")
    stats = {k.split("/")[-1]: torch.tensor(g[k]) for k in g.files if k.startswith("stats/axis_normalize")}
    toy = torch.tensor(g["toy_pose_samples"].astype(np.float32), device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(9)
    gt = toy[torch.arange(T, device=DEV) % toy.shape[0]] + torch.randn(T, 63, device=DEV, generator=gen) * 0.02
    init = gt + torch.randn(T, 63, device=DEV, generator=gen) * 0.05
    with torch.no_grad():
        joints3d = bm.fk_joints(gt) + torch.randn(T, 22, 3, device=DEV, generator=gen) * 0.04
    noise = torch.randn(steps, T, 63, device=DEV, generator=gen)

    class Args:
        device = DEV

    nz = Posenormalizer(stats, device=DEV, normalize=True, min_max=False, rot_rep="axis")
    md = MotionDenoise(cfg, Args(), m, bm, sde_N=500, batch_size=F, normalizer=nz)
    kw = dict(time_strategy="3", iterations=1, steps_per_iter=steps)
    res = md.optimize_sequences(joints3d.reshape(S, F, 22, 3), gt.reshape(S, F, 63), noise=noise, init_poses=init.reshape(S, F, 63), **kw)
    big, log = res["pose_body"].clone(), md.loss_log.clone()
    assert torch.isfinite(big).all() and torch.isfinite(log).all() and log.shape == (steps, S, 3)
    assert rel(big.reshape(T, 63), init) > 1e-4          # the poses moved
    del res
    torch.cuda.empty_cache()
    for s0 in (0, S // 2, S - sub):
        fr = slice(s0 * F, (s0 + sub) * F)
        one = md.optimize_sequences(joints3d[fr].reshape(sub, F, 22, 3), gt[fr].reshape(sub, F, 63), noise=noise[:, fr].contiguous(),
                                    init_poses=init[fr].reshape(sub, F, 63), **kw)
        assert rel(big[s0:s0 + sub], one["pose_body"]) < 1e-5, s0
        assert rel(log[:, s0:s0 + sub, :2], md.loss_log[:, :, :2]) < 1e-5, s0      # (column 2, the prior term, is one number per C call: the sum over its sequences)
    torch.cuda.empty_cache()


def test_em_sampler_at_two_million_samples():
    """dposer_em_sampler at 2^21 samples, four steps with injected noise: the rows of a slice are the rows of that slice sampled by itself."""
    from dposer_amd.algorithms.advanced import sampling, sde_lib
    B, N = 1 << 21, 4
    cfg, m, p = make_model(12, precision="bf16")
    sde = sde_lib.subVPSDE(0.1, 20.0, N)
    cfg.sampling.corrector = "none"
    gen = torch.Generator(device=DEV).manual_seed(6)
    z = torch.randn(B, 63, device=DEV, generator=gen)
    noise = torch.randn(N, 1, B, 63, device=DEV, generator=gen)
    _, x = sampling.get_sampling_fn(cfg, sde, (B, 63), lambda v: v, 1e-3, device=DEV)(m, z=z, noise=noise)
    assert torch.isfinite(x).all()
    for s0 in (0, B // 2 + 130, B - 96):
        sl = slice(s0, s0 + 96)
        _, xs = sampling.get_sampling_fn(cfg, sde, (96, 63), lambda v: v, 1e-3, device=DEV)(m, z=z[sl].contiguous(), noise=noise[:, :, sl].contiguous())
        assert rel(x[sl], xs) < 1e-6, s0
    del x, z, noise
    torch.cuda.empty_cache()
