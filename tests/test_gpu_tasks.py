"""GPU tests of the task-loop counterparts and the DPoser prior module: a few optimisation
steps with injected noise against the same loop restated with the CPU oracle."""
import numpy as np
import pytest
import torch

from gpu_common import DEV, make_model, t2n
from helpers import _log_measured, load, rel_err
from oracle import fk_ref, fk_torch
from oracle import score_ref as R
from oracle import task_loops

pytestmark = pytest.mark.gpu


def test_completion_loop_matches_oracle():
    from dposer_amd.algorithms.advanced import sde_lib
    from dposer_amd.tasks.completion import DPoserComp
    from dposer_amd.utils.misc import create_mask
    cfg, m, p = make_model(61, precision="fp32")
    B, iters, spi = 24, 2, 4
    g = load("g10_normalizer")
    poses = torch.tensor(g["norm_minmax0"][:B])
    torch.manual_seed(0)
    mask, obs = create_mask(poses, part="legs")
    noise = np.random.RandomState(1).standard_normal((iters * spi, B, 63)).astype(np.float32)
    sde = sde_lib.subVPSDE(0.1, 20.0, 1000)
    comp = DPoserComp(m, sde, continuous=True, batch_size=B)
    for step in (0, 3, 7):
        assert comp.quan_t(step, iters * spi, 1000) == R.completion_quan_t(step, iters * spi, 1000)
    out = comp.optimize(obs.to(DEV), mask.to(DEV), iterations=iters, steps_per_iter=spi, noise=torch.tensor(noise, device=DEV))
    ref = task_loops.completion_optimize(p, R.SubVP(), obs.numpy(), mask.numpy(), noise, iterations=iters, steps_per_iter=spi)
    assert rel_err(t2n(out), ref) < 2e-4
    assert np.array_equal(t2n(out) * mask.numpy(), obs.numpy() * mask.numpy())      # observed entries untouched


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-6), ("bf16", 2e-4)])      # measured 5.0e-8 / 6.7e-5
def test_completion_loop_matches_the_reference_loop(precision, tol):
    """tasks/completion.DPoserComp.optimize vs the output of the reference's own run/completion.py:167-207 loop (golden g14:
    B = 16, 2 x 4 steps, legs masked, recorded z)."""
    from dposer_amd.algorithms.advanced import sde_lib
    from dposer_amd.tasks.completion import DPoserComp
    g = load("g14_completion_loop")
    cfg, m, p = make_model(int(g["seed"]), precision=precision)
    comp = DPoserComp(m, sde_lib.subVPSDE(0.1, 20.0, 1000), continuous=True, batch_size=g["observation"].shape[0])
    obs, mask = torch.tensor(g["observation"], device=DEV), torch.tensor(g["mask"], device=DEV)
    out = comp.optimize(obs, mask, iterations=int(g["iterations"]), steps_per_iter=int(g["steps_per_iter"]),
                        noise=torch.tensor(g["noise"], device=DEV))
    assert rel_err(t2n(out), g["out"]) < tol
    assert np.array_equal(t2n(out) * g["mask"], g["observation"] * g["mask"])


def test_dposer_module_forward():
    """DPoser(batch_size, config_path, args).forward(poses, betas, quan_t) (run/smplify.py:109-115) vs the oracle."""
    from dposer_amd.dataset.AMASS import Posenormalizer
    from dposer_amd.prior import DPoser
    cfg, m, p = make_model(62, precision="fp32")
    g = load("g10_normalizer")
    stats = {k.split("/")[-1]: torch.tensor(g[k]) for k in g.files if k.startswith("stats/axis_normalize")}
    B = 16

    class Args:
        device = DEV
        sde_N = 500

    nz = Posenormalizer(stats, device=DEV, normalize=True, min_max=False, rot_rep="axis")
    prior = DPoser(batch_size=B, config_path="configs.subvp.amass_scorefc_continuous.get_config", args=Args(), model=m, normalizer=nz)
    raw = torch.tensor(g["raw"][:B])
    full = torch.cat([raw, torch.zeros(B, 6)], dim=1).to(DEV).requires_grad_(True)      # [B, 69] like SMPL body pose
    z = np.random.RandomState(3).standard_normal((B, 63)).astype(np.float32)
    loss = prior(full, None, 250, z=torch.tensor(z, device=DEV))
    loss.backward()
    x0 = (raw - stats["mean_poses"]) / stats["std_poses"]
    t = torch.ones(B) * torch.linspace(1.0, 1e-3, 500)[250]
    ref, gref = R.dposer_prior_loss(p, R.SubVP(N=500), x0, t, torch.tensor(z), weighted=True, reduction="sum_over_batch", batch_size=B)
    assert abs(float(loss) - ref.item()) / abs(ref.item()) < 2e-4
    assert rel_err(t2n(full.grad)[:, :63], (gref / stats["std_poses"]).numpy()) < 2e-4
    assert float(full.grad[:, 63:].abs().max()) == 0.0


def test_motion_denoise_steps_match_oracle():
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    from dposer_amd.dataset.AMASS import Posenormalizer
    from dposer_amd.tasks.motion_denoising import MotionDenoise
    cfg, m, p = make_model(63, precision="fp32")
    asset = make_synthetic_smplx_asset(seed=0)
    bm = BodyModel(asset).to(DEV)
    g = load("g10_normalizer")
    stats = {k.split("/")[-1]: torch.tensor(g[k]) for k in g.files if k.startswith("stats/axis_normalize")}
    T, iters, spi = 12, 1, 3
    gt = g["raw"][:T].astype(np.float32)
    rs = np.random.RandomState(7)
    init = (gt + rs.standard_normal(gt.shape) * 0.05).astype(np.float32)
    _, jgt, _, _ = fk_ref.smplx_forward(asset, gt.astype(np.float64), dtype=np.float64)
    joints3d = (jgt[:, :22] + rs.standard_normal((T, 22, 3)) * 0.04).astype(np.float32)
    noise = rs.standard_normal((iters * spi, T, 63)).astype(np.float32)

    class Args:
        device = DEV

    nz = Posenormalizer(stats, device=DEV, normalize=True, min_max=False, rot_rep="axis")
    md = MotionDenoise(cfg, Args(), m, bm, sde_N=500, batch_size=T, normalizer=nz)
    res = md.optimize(torch.tensor(joints3d, device=DEV), gt_poses=torch.tensor(gt, device=DEV), time_strategy="3", iterations=iters,
                      steps_per_iter=spi, noise=torch.tensor(noise, device=DEV), init_poses=torch.tensor(init, device=DEV))
    final, ref = task_loops.motion_denoise_optimize(p, R.SubVP(N=500), asset, stats["mean_poses"], stats["std_poses"], joints3d, gt, init, noise,
                                                    iterations=iters, steps_per_iter=spi)
    # fp32 kernels against the restatement over 2 x 4 Adam steps; the bound also covers the vertex / joint residual gradients'
    # v_rcp_f32 / v_rsq_f32 (1 ulp each, csrc/tasks.hip: k_md_vert_grad) in place of IEEE divisions -- measured 9e-8
    assert rel_err(t2n(res["pose_body"]), final) < 1e-5
    assert res["MPJPE"].shape == (T,) and np.isfinite(res["MPVPE"]).all()


def _motion_denoise_golden(tag):
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    from dposer_amd.dataset.AMASS import Posenormalizer
    from dposer_amd.tasks.motion_denoising import MotionDenoise
    g = load("g15_motion_denoise_loop")
    cfg, m, p = make_model(int(g["seed"]), precision="fp32")
    bm = BodyModel(make_synthetic_smplx_asset(seed=0)).to(DEV)
    st = load("g10_normalizer")
    stats = {k.split("/")[-1]: torch.tensor(st[k]) for k in st.files if k.startswith("stats/axis_normalize")}

    class Args:
        device = DEV

    T = int(g["T"])
    nz = Posenormalizer(stats, device=DEV, normalize=True, min_max=False, rot_rep="axis")
    md = MotionDenoise(cfg, Args(), m, bm, sde_N=int(g["sde_N"]), batch_size=T, normalizer=nz)
    dev = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=DEV)
    res = md.optimize(dev(g[f"{tag}_joints3d"]), gt_poses=dev(g["gt"]), time_strategy="3", iterations=int(g["iterations"]),
                      steps_per_iter=int(g["steps_per_iter"]), noise=dev(g[f"{tag}_noise"]), init_poses=dev(g["init"]))
    return g, res


def test_motion_denoise_loop_matches_the_reference_loop():
    """tasks/motion_denoising.MotionDenoise.optimize (HIP LBS forward + backward, HIP prior loss) vs the reference's own
    run/motion_denoising.py:199-300 loop around a torch body model on the same synthetic asset (golden g15, case a)."""
    g, res = _motion_denoise_golden("a")
    assert rel_err(t2n(res["pose_body"]), g["a_pose_final"]) < 5e-6              # measured 1.4e-7
    for k in ("init_MPJPE", "MPJPE", "MPVPE"):
        assert np.allclose(res[k], g[f"a_{k}"], rtol=2e-3, atol=1e-3), k


def test_motion_denoise_zero_data_residual_stays_finite():
    """Case b of g15: the observation is the joint set of the initial pose.  The reference's fp64 stand-in gets an exactly zero
    data term and drops it (`if data_term > 0`); the fp32 kernels see rounding-level residuals instead -- the loop must stay
    finite (sqrt'(0) guarded without a host sync) and end near the reference's result."""
    g, res = _motion_denoise_golden("b")
    assert np.isfinite(t2n(res["pose_body"])).all() and np.isfinite(res["MPJPE"]).all()
    assert np.abs(t2n(res["pose_body"]) - g["b_pose_final"]).max() < 0.2          # Adam lr 0.03 x 6 steps bounds the drift


def _md_setup(T=12, min_max=False, seed=7, embedding="positional"):
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    from dposer_amd.dataset.AMASS import Posenormalizer
    from dposer_amd.tasks.motion_denoising import MotionDenoise
    cfg, m, p = make_model(63, precision="fp32", embedding=embedding)
    asset = make_synthetic_smplx_asset(seed=0)
    bm = BodyModel(asset).to(DEV)
    g = load("g10_normalizer")
    stats = {k.split("/")[-1]: torch.tensor(g[k]) for k in g.files if k.startswith("stats/axis_normalize")}
    gt = (g["raw"][:T] if T <= g["raw"].shape[0] else g["toy_pose_samples"][np.arange(T) % g["toy_pose_samples"].shape[0]]).astype(np.float32)
    rs = np.random.RandomState(seed)
    init = (gt + rs.standard_normal(gt.shape) * 0.05).astype(np.float32)
    _, jgt, _, _ = fk_ref.smplx_forward(asset, gt.astype(np.float64), dtype=np.float64)
    joints3d = (jgt[:, :22] + rs.standard_normal((T, 22, 3)) * 0.04).astype(np.float32)

    class Args:
        device = DEV

    nz = Posenormalizer(stats, device=DEV, normalize=True, min_max=min_max, rot_rep="axis")
    md = MotionDenoise(cfg, Args(), m, bm, sde_N=500, batch_size=T, normalizer=nz)
    dev = lambda a: torch.tensor(a, device=DEV)
    return md, dev(joints3d), dev(gt), dev(init), rs


@pytest.mark.parametrize("min_max,embedding", [(False, "positional"), (True, "positional"), (False, "fourier")])
def test_motion_denoise_one_call_loop_matches_the_autograd_loop(min_max, embedding):
    """dposer_motion_denoise_optimize (all steps queued from C: loss gradients + Adam as kernels) vs the same steps through
    autograd and torch.optim.Adam around the same HIP kernels, with injected prior noise; z-score and min-max normalisers;
    two outer iterations so the loss weights change.  The per-step loss log equals the autograd loop's loss values."""
    T, iters, spi = 12, 2, 4
    md, joints3d, gt, init, rs = _md_setup(T, min_max, embedding=embedding)
    assert md._fused_supported()
    noise = torch.tensor(rs.standard_normal((iters * spi, T, 63)).astype(np.float32), device=DEV)
    kw = dict(gt_poses=gt, time_strategy="3", iterations=iters, steps_per_iter=spi, noise=noise, init_poses=init)
    res_f = md.optimize(joints3d, fused=True, **kw)
    log = t2n(md.loss_log)
    res_u = md.optimize(joints3d, fused=False, **kw)
    assert rel_err(t2n(res_f["pose_body"]), t2n(res_u["pose_body"])) < 2e-5
    assert np.allclose(res_f["MPJPE"], res_u["MPJPE"], rtol=1e-4, atol=1e-4)
    assert log.shape == (iters * spi, 1, 3) and np.isfinite(log).all() and (log > 0).all()
    log = log[:, 0]
    # first step: the loss values at the initial pose, recomputed with the public pieces
    from dposer_amd.prior import prior_loss
    with torch.no_grad():
        body = md.body_model(betas=md.betas, pose_body=init)
        temp = body.v[:-1] - body.v[1:]
        l_temp = float(torch.mean(torch.sqrt(torch.sum(temp * temp, dim=2))))
        d = body.Jtr[:, :22] - joints3d
        l_data = float(torch.mean(torch.sqrt(torch.sum(d * d, dim=2))))
    assert abs(log[0, 0] - l_temp) / l_temp < 1e-5 and abs(log[0, 1] - l_data) / l_data < 1e-5


def test_motion_denoise_batch_of_sequences_equals_one_sequence_at_a_time():
    """optimize_sequences advances S sequences with the same launches (frames_per_sequence in the C entry: temporal neighbours,
    data-term decision and loss means per sequence).  With injected prior noise every sequence must come out exactly as
    ``optimize`` returns it alone -- per-pose kernels do not depend on the batch they run in."""
    F, S, iters, spi = 8, 3, 2, 3
    md, joints3d, gt, init, rs = _md_setup(F * S)
    noise = torch.tensor(rs.standard_normal((iters * spi, F * S, 63)).astype(np.float32), device=DEV)
    md.batch_size = F
    md.betas = md.betas[:F]
    kw = dict(time_strategy="3", iterations=iters, steps_per_iter=spi)
    res = md.optimize_sequences(joints3d.reshape(S, F, 22, 3), gt.reshape(S, F, 63), noise=noise, init_poses=init.reshape(S, F, 63), **kw)
    log = t2n(md.loss_log)
    assert res["pose_body"].shape == (S, F, 63) and res["MPJPE"].shape == (S, F) and log.shape == (iters * spi, S, 3)
    for i in range(S):
        sl = slice(i * F, (i + 1) * F)
        one = md.optimize(joints3d[sl], gt_poses=gt[sl], noise=noise[:, sl].contiguous(), init_poses=init[sl], **kw)
        assert torch.equal(res["pose_body"][i], one["pose_body"]), i
        assert np.allclose(res["MPVPE"][i], one["MPVPE"], rtol=1e-6, atol=1e-6)
        assert np.allclose(log[:, i, :2], t2n(md.loss_log)[:, 0, :2], rtol=1e-6)


@pytest.mark.parametrize("S,F,per_frame_betas", [(3, 8, False), (2, 23, False), (17, 60, False), (3, 12, True)])
def test_motion_denoise_fused_skinning_temporal_gradient_is_bit_identical(S, F, per_frame_betas, monkeypatch, tuning_env):
    """The three homes of the temporal term's gradient inside the one-call loop: the first two must carry the same bits -- poses and loss log --,
    the third the same vertices / vertex gradient and a backward that agrees to fp32 rounding:
    k_skin_x4 + k_md_vert_grad (vertices and their gradient through HBM); dposer_lbs_forward_temporal_grad (skinning + gradient in one
    pass, one run of frames per sequence and several: halo frames recomputed); and, round 6, dposer_lbs_backward_temporal (no skinning
    kernel at all: the skinning BACKWARD skins the frame and its two neighbours in registers and forms the gradient itself; workgroups of
    four poses that straddle sequence boundaries at F = 23, a partly filled last workgroup at 46 frames; 1020 frames = a size where it is
    the default: from 960)."""
    iters, spi = 2, 3
    md, joints3d, gt, init, rs = _md_setup(F * S)
    noise = torch.tensor(rs.standard_normal((iters * spi, F * S, 63)).astype(np.float32), device=DEV)
    md.batch_size = F
    md.betas = md.betas[:F]
    if per_frame_betas:                                  # a rest shape per frame ([T, V, 3] operands: the VSB instantiations of every kernel)
        md.betas = torch.tensor(rs.standard_normal((F * S, 10)).astype(np.float32) * 0.5, device=DEV)
    kw = dict(time_strategy="3", iterations=iters, steps_per_iter=spi)
    out = {}
    from dposer_amd import _C
    for tag, fused, nseg in (("two-kernel", "0", None), ("fused", "1", None), ("fused-2", "1", "2"), ("fused-3", "1", "3"), ("in-backward", "2", None),
                             ("in-backward-5", "2", None), ("default", None, None)):
        # (the same skinning-backward kernel under every variant -- the matrix-pipe one, forced onto the small batches: the small-batch
        #  kernels sum in another order, which is not what this test is about; "-5": five poses per workgroup, the form large batches take)
        tuning_env(DPOSER_MD_FUSED_TEMPORAL=fused, DPOSER_SKIN_TEMPORAL_NSEG=nseg, DPOSER_LBS_JOINT_STREAM_MIN="1",
                   DPOSER_SKIN_BWD_MFMA="5" if tag == "in-backward-5" else None)
        if tag.startswith("in-backward"):
            assert _C.lib().dposer_lbs_temporal_in_backward_ok(md.body_model.bm._handle(), 4, F * S) == 1
        res = md.optimize_sequences(joints3d.reshape(S, F, 22, 3), gt.reshape(S, F, 63), noise=noise, init_poses=init.reshape(S, F, 63), **kw)
        out[tag] = (res["pose_body"].clone(), md.loss_log.clone())
    assert torch.isfinite(out["two-kernel"][0]).all()
    in_bwd_default = F * S >= 960
    for tag in ("fused", "fused-2", "fused-3", "in-backward", "in-backward-5", "default"):
        if tag.startswith("in-backward") or (tag == "default" and in_bwd_default):
            # same vertices and the same vertex gradient, bit for bit (they are formed by the same contracted expressions); the backward half
            # then blends the own pose's 3 x 3 transform and forms T^T dv with FMAs where the plain backward kernel rounds twice: agreement to
            # fp32 rounding, not to the bit (the contracted form is what makes this mode 5 % faster: profiles/r06_skin_contract_ab.md)
            err = rel_err(t2n(out[tag][0]), t2n(out["two-kernel"][0]))
            _log_measured(f"temporal term inside the skinning backward vs two kernels, poses ({F * S} frames)", err)
            assert err < 2e-6, (tag, err)
            assert np.allclose(t2n(out[tag][1]), t2n(out["two-kernel"][1]), rtol=1e-5, atol=0), tag
            continue
        assert torch.equal(out[tag][0], out["two-kernel"][0]), (tag, float((out[tag][0] - out["two-kernel"][0]).abs().max()))
        assert torch.equal(out[tag][1], out["two-kernel"][1]), (tag, float((out[tag][1] - out["two-kernel"][1]).abs().max()))
    if in_bwd_default:
        assert torch.equal(out["default"][0], out["in-backward"][0]) and torch.equal(out["default"][1], out["in-backward"][1])      # deterministic, and what ships
    # four or five poses per workgroup: every vertex sum is formed by one wave in one order either way
    assert torch.equal(out["in-backward-5"][0], out["in-backward"][0]) and torch.equal(out["in-backward-5"][1], out["in-backward"][1])
    tuning_env(DPOSER_LBS_JOINT_STREAM_MIN=None)
    assert _C.lib().dposer_lbs_temporal_in_backward_ok(md.body_model.bm._handle(), 4, F * S) == (1 if F * S >= 320 else 0)      # (possible from 320 frames per call, the default from 960)


def test_motion_denoise_under_the_ve_sde():
    """training.sde = 'vesde' (motion_denoising.py:60-62 builds it).  Continuous score function: the one-call loop takes it (the prior's
    label is sigma(t), its output the score) and lands where the step-by-step autograd loop lands.  The discrete VE score function
    (continuous = False: the prior's label is round((T - t)(N - 1))) is on the one-call loop too since round 6 (DPOSER_SDE_VE_DISCRETE); its
    reference here is the autograd loop with the prior forced onto the step-by-step form (the reference's own discrete outputs: golden g25)."""
    from dposer_amd.algorithms.advanced import sde_lib
    F, S, iters, spi = 6, 2, 1, 3
    md, joints3d, gt, init, rs = _md_setup(F * S)
    md.sde = sde_lib.VESDE(sigma_min=0.01, sigma_max=50.0, N=500)
    md.batch_size = F
    md.betas = md.betas[:F]
    assert md.continuous is True                                                # read from config.training.continuous by __init__ (motion_denoising.py:94)
    assert md._fused_supported()
    noise = torch.tensor(rs.standard_normal((iters * spi, F * S, 63)).astype(np.float32), device=DEV)
    kw = dict(time_strategy="3", iterations=iters, steps_per_iter=spi)
    one_f = md.optimize(joints3d[F:], gt_poses=gt[F:], noise=noise[:, F:].contiguous(), init_poses=init[F:], fused=True, **kw)
    one_u = md.optimize(joints3d[F:], gt_poses=gt[F:], noise=noise[:, F:].contiguous(), init_poses=init[F:], fused=False, **kw)
    assert rel_err(t2n(one_f["pose_body"]), t2n(one_u["pose_body"])) < 2e-5
    assert float((one_f["pose_body"] - init[F:]).abs().max()) > 1e-3            # (the poses moved)
    md.continuous = False
    assert md._fused_supported()
    res = md.optimize_sequences(joints3d.reshape(S, F, 22, 3), gt.reshape(S, F, 63), noise=noise, init_poses=init.reshape(S, F, 63), **kw)
    assert res["pose_body"].shape == (S, F, 63) and res["MPJPE"].shape == (S, F) and np.isfinite(res["MPVPE"]).all()
    one = md.optimize(joints3d[F:], gt_poses=gt[F:], noise=noise[:, F:].contiguous(), init_poses=init[F:], fused=True, **kw)
    assert torch.equal(res["pose_body"][1], one["pose_body"])
    # ... against the autograd loop whose prior runs step by step on the host mirror of the reference's discrete score function
    from dposer_amd import prior as prior_mod
    from dposer_amd.algorithms.advanced import sde_lib as sl
    real = prior_mod.prior_loss

    def stepwise(model, sde, x0, t, *, weighted=True, reduction="mean", batch_size=None, z=None, seed=0, step=0, continuous=True):
        n = x0.numel() if reduction == "mean" else (batch_size if batch_size is not None else x0.shape[0])
        return prior_mod._prior_loss_unfused(model, sde, x0, float(t), bool(weighted), 1.0 / float(n), z, continuous=False)
    import dposer_amd.tasks.motion_denoising as mdmod
    monkey = mdmod.prior_loss
    mdmod.prior_loss = stepwise
    try:
        one_s = md.optimize(joints3d[F:], gt_poses=gt[F:], noise=noise[:, F:].contiguous(), init_poses=init[F:], fused=False, **kw)
    finally:
        mdmod.prior_loss = monkey
    assert rel_err(t2n(one["pose_body"]), t2n(one_s["pose_body"])) < 2e-5
    # the discrete score function conditions the network on round((T - t)(N - 1)) instead of sigma(t): it must really have been used
    assert not torch.equal(one["pose_body"], one_u["pose_body"])


def test_evaluate_motion_denoising_shards_sequences_over_ranks():
    """evaluate_motion_denoising: contiguous shard of the sequences per rank, several sequences per call, metric means from (sum,
    count) pairs.  Two 'ranks' evaluated one after the other cover all sequences; their frame-weighted means combine to the
    single-process means (time strategy '2': a fixed t, so the schedule does not depend on how the calls are split; the in-kernel
    noise is keyed by the frame index inside a call, so the same sequences-per-call split is used on both sides)."""
    from dposer_amd.tasks.motion_denoising import evaluate_motion_denoising
    F, S = 6, 5
    md, joints3d, gt, init, rs = _md_setup(F * S)
    md.batch_size = F
    md.betas = md.betas[:F]
    j, g = joints3d.reshape(S, F, 22, 3), gt.reshape(S, F, 63)
    kw = dict(time_strategy="2", sample_time=300, iterations=1, steps_per_iter=4, sequences_per_call=1)
    c0 = md._calls
    whole, n = evaluate_motion_denoising(md, j, g, num_replicas=1, rank=0, **kw)
    assert n == S and set(whole) == {"init_MPJPE", "MPJPE", "MPVPE"} and all(np.isfinite(v) for v in whole.values())
    assert whole["MPJPE"] < whole["init_MPJPE"] * 1.5
    md._calls = c0
    m0, n0 = evaluate_motion_denoising(md, j, g, num_replicas=2, rank=0, **kw)
    m1, n1 = evaluate_motion_denoising(md, j, g, num_replicas=2, rank=1, **kw)
    assert (n0, n1) == (3, 2)
    for k in whole:
        assert abs((m0[k] * n0 + m1[k] * n1) / S - whole[k]) < 1e-4 * abs(whole[k]), k


def test_motion_denoise_one_call_loop_inkernel_noise():
    """No injected noise: the prior's z comes from Philox(seed, step0 + i).  Same object state -> same result; the next call
    (advanced call counter) draws different noise; a zero-residual observation keeps everything finite (data term dropped on
    the device)."""
    T = 10
    md, joints3d, gt, init, rs = _md_setup(T)
    kw = dict(gt_poses=gt, time_strategy="3", iterations=1, steps_per_iter=5, init_poses=init)
    c0 = md._calls
    a = md.optimize(joints3d, **kw)["pose_body"]
    md._calls = c0
    b = md.optimize(joints3d, **kw)["pose_body"]
    c = md.optimize(joints3d, **kw)["pose_body"]
    assert torch.equal(a, b) and not torch.equal(a, c) and torch.isfinite(c).all()
    with torch.no_grad():
        j0 = md.body_model(betas=md.betas, pose_body=init).Jtr[:, :22].contiguous()
    z = md.optimize(j0, **kw)
    assert torch.isfinite(z["pose_body"]).all() and np.isfinite(z["MPJPE"]).all()


def test_evaler_min_over_hypotheses():
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    from dposer_amd.dataset.AMASS import Evaler
    asset = make_synthetic_smplx_asset(seed=0)
    bm = BodyModel(asset).to(DEV)
    g = load("g10_normalizer")
    gts = g["raw"][:5].astype(np.float32)
    rs = np.random.RandomState(2)
    outs = np.stack([gts + rs.standard_normal(gts.shape).astype(np.float32) * s for s in (0.2, 0.05, 0.1)], axis=1)
    ev = Evaler(bm, part="legs")
    res = ev.multi_eval_bodys(torch.tensor(outs, device=DEV), torch.tensor(gts, device=DEV))
    from dposer_amd.body_model.utils import BodyPartIndices, BodySegIndices
    ji, vi = np.array(BodyPartIndices.legs) + 1, np.array(BodySegIndices.legs)
    vg, jg, _, _ = fk_ref.smplx_forward(asset, gts.astype(np.float64), dtype=np.float64)
    mv, mj = [], []
    for h in range(3):
        vo, jo, _, _ = fk_ref.smplx_forward(asset, outs[:, h].astype(np.float64), dtype=np.float64)
        mv.append(np.sqrt(((vo - vg)[:, vi] ** 2).sum(-1)).mean(-1) * 1000)
        mj.append(np.sqrt(((jo - jg)[:, ji] ** 2).sum(-1)).mean(-1) * 1000)
    assert np.allclose(res["mpvpe_all"], np.min(mv, 0), rtol=1e-4, atol=1e-3)
    assert np.allclose(res["mpjpe_body"], np.min(mj, 0), rtol=1e-4, atol=1e-3)


def test_create_mask_mean_pose_observation(tmp_path, monkeypatch):
    """observation_type != 'noise' (misc.py:44-53): masked entries come from the SMPL mean-params file (rot6d), converted to
    axis-angle for rot_N = 3.  The asset is user supplied; a synthetic file with the same layout stands in."""
    from dposer_amd.body_model import constants
    from dposer_amd.utils.misc import create_mask, mask_indices
    from oracle import fk_ref
    rs = np.random.RandomState(4)
    aa = (rs.standard_normal((24, 3)) * 0.5)
    R = fk_ref.batch_rodrigues(aa)                                            # [24, 3, 3]
    pose6d = R[:, :, :2].reshape(24, 6).astype(np.float32).reshape(-1)        # row-major 3x2 = first two columns (transforms.py:227-235)
    f = tmp_path / "smpl_mean_params.npz"
    np.savez(f, pose=pose6d, shape=np.zeros(10, np.float32), cam=np.zeros(3, np.float32))
    monkeypatch.setattr(constants, "SMPL_MEAN_PATH", str(f))
    x = torch.tensor(rs.standard_normal((5, 63)).astype(np.float32), device=DEV)
    mask, obs = create_mask(x, part="legs", observation_type="mean")
    idx = mask_indices("legs", 3).numpy()
    keep = np.setdiff1d(np.arange(63), idx)
    assert np.array_equal(t2n(obs)[:, keep], t2n(x)[:, keep]) and float(mask[:, idx].abs().sum()) == 0.0
    want = aa[1:22].reshape(-1)[idx]                                          # body joints 1..21 of the mean pose, as axis-angle
    assert np.abs(t2n(obs)[:, idx] - want[None]).max() < 1e-5
    x6 = torch.tensor(rs.standard_normal((3, 126)).astype(np.float32), device=DEV)
    mask6, obs6 = create_mask(x6, part="left_arm", observation_type="mean")
    idx6 = mask_indices("left_arm", 6).numpy()
    assert np.abs(t2n(obs6)[:, idx6] - pose6d[6:][idx6][None]).max() == 0.0


def test_amass_dataset_rot6d_uses_gpu_conversion(tmp_path):
    import os
    from dposer_amd.dataset.AMASS import AMASSDataset
    from dposer_amd.utils.transforms import axis_angle_to_rot6d
    toy = torch.tensor(load("g10_normalizer")["toy_pose_samples"])
    os.makedirs(tmp_path / "v" / "train")
    torch.save(toy, tmp_path / "v" / "train" / "pose_body.pt")
    ds = AMASSDataset(str(tmp_path), version="v", subset="train", rot_rep="rot6d", normalize=False)
    assert ds.poses.shape == (toy.shape[0], 126) and not ds.poses.is_cuda
    want = axis_angle_to_rot6d(toy.reshape(-1, 3).to(DEV)).reshape(toy.shape[0], -1).cpu()
    assert torch.equal(ds.poses, want)
    # the 6 numbers are the first two columns of the rotation matrix, row-major 3x2 (transforms.py:227-235)
    R = fk_ref.batch_rodrigues(toy[:4].reshape(-1, 3).double().numpy())
    assert np.abs(ds.poses[:4].reshape(-1, 3, 2).numpy() - R[:, :, :2]).max() < 1e-5


def test_evaluate_completion_end_to_end():
    """run/completion.py:215-323 per-rank body: shard -> mask -> hypotheses -> de-normalise -> Evaler -> metric means, with the
    metrics kept on the device; completing the legs must beat leaving the noise observation in place, and the means must equal
    the ones computed from the gathered per-sample values."""
    from dposer_amd.algorithms.advanced import sde_lib
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    from dposer_amd.dataset.AMASS import Evaler, Posenormalizer
    from dposer_amd.tasks.completion import DPoserComp, evaluate_completion
    from dposer_amd.utils.misc import create_mask
    cfg, m, p = make_model(64, precision="bf16")
    bm = BodyModel(make_synthetic_smplx_asset(seed=0)).to(DEV)
    g = load("g10_normalizer")
    stats = {k.split("/")[-1]: torch.tensor(g[k]) for k in g.files if k.startswith("stats/axis_normalize")}
    nz = Posenormalizer(stats, device=DEV, normalize=True, min_max=False, rot_rep="axis")
    poses = torch.tensor(g["norm_minmax0"][:48])                    # z-scored toy poses
    sde = sde_lib.subVPSDE(0.1, 20.0, 1000)
    kw = dict(iterations=1, steps_per_iter=3)
    torch.manual_seed(5)
    means, n = evaluate_completion(m, sde, nz, bm, poses, part="legs", hypo=2, batch_size=16, optimize_kwargs=kw)
    assert n == 48 and set(means) == {"mpvpe_all", "mpjpe_body"}
    assert all(np.isfinite(v) and v > 0 for v in means.values())
    # two "ranks" evaluated one after the other cover the same samples (drop_last per rank): 24 = 16 + a dropped tail of 8 each
    torch.manual_seed(5)
    m0, n0 = evaluate_completion(m, sde, nz, bm, poses, part="legs", hypo=1, batch_size=16, num_replicas=2, rank=0, optimize_kwargs=kw)
    m1, n1 = evaluate_completion(m, sde, nz, bm, poses, part="legs", hypo=1, batch_size=16, num_replicas=2, rank=1, optimize_kwargs=kw)
    assert (n0, n1) == (16, 16)
    # the same numbers by hand for rank 0's batch
    torch.manual_seed(5)
    batch = poses[:16].to(DEV)
    mask, obs = create_mask(batch, part="legs")
    out = DPoserComp(m, sde, True, batch_size=16).optimize(obs, mask, **kw)
    ev = Evaler(bm, part="legs").multi_eval_bodys(nz.offline_denormalize(out[:, None], to_axis=True), nz.offline_denormalize(batch, to_axis=True))
    assert abs(float(np.mean(ev["mpjpe_body"])) - m0["mpjpe_body"]) < 1e-3 * m0["mpjpe_body"]


def test_completion_loop_vp_sde_and_fixed_time_strategy_vs_oracle():
    """The fused completion loop under the VP SDE (std = sqrt(1 - e^{2 lmc})) and with time strategy '2' (one fixed time step,
    completion.py:187): vs the oracle loop / a step-by-step loop through prior_loss + torch.optim.Adam."""
    from dposer_amd.algorithms.advanced import sde_lib
    from dposer_amd.tasks.completion import DPoserComp
    from dposer_amd.utils.misc import create_mask
    cfg, m, p = make_model(65, precision="fp32")
    B, iters, spi = 40, 2, 3
    g = load("g10_normalizer")
    torch.manual_seed(2)
    mask, obs = create_mask(torch.tensor(g["norm_minmax0"][:B]), part="arms")
    noise = np.random.RandomState(3).standard_normal((iters * spi, B, 63)).astype(np.float32)
    comp = DPoserComp(m, sde_lib.VPSDE(0.1, 20.0, 1000), continuous=True, batch_size=B)
    out = comp.optimize(obs.to(DEV), mask.to(DEV), iterations=iters, steps_per_iter=spi, noise=torch.tensor(noise, device=DEV))
    ref = task_loops.completion_optimize(p, R.VP(), obs.numpy(), mask.numpy(), noise, iterations=iters, steps_per_iter=spi)
    assert rel_err(t2n(out), ref) < 2e-6
    # time strategy '2' against the un-fused route (prior_loss + torch Adam, one step at a time)
    sde = sde_lib.subVPSDE(0.1, 20.0, 1000)
    comp = DPoserComp(m, sde, continuous=True, batch_size=B)
    nz = torch.tensor(noise, device=DEV)
    fused = comp.optimize(obs.to(DEV), mask.to(DEV), time_strategy="2", sample_time=700, iterations=iters, steps_per_iter=spi, noise=nz)
    x = obs.to(DEV).clone().requires_grad_(True)
    opt = torch.optim.Adam([x], 0.1, betas=(0.9, 0.999))
    ts = torch.linspace(1.0, 1e-3, 1000)
    w = comp.get_loss_weights()
    for step in range(iters * spi):
        it = step // spi
        opt.zero_grad()
        lp = comp.loss(x, float(ts[700]), weighted=True, z=nz[step])
        ld = torch.nn.functional.mse_loss(x * mask.to(DEV), obs.to(DEV) * mask.to(DEV))
        (w["dposer"](lp, it) + w["data"](ld, it)).backward()
        opt.step()
    slow = obs.to(DEV) * mask.to(DEV) + x.detach() * (1 - mask.to(DEV))
    assert rel_err(t2n(fused), t2n(slow)) < 2e-6


@pytest.mark.parametrize("part", ["left_leg", "right_leg", "left_arm", "right_arm", "trunk", "hands", "legs", "arms", None])
def test_evaler_matches_the_reference_evaluator(part):
    """g18 = the reference's own Evaler (lib/dataset/AMASS.py:263-316) on oracle.fk_torch bodies: part vertex / joint index sets
    (joint index + 1 skips the pelvis), per-sample MPVPE / MPJPE in mm, minimum over hypotheses -- against the device evaluator on
    the HIP body model (same synthetic asset)."""
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    from dposer_amd.dataset.AMASS import Evaler
    from helpers import load
    g = load("g18_evaler")
    bm = BodyModel(make_synthetic_smplx_asset(seed=0)).to(DEV)
    ev = Evaler(bm, part=part)
    outs, gts = torch.tensor(g["outs"], device=DEV), torch.tensor(g["gts"], device=DEV)
    tag = part or "all"
    r = ev.multi_eval_bodys(outs, gts)
    r0 = ev.eval_bodys(outs[:, 0].contiguous(), gts)
    for k in ("mpvpe_all", "mpjpe_body"):
        assert np.abs(r[k] - g[f"{tag}/{k}"]).max() / np.abs(g[f"{tag}/{k}"]).max() < 2e-5, (part, k)      # errors of ~100 mm to 1e-3 mm
        assert np.abs(r0[k].cpu().numpy() - g[f"{tag}/h0_{k}"]).max() / np.abs(g[f"{tag}/h0_{k}"]).max() < 2e-5, (part, k)


@pytest.mark.parametrize("min_max", [False, True])
def test_motion_denoise_one_call_loop_with_the_rot6d_representation(min_max):
    """rot_rep = 'rot6d' (a shipped data option, lib/dataset/AMASS.py:187-259): the network lives in 6 J = 126 coordinates (first
    two columns of every joint's rotation matrix, normalised with 126 statistics) while the pose being optimised stays axis-angle.
    The one-call loop converts in its normalise kernel and brings the prior gradient back through the Rodrigues Jacobian in its
    update kernel; it must track the step-by-step loop, whose conversion is plain differentiable torch (the reference goes through
    torchgeometry).  Round 3 ran this configuration step by step -- and without a gradient path through the conversion."""
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    from dposer_amd.dataset.AMASS import Posenormalizer
    from dposer_amd.tasks.motion_denoising import MotionDenoise, _axis_angle_to_rot6d_autograd
    from dposer_amd.utils.transforms import axis_angle_to_rot6d
    T, iters, spi = 12, 2, 3
    cfg, m, p = make_model(64, D=126, precision="fp32")
    asset = make_synthetic_smplx_asset(seed=0)
    bm = BodyModel(asset).to(DEV)
    g = load("g10_normalizer")
    gt = g["raw"][:T].astype(np.float32)
    rs = np.random.RandomState(3)
    # 126-D statistics of the toy poses in the 6-D representation
    six = t2n(axis_angle_to_rot6d(torch.tensor(g["raw"], device=DEV).reshape(-1, 3))).reshape(g["raw"].shape[0], -1)
    stats = dict(mean_poses=torch.tensor(six.mean(0)), std_poses=torch.tensor(six.std(0) + 1e-3), min_poses=torch.tensor(six.min(0) - 1e-3),
                 max_poses=torch.tensor(six.max(0) + 1e-3))
    init = (gt + rs.standard_normal(gt.shape) * 0.05).astype(np.float32)
    _, jgt, _, _ = fk_ref.smplx_forward(asset, gt.astype(np.float64), dtype=np.float64)
    joints3d = torch.tensor((jgt[:, :22] + rs.standard_normal((T, 22, 3)) * 0.04).astype(np.float32), device=DEV)

    class Args:
        device = DEV

    nz = Posenormalizer(stats, device=DEV, normalize=True, min_max=min_max, rot_rep="rot6d")
    md = MotionDenoise(cfg, Args(), m, bm, sde_N=500, batch_size=T, normalizer=nz)
    assert md._fused_supported()
    # the torch conversion of the step-by-step loop is the kernels' conversion
    aa = torch.tensor(init, device=DEV).reshape(-1, 3)
    assert float((_axis_angle_to_rot6d_autograd(aa) - axis_angle_to_rot6d(aa)).abs().max()) < 2e-6
    noise = torch.tensor(rs.standard_normal((iters * spi, T, 126)).astype(np.float32), device=DEV)
    kw = dict(gt_poses=torch.tensor(gt, device=DEV), time_strategy="3", iterations=iters, steps_per_iter=spi, noise=noise,
              init_poses=torch.tensor(init, device=DEV))
    res_f = md.optimize(joints3d, fused=True, **kw)
    log = t2n(md.loss_log)
    res_u = md.optimize(joints3d, fused=False, **kw)
    assert float((res_f["pose_body"] - kw["init_poses"]).abs().max()) > 1e-2           # the loop moved the pose
    assert rel_err(t2n(res_f["pose_body"]), t2n(res_u["pose_body"])) < 5e-5
    assert np.allclose(res_f["MPJPE"], res_u["MPJPE"], rtol=1e-4, atol=1e-4)
    assert log.shape == (iters * spi, 1, 3) and np.isfinite(log).all() and (log > 0).all()
