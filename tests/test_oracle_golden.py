"""Pins the CPU oracle (oracle/score_ref.py, oracle/fk_ref.py) to golden vectors captured from
the imported reference (tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from helpers import load, rel_err, probe, masks_from_keep
from weights import make_mlp_weights, make_weights
from oracle import score_ref as R
from oracle import fk_ref

torch.set_num_threads(8)
TOL = 2e-5          # fp32 restatement vs fp32 reference: summation-order noise only


@pytest.mark.parametrize("tag,D", [("axis_pos", 63), ("rot6d_pos", 126)])
def test_forward_positional(tag, D):
    g = load("g1_forward")
    p = make_weights(int(g[f"{tag}_seed"]), D=D)
    p["sigmas"] = R.sigma_table()
    x, t = torch.tensor(g[f"{tag}_x"]), torch.tensor(g[f"{tag}_t"])
    out = R.scorefc_forward(p, x, t * 999)
    assert rel_err(out, g[f"{tag}_model"]) < TOL
    assert rel_err(R.score_fn(p, R.SubVP(), x, t), g[f"{tag}_score_subvp"]) < TOL
    assert rel_err(R.score_fn(p, R.VP(), x, t), g[f"{tag}_score_vp"]) < TOL


def test_forward_fourier_ve():
    g = load("g1_forward")
    p = make_weights(int(g["axis_fourier_seed"]), D=63, fourier=True)
    p["sigmas"] = R.sigma_table()
    x, t = torch.tensor(g["axis_fourier_x"]), torch.tensor(g["axis_fourier_t"])
    out = R.score_fn(p, R.VE(), x, t, embedding_type="fourier")
    assert rel_err(out, g["axis_fourier_score_ve"]) < TOL


@pytest.mark.parametrize("tag", ["nodrop", "drop"])
def test_dsm_loss_and_grads(tag):
    g = load("g3_loss_grads")
    p = make_weights(int(g["seed"]))
    p["sigmas"] = R.sigma_table()
    names = [n for n in R.param_names()]
    leaves = {n: p[n].clone().requires_grad_(True) for n in names}
    full = dict(p)
    full.update(leaves)
    batch = torch.tensor(g[f"{tag}_batch"])
    t = torch.tensor(g[f"{tag}_u"]) * (1.0 - 1e-5) + 1e-5
    z = torch.tensor(g[f"{tag}_z"])
    fw = {}
    if tag == "drop":
        fw = dict(drop_masks=masks_from_keep(g["drop_keep"]), drop_p=0.1)
    loss = R.dsm_loss(full, R.SubVP(), batch, t, z, **fw)
    assert abs(loss.item() - float(g[f"{tag}_loss"])) / float(g[f"{tag}_loss"]) < 1e-5
    grads = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
    for n, gr in zip(names, grads):
        ref = g[f"{tag}_grad/{n}"]
        if gr is None:
            assert n.startswith("pre_dense_cond") and ref.shape == (1,)
            continue
        assert rel_err(probe(n, gr), ref) < 1e-4, n
    if tag == "nodrop":
        lw = R.dsm_loss(p, R.SubVP(), batch, t, z, reduce_mean=False, likelihood_weighting=True)
        assert abs(lw.item() - float(g["lw_loss"])) / float(g["lw_loss"]) < 1e-5


def test_train_steps():
    g = load("g4_train_steps")
    p = make_weights(int(g["seed"]))
    p["sigmas"] = R.sigma_table()
    names = R.param_names()
    st = R.TrainState(p, names)
    batch = torch.tensor(g["batch"])
    for i in range(5):
        st.step = int(g[f"s{i}_step"])
        t = torch.tensor(g[f"s{i}_u"]) * (1.0 - 1e-5) + 1e-5
        z = torch.tensor(g[f"s{i}_z"])
        loss, _, _ = R.train_step(st, R.SubVP(), batch, t, z,
                                  drop_masks=masks_from_keep(g[f"s{i}_keep"]), drop_p=0.1)
        assert abs(loss.item() - float(g[f"s{i}_loss"])) / float(g[f"s{i}_loss"]) < 2e-4, i
        for n in names:
            assert rel_err(probe(n, st.p[n]), g[f"s{i}_param/{n}"]) < 1e-5, (i, n)
            assert rel_err(probe(n, st.ema[n]), g[f"s{i}_ema/{n}"]) < 1e-5, (i, n)
            if f"s{i}_m/{n}" in g.files:
                assert rel_err(probe(n, st.m[n]), g[f"s{i}_m/{n}"]) < 2e-3, (i, n)
                assert rel_err(probe(n, st.v[n]), g[f"s{i}_v/{n}"]) < 2e-3, (i, n)
    assert st.ema_updates == int(g["ema_num_updates"])


def _sampler_case(g, tag, N, **kw):
    p = make_weights(int(g["seed"]))
    p["sigmas"] = R.sigma_table()
    z0 = torch.tensor(g[f"{tag}_z0"])
    if f"{tag}_noise" in g.files:
        noise = torch.tensor(g[f"{tag}_noise"])
    else:
        rs = np.random.RandomState(int(g[f"{tag}_noise_seed"]))
        noise = torch.tensor(np.stack([rs.standard_normal(z0.shape).astype(np.float32)
                                       for _ in range(int(g[f"{tag}_noise_count"]))]))
    return p, z0, noise


def test_sampler_em8():
    g = load("g5_sampler")
    p, z0, noise = _sampler_case(g, "em8", 8)
    trajs, x = R.pc_sampler(p, R.SubVP(N=8), z0, noise)
    assert rel_err(trajs, g["em8_trajs"]) < 1e-4
    assert rel_err(x, g["em8_final"]) < 1e-4


def test_sampler_denoise_start_step():
    g = load("g5_sampler")
    p, z0, noise = _sampler_case(g, "den8", 8)
    noises = [None] * 3 + list(noise)
    trajs, x = R.pc_sampler(p, R.SubVP(N=8), z0, noises, start_step=3)
    assert rel_err(trajs, g["den8_trajs"]) < 1e-4
    assert rel_err(x, g["den8_final"]) < 1e-4


def test_sampler_completion():
    g = load("g5_sampler")
    p, z0, noise = _sampler_case(g, "comp8", 8)
    # draw order per step: impute-after-corrector, EM z, impute-after-predictor
    imp = [(noise[3 * i], noise[3 * i + 2]) for i in range(8)]
    em = [noise[3 * i + 1] for i in range(8)]
    trajs, x = R.pc_sampler(p, R.SubVP(N=8), z0, em, observation=torch.tensor(g["comp8_obs"]),
                            mask=torch.tensor(g["comp8_mask"]), impute_noises=imp)
    assert rel_err(trajs, g["comp8_trajs"]) < 1e-4
    assert rel_err(x, g["comp8_final"]) < 1e-4


def test_sampler_langevin():
    g = load("g5_sampler")
    p, z0, noise = _sampler_case(g, "lang4", 4)
    cor = [None] * 996 + [noise[2 * i] for i in range(4)]
    em = [None] * 996 + [noise[2 * i + 1] for i in range(4)]
    trajs, x = R.pc_sampler(p, R.SubVP(N=1000), z0, em, corrector_noises=cor, start_step=996)
    assert np.isfinite(g["lang4_trajs"]).all()
    assert rel_err(trajs, g["lang4_trajs"]) < 1e-4
    assert rel_err(x, g["lang4_final"]) < 1e-4


def test_sampler_em1000():
    g = load("g5_sampler")
    p, z0, noise = _sampler_case(g, "em1000", 1000)
    trajs, x = R.pc_sampler(p, R.SubVP(N=1000), z0, noise)
    assert rel_err(trajs[99::100], g["em1000_trajs"]) < 2e-3
    assert rel_err(x, g["em1000_final"]) < 2e-3


def test_prior_loss():
    g = load("g7_prior_loss")
    p = make_weights(int(g["seed"]))
    p["sigmas"] = R.sigma_table()
    x0 = torch.tensor(g["x0"])
    sde = R.SubVP()
    ts = torch.linspace(1.0, 1e-3, 1000)
    for step in (0, 99, 100, 199):
        q = R.completion_quan_t(step, 200, 1000)
        assert q == int(g[f"s{step}_quan_t"])
        t = torch.ones(16) * ts[q]
        z = torch.tensor(g[f"s{step}_z"])
        loss, grad = R.dposer_prior_loss(p, sde, x0, t, z, weighted=bool(q))
        assert abs(loss.item() - float(g[f"s{step}_loss"])) / abs(float(g[f"s{step}_loss"])) < 1e-4
        assert rel_err(grad, g[f"s{step}_grad"]) < 1e-4
        lu, _ = R.dposer_prior_loss(p, sde, x0, t, z, weighted=False)
        assert abs(lu.item() - float(g[f"s{step}_loss_unweighted"])) / abs(float(g[f"s{step}_loss_unweighted"])) < 1e-4


def test_fourier_embedding_paths_of_the_restatement_match_the_reference():
    """Golden g20 (the reference with `embedding_type = 'fourier'`): DSM loss + gradients, EM sampler with and without the
    completion imputation, prior loss -- the restatement with `embedding_type="fourier"`."""
    g = load("g20_fourier_paths")
    fw = dict(embedding_type="fourier")
    p = make_weights(int(g["seed"]), D=63, fourier=True)
    p["sigmas"] = R.sigma_table()
    names = [n for n in R.param_names(fourier=True)]
    leaves = {n: p[n].clone().requires_grad_(True) for n in names}
    full = dict(p)
    full.update(leaves)
    t = torch.tensor(g["dsm_u"]) * (1.0 - 1e-5) + 1e-5
    loss = R.dsm_loss(full, R.SubVP(), torch.tensor(g["dsm_batch"]), t, torch.tensor(g["dsm_z"]), **fw)
    assert abs(loss.item() - float(g["dsm_loss"])) / float(g["dsm_loss"]) < 1e-5
    grads = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
    for n, gr in zip(names, grads):
        ref = g[f"dsm_grad/{n}"]
        if gr is None or ref.shape == (1,):
            assert ref.shape == (1,), n       # no gradient in the reference: pre_dense_cond (unused), gauss_proj.W (requires_grad = False)
            continue
        assert rel_err(probe(n, gr), ref) < 1e-4, n
    z0, noise = torch.tensor(g["em8_z0"]), torch.tensor(g["em8_noise"])
    trajs, x = R.pc_sampler(p, R.SubVP(N=8), z0, noise, **fw)
    assert rel_err(trajs, g["em8_trajs"]) < 1e-4 and rel_err(x, g["em8_final"]) < 1e-4
    nz = torch.tensor(g["comp8_noise"]).reshape(8, 3, 16, 63)
    trajs, x = R.pc_sampler(p, R.SubVP(N=8), torch.tensor(g["comp8_z0"]), nz[:, 1], observation=torch.tensor(g["comp8_obs"]),
                            mask=torch.tensor(g["comp8_mask"]), impute_noises=[(nz[i, 0], nz[i, 2]) for i in range(8)], **fw)
    assert rel_err(trajs, g["comp8_trajs"]) < 1e-4 and rel_err(x, g["comp8_final"]) < 1e-4
    x0 = torch.tensor(g["prior_x0"])
    ts = torch.linspace(1.0, 1e-3, 1000)
    for step in (0, 199):
        q = int(g[f"prior_s{step}_quan_t"])
        lp, grad = R.dposer_prior_loss(p, R.SubVP(), x0, torch.ones(16) * ts[q], torch.tensor(g[f"prior_s{step}_z"]), weighted=bool(q), **fw)
        assert abs(lp.item() - float(g[f"prior_s{step}_loss"])) / abs(float(g[f"prior_s{step}_loss"])) < 1e-4
        assert rel_err(grad, g[f"prior_s{step}_grad"]) < 1e-4


def test_scalar_tables():
    g = load("g8_scalars")
    t = torch.tensor(g["t"])
    x = torch.ones(1000, 1)
    for name, sde in (("subvp", R.SubVP()), ("vp", R.VP()), ("ve", R.VE())):
        mean, std = sde.marginal_prob(x, t)
        drift, diff = sde.sde(x, t)
        a, s = sde.alpha_sigma(t)
        for got, key in ((mean, "mean"), (std, "std"), (drift, "drift"), (diff, "diffusion"),
                         (a, "alpha"), (s, "sigma")):
            ref = g[f"{name}_{key}"]
            assert np.allclose(np.broadcast_to(got.numpy(), ref.shape), ref, rtol=2e-6, atol=1e-7), (name, key)
    assert np.array_equal(R.sigma_table().numpy(), g["sigmas_buffer"])
    emb = R.timestep_embedding(t[::50] * 999, 512)
    assert np.allclose(emb.numpy(), g["temb"], atol=2e-6)


def test_rot6d():
    g = load("g11_rot6d")
    out = fk_ref.rot6d_to_mat3x3(g["rot6d"].astype(np.float32))
    assert np.allclose(out, g["rotmat"], atol=2e-6)


def test_completion_loop_restatement_matches_the_reference_loop():
    """oracle/task_loops.completion_optimize vs the reference's own DPoserComp.optimize (g14: 2 x 4 steps, legs masked)."""
    from oracle import task_loops
    g = load("g14_completion_loop")
    p = make_weights(int(g["seed"]))
    p["sigmas"] = R.sigma_table()
    out = task_loops.completion_optimize(p, R.SubVP(), g["observation"], g["mask"], g["noise"], iterations=int(g["iterations"]),
                                         steps_per_iter=int(g["steps_per_iter"]))
    assert rel_err(out, g["out"]) < 1e-5
    m = g["mask"]
    assert np.array_equal(out * m, g["observation"] * m)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_motion_denoise_loop_restatement_matches_the_reference_loop(tag):
    """oracle/task_loops.motion_denoise_optimize vs the reference's own MotionDenoise.optimize driving the same torch body model
    (g15).  Case b: the observed joints equal the joints of the initial pose, so the reference's `data_term > 0` guard drops the
    data term at step 0 (its gradient would be 0/0)."""
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    from oracle import task_loops
    g = load("g15_motion_denoise_loop")
    st = load("g10_normalizer")
    p = make_weights(int(g["seed"]))
    p["sigmas"] = R.sigma_table()
    asset = make_synthetic_smplx_asset(seed=0)
    final, res = task_loops.motion_denoise_optimize(p, R.SubVP(N=int(g["sde_N"])), asset, st["stats/axis_normalize2/mean_poses"],
                                                    st["stats/axis_normalize2/std_poses"], g[f"{tag}_joints3d"], g["gt"], g["init"],
                                                    g[f"{tag}_noise"], iterations=int(g["iterations"]), steps_per_iter=int(g["steps_per_iter"]))
    assert np.isfinite(final).all()
    assert rel_err(final, g[f"{tag}_pose_final"]) < 1e-5
    for k in ("init_MPJPE", "MPJPE", "MPVPE"):
        assert np.allclose(res[k], g[f"{tag}_{k}"], rtol=1e-4, atol=1e-5), k


def test_guided_em_step_restatement_matches_reference_golden():
    """oracle.score_ref.em_guided_step vs the reference's EulerMaruyamaPredictor.update_fn_guide (golden g16: sub-VP and VP,
    two times, legs masked, injected z)."""
    g = load("g16_guided_step")
    p = make_weights(int(g["seed"]))
    p["sigmas"] = R.sigma_table()
    x_t, obs, mask = (torch.tensor(g[k]) for k in ("x_t", "obs", "mask"))
    for name, sde in (("subvp", R.SubVP(N=1000)), ("vp", R.VP(N=1000))):
        for tv in (0.9, 0.3):
            tag = f"{name}_t{int(tv * 10)}"
            y_hat, y_mean = R.em_guided_step(p, sde, x_t, torch.ones(x_t.shape[0]) * tv, torch.tensor(g[f"{tag}_z"]), obs, mask, grad_step=0.7)
            assert rel_err(y_mean.numpy(), g[f"{tag}_y_mean"]) < 1e-5, tag
            assert rel_err(y_hat.numpy(), g[f"{tag}_y_hat"]) < 1e-4, tag



def test_auxiliary_loss_step_restatement_matches_the_reference_step():
    """g17 = the reference's own get_step_fn(auxiliary_loss=True) (losses.py:91-119, :242-258) with oracle.fk_torch on the
    synthetic asset as its body model: the restated multi-step denoise, SNR weights, v2v / j2j terms, clipped gradients and the
    Adam update must reproduce it."""
    from oracle import fk_torch
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    g = load("g17_aux_loss")
    asset = make_synthetic_smplx_asset(seed=0)
    p = make_weights(int(g["seed"]))
    p["sigmas"] = R.sigma_table()
    names = R.param_names()
    st = R.TrainState(p, names)
    st.step = int(g["step"])
    batch = torch.tensor(g["batch"])
    t = torch.tensor(g["u"]) * (1.0 - 1e-5) + 1e-5
    z = torch.tensor(g["z"])
    mean, std = torch.tensor(g["mean"]), torch.tensor(g["std"])
    terms = {}

    def body(pose):
        v, j = fk_torch.smplx_forward(asset, pose.double())
        return v.float(), j.float()

    def loss_of(full):
        total, d = R.aux_loss(full, R.SubVP(), batch, t, z, denormalize=lambda x: x * std + mean, body_model=body,
                              denoise_steps=int(g["denoise_steps"]))
        terms.update({k: float(v) for k, v in d.items()})
        return total

    loss, grads, norm = R.train_step(st, R.SubVP(), batch, t, z, loss_override=loss_of)
    for k in ("step_loss", "score_loss", "v2v_loss", "j2j_loss"):
        assert abs(terms[k] - float(g[k])) / abs(float(g[k])) < 2e-5, (k, terms[k], float(g[k]))
    coef = min(1.0, 1.0 / (norm + 1e-6))                      # the golden holds the gradients AFTER clip_grad_norm_(1.0)
    for n in names:
        if grads[n] is not None:
            assert rel_err(probe(n, grads[n] * coef), g[f"grad/{n}"]) < 2e-4, n
        assert rel_err(probe(n, st.p[n]), g[f"param/{n}"]) < 1e-5, n


@pytest.mark.parametrize("act", ["elu", "relu", "lrelu"])
def test_other_activations_forward_and_dsm_gradients(act):
    """g19: the reference's ScoreModelFC with config.model.nonlinearity = elu / relu / lrelu (model.py:54-66)."""
    g = load("g19_activations")
    p = make_weights(int(g["seed"]))
    p["sigmas"] = R.sigma_table()
    batch, t = torch.tensor(g["batch"]), torch.tensor(g["t"])
    assert rel_err(R.scorefc_forward(p, batch, t * 999, nonlinearity=act), g[f"{act}_model"]) < TOL
    names = R.param_names()
    leaves = {n: p[n].clone().requires_grad_(True) for n in names}
    full = dict(p)
    full.update(leaves)
    tt = torch.tensor(g["u"]) * (1.0 - 1e-5) + 1e-5
    loss = R.dsm_loss(full, R.SubVP(), batch, tt, torch.tensor(g["z"]), nonlinearity=act)
    assert abs(loss.item() - float(g[f"{act}_loss"])) / float(g[f"{act}_loss"]) < 1e-5
    grads = torch.autograd.grad(loss, [leaves[n] for n in names], allow_unused=True)
    for n, gr in zip(names, grads):
        if gr is not None:
            assert rel_err(probe(n, gr), g[f"{act}_grad/{n}"]) < 1e-4, n


@pytest.mark.parametrize("tag,D,H,nb,act,seed", [("swish1024", 63, 1024, 2, "swish", 31), ("lrelu64", 126, 64, 2, "lrelu", 32), ("elu256", 63, 256, 1, "elu", 33)])
def test_timemlps_forward_gradients_and_dsm_loss(tag, D, H, nb, act, seed):
    """oracle.score_ref.timemlps_forward against the reference's TimeMLPs (golden g22): output, gradients of the recorded linear
    functional w.r.t. the input and every parameter, and the sub-VP DSM loss with the recorded draws."""
    g = load("g22_timemlps")
    p = {k: v.clone().requires_grad_(True) for k, v in make_mlp_weights(seed, D, H, nb).items()}
    x = torch.tensor(g[f"{tag}/x"]).requires_grad_(True)
    t, c = torch.tensor(g[f"{tag}/t"]), torch.tensor(g[f"{tag}/c"])
    y = R.timemlps_forward(p, x, t, n_blocks=nb, nonlinearity=act)
    assert rel_err(y.detach(), g[f"{tag}/y"]) < TOL
    (y * c).sum().backward()
    assert rel_err(x.grad, g[f"{tag}/dx"]) < TOL
    for n, w in p.items():
        assert rel_err(probe(n, w.grad), g[f"{tag}/grad/{n}"]) < 5 * TOL, n
        w.grad = None
    sde = R.SubVP()
    tt = torch.tensor(g[f"{tag}/dsm_u"]) * (1.0 - 1e-5) + 1e-5                  # losses.py:110
    z, batch = torch.tensor(g[f"{tag}/dsm_z"]), torch.tensor(g[f"{tag}/dsm_batch"])
    mean, std = sde.marginal_prob(batch, tt)
    score = -R.timemlps_forward(p, mean + std[:, None] * z, tt * 999, n_blocks=nb, nonlinearity=act) / std[:, None]      # utils.py:152-160
    loss = torch.mean(torch.mean(torch.square(score * std[:, None] + z), dim=-1))                                        # losses.py:121-131
    assert abs(loss.item() - float(g[f"{tag}/dsm_loss"])) < 5 * TOL * abs(float(g[f"{tag}/dsm_loss"]))
    loss.backward()
    for n, w in p.items():
        assert rel_err(probe(n, w.grad), g[f"{tag}/dsm_grad/{n}"]) < 10 * TOL, n


def test_ve_sde_paths():
    """The oracle under the variance-exploding SDE (sde_lib.py:234-292) against the reference's own outputs (golden g21): DSM loss +
    gradients, the EM sampler (plain and with completion imputation), the prior loss + its gradient, the completion loop."""
    from oracle import task_loops
    g = load("g21_ve_paths")
    mk = lambda N: R.VE(float(g["sigma_min"]), float(g["sigma_max"]), N)
    w = make_weights(int(g["seed"]), D=63)
    p = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    p["sigmas"] = R.sigma_table()
    t = torch.tensor(g["dsm_u"]) * (1.0 - 1e-5) + 1e-5
    loss = R.dsm_loss(p, mk(1000), torch.tensor(g["dsm_batch"]), t, torch.tensor(g["dsm_z"]))
    assert abs(loss.item() - float(g["dsm_loss"])) < TOL * abs(float(g["dsm_loss"]))
    loss.backward()
    for n in w:
        ref = g[f"dsm_grad/{n}"]
        if ref.shape == (1,):
            assert p[n].grad is None or float(p[n].grad.abs().max()) == 0.0
        else:
            assert rel_err(probe(n, p[n].grad), ref) < 2e-4, n
    p = dict(w)
    p["sigmas"] = R.sigma_table()
    with torch.no_grad():
        trajs, x = R.pc_sampler(p, mk(8), torch.tensor(g["em8_z0"]), list(torch.tensor(g["em8_noise"])))
        assert rel_err(trajs, g["em8_trajs"]) < 1e-4 and rel_err(x, g["em8_final"]) < 1e-4
        noise = torch.tensor(g["comp8_noise"])
        imp = [(noise[3 * i], noise[3 * i + 2]) for i in range(8)]
        em = [noise[3 * i + 1] for i in range(8)]
        trajs, x = R.pc_sampler(p, mk(8), torch.tensor(g["comp8_z0"]), em, observation=torch.tensor(g["comp8_obs"]),
                                mask=torch.tensor(g["comp8_mask"]), impute_noises=imp)
        assert rel_err(trajs, g["comp8_trajs"]) < 1e-4 and rel_err(x, g["comp8_final"]) < 1e-4
        x0 = torch.tensor(g["prior_x0"])
        for step in (0, 199):
            tt = torch.ones(x0.shape[0]) * float(g[f"prior_s{step}_t"])
            lp, gp = R.dposer_prior_loss(p, mk(1000), x0, tt, torch.tensor(g[f"prior_s{step}_z"]), weighted=bool(int(g[f"prior_s{step}_quan_t"])))
            assert abs(lp.item() - float(g[f"prior_s{step}_loss"])) < 2e-4 * abs(float(g[f"prior_s{step}_loss"]))
            assert rel_err(gp, g[f"prior_s{step}_grad"]) < 2e-4
    out = task_loops.completion_optimize(p, mk(1000), g["loop_observation"], g["loop_mask"], g["loop_noise"],
                                         iterations=int(g["loop_iterations"]), steps_per_iter=int(g["loop_steps_per_iter"]))
    assert rel_err(out, g["loop_out"]) < 2e-4


def test_discrete_ve_score_function_paths():
    """The oracle under the VE SDE with the DISCRETE score function (utils.py:175-181: the network conditioned on round((T - t)(N - 1)))
    against the reference's own outputs (golden g25): EM sampler (plain and with completion imputation), prior loss + gradient, completion loop."""
    from oracle import task_loops
    g = load("g25_ve_discrete_paths")
    mk = lambda N: R.VE(float(g["sigma_min"]), float(g["sigma_max"]), N, discrete=True)
    p = dict(make_weights(int(g["seed"]), D=63))
    p["sigmas"] = R.sigma_table()
    with torch.no_grad():
        trajs, x = R.pc_sampler(p, mk(8), torch.tensor(g["em8_z0"]), list(torch.tensor(g["em8_noise"])))
        assert rel_err(trajs, g["em8_trajs"]) < 1e-4 and rel_err(x, g["em8_final"]) < 1e-4
        noise = torch.tensor(g["comp8_noise"])
        imp = [(noise[3 * i], noise[3 * i + 2]) for i in range(8)]
        em = [noise[3 * i + 1] for i in range(8)]
        trajs, x = R.pc_sampler(p, mk(8), torch.tensor(g["comp8_z0"]), em, observation=torch.tensor(g["comp8_obs"]),
                                mask=torch.tensor(g["comp8_mask"]), impute_noises=imp)
        assert rel_err(trajs, g["comp8_trajs"]) < 1e-4 and rel_err(x, g["comp8_final"]) < 1e-4
        x0 = torch.tensor(g["prior_x0"])
        for step in (0, 100, 199):
            tt = torch.ones(x0.shape[0]) * float(g[f"prior_s{step}_t"])
            lp, gp = R.dposer_prior_loss(p, mk(1000), x0, tt, torch.tensor(g[f"prior_s{step}_z"]), weighted=bool(int(g[f"prior_s{step}_quan_t"])))
            assert abs(lp.item() - float(g[f"prior_s{step}_loss"])) < 2e-4 * abs(float(g[f"prior_s{step}_loss"]))
            assert rel_err(gp, g[f"prior_s{step}_grad"]) < 2e-4
    out = task_loops.completion_optimize(p, mk(1000), g["loop_observation"], g["loop_mask"], g["loop_noise"],
                                         iterations=int(g["loop_iterations"]), steps_per_iter=int(g["loop_steps_per_iter"]))
    assert rel_err(out, g["loop_out"]) < 2e-4
    # ... and the labels really differ from the continuous ones: the continuous oracle must NOT reproduce the golden
    with torch.no_grad():
        _, xc = R.pc_sampler(p, R.VE(float(g["sigma_min"]), float(g["sigma_max"]), 8), torch.tensor(g["em8_z0"]), list(torch.tensor(g["em8_noise"])))
    assert rel_err(xc, g["em8_final"]) > 1e-3


def test_discrete_vp_score_function_paths():
    """The oracle under the VP SDE with the DISCRETE score function (utils.py:157-162: label t (N - 1), std from the DDPM table) against the
    reference's own outputs (golden g26): EM sampler (plain and with completion imputation), prior loss + gradient, completion loop."""
    from oracle import task_loops
    g = load("g26_vp_discrete_paths")
    mk = lambda N: R.VP(float(g["beta_min"]), float(g["beta_max"]), N, discrete=True)
    p = dict(make_weights(int(g["seed"]), D=63))
    p["sigmas"] = R.sigma_table()
    with torch.no_grad():
        trajs, x = R.pc_sampler(p, mk(8), torch.tensor(g["em8_z0"]), list(torch.tensor(g["em8_noise"])))
        assert rel_err(trajs, g["em8_trajs"]) < 1e-4 and rel_err(x, g["em8_final"]) < 1e-4
        noise = torch.tensor(g["comp8_noise"])
        imp = [(noise[3 * i], noise[3 * i + 2]) for i in range(8)]
        em = [noise[3 * i + 1] for i in range(8)]
        trajs, x = R.pc_sampler(p, mk(8), torch.tensor(g["comp8_z0"]), em, observation=torch.tensor(g["comp8_obs"]),
                                mask=torch.tensor(g["comp8_mask"]), impute_noises=imp)
        assert rel_err(trajs, g["comp8_trajs"]) < 1e-4 and rel_err(x, g["comp8_final"]) < 1e-4
        x0 = torch.tensor(g["prior_x0"])
        for step in (0, 100, 199):
            tt = torch.ones(x0.shape[0]) * float(g[f"prior_s{step}_t"])
            lp, gp = R.dposer_prior_loss(p, mk(1000), x0, tt, torch.tensor(g[f"prior_s{step}_z"]), weighted=bool(int(g[f"prior_s{step}_quan_t"])))
            assert abs(lp.item() - float(g[f"prior_s{step}_loss"])) < 2e-4 * abs(float(g[f"prior_s{step}_loss"]))
            assert rel_err(gp, g[f"prior_s{step}_grad"]) < 2e-4
    out = task_loops.completion_optimize(p, mk(1000), g["loop_observation"], g["loop_mask"], g["loop_noise"],
                                         iterations=int(g["loop_iterations"]), steps_per_iter=int(g["loop_steps_per_iter"]))
    assert rel_err(out, g["loop_out"]) < 2e-4
    # ... and the labels really differ from the continuous ones: the continuous oracle must NOT reproduce the golden
    with torch.no_grad():
        _, xc = R.pc_sampler(p, R.VP(float(g["beta_min"]), float(g["beta_max"]), 8), torch.tensor(g["em8_z0"]), list(torch.tensor(g["em8_noise"])))
    assert rel_err(xc, g["em8_final"]) > 1e-3


def test_fk_oracle_against_smplx_golden(tmp_path):
    """oracle/fk_ref.py against outputs of the reference's own dependency smplx==0.1.28 (golden g23, written by tests/golden/pin_fk_parity.py
    wherever smplx can be installed -- it cannot here).  Until that file exists the FK / LBS half of the oracle stays "parity unpinned"
    and this test says so instead of passing."""
    import os
    import pytest
    from helpers import GOLDEN
    path = os.path.join(GOLDEN, "g23_smplx_pin.npz")
    if not os.path.exists(path):
        pytest.skip("FK / LBS parity unpinned: run tests/golden/pin_fk_parity.py where smplx==0.1.28 is importable")
    import pin_fk_parity as P
    from pin_cases import build_case, case_inputs
    from oracle import fk_ref
    g = np.load(path, allow_pickle=False)
    for i, layout in enumerate([str(c) for c in g["cases"]]):
        _, _, loaded = build_case(g, i, layout, tmp_path)
        d = case_inputs(g, layout)
        full, shape = P.full_pose_and_shape(P.CASES[i][0], d)
        v, j, _ = fk_ref.model_forward(loaded, full.astype(np.float64), shape=shape.astype(np.float64), transl=d["transl"].astype(np.float64))
        assert np.abs(v - g[f"{layout}/vertices"]).max() < 1e-5 and np.abs(j - g[f"{layout}/joints"]).max() < 1e-5, layout
