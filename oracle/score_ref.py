"""CPU restatement (torch CPU tensors, explicit math) of the reference's score path.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Every function cites the reference file:line
(relative to /root/reference) it follows.  Nothing here is imported by ``dposer_amd``.

All functions take a plain ``dict`` of parameter tensors keyed by the reference's ``state_dict``
names (``pre_dense.weight`` ...), so the same weights can be fed to the reference, to this oracle
and to the HIP path.  ``dtype`` follows the parameters (fp32 for parity with the reference CPU
path, fp64 when the oracle is used as the high-precision arbiter).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------------
# model.py
# --------------------------------------------------------------------------------------------
def sigma_table(sigma_min=0.01, sigma_max=50.0, num_scales=1000) -> torch.Tensor:
    """lib/algorithms/advanced/model.py:24-34,128 -- VE noise levels, f64 linspace -> f32 buffer."""
    s = np.exp(np.linspace(np.log(sigma_max), np.log(sigma_min), num_scales))
    return torch.tensor(s, dtype=torch.float)


def timestep_embedding(labels: torch.Tensor, dim: int, max_positions=10000) -> torch.Tensor:
    """lib/algorithms/advanced/model.py:37-51 (sinusoidal embedding of the labels t*999)."""
    half = dim // 2
    scale = math.log(max_positions) / (half - 1)
    if not labels.is_floating_point():
        labels = labels.float()                                                # model.py:46 `timesteps.float()` (integer labels: the discrete score functions)
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -scale).to(labels.dtype)
    arg = labels[:, None] * freq[None, :]
    emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)
    if dim % 2 == 1:
        emb = torch.nn.functional.pad(emb, (0, 1))
    return emb


def fourier_embedding(x: torch.Tensor, W: torch.Tensor) -> torch.Tensor:
    """lib/algorithms/advanced/model.py:19-21 (GaussianFourierProjection.forward)."""
    proj = x[:, None] * W[None, :] * 2 * np.pi
    return torch.cat([torch.sin(proj), torch.cos(proj)], dim=-1)


def silu(x):
    return x * torch.sigmoid(x)


def group_norm(x: torch.Tensor, gamma, beta, groups=32, eps=1e-5):
    """nn.GroupNorm(32, C) on a [B, C] input (model.py:112,133,137): groups of C/32 contiguous
    channels, biased variance, eps inside the sqrt, per-channel affine."""
    B, C = x.shape
    xg = x.reshape(B, groups, C // groups)
    mean = xg.mean(dim=2, keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=2, keepdim=True)
    xhat = ((xg - mean) / torch.sqrt(var + eps)).reshape(B, C)
    return xhat * gamma[None, :] + beta[None, :]


def _lin(p: Params, name: str, x):
    return x @ p[name + ".weight"].t() + p[name + ".bias"]


def scorefc_forward(p: Params, batch: torch.Tensor, labels: torch.Tensor, *, n_blocks=2,
                    embed_dim: Optional[int] = None, embedding_type="positional",
                    scale_by_sigma=True, drop_masks: Optional[Sequence[torch.Tensor]] = None,
                    drop_p: float = 0.0, nonlinearity: str = "swish") -> torch.Tensor:
    """ScoreModelFC.forward -- lib/algorithms/advanced/model.py:141-196.

    ``labels`` is what the reference calls ``t`` inside the model (= t*999 from get_score_fn).
    ``drop_masks``: optional list of 1+2*n_blocks {0,1} keep-masks [B, H]; when given, dropout is
    applied as ``h * mask / (1 - drop_p)`` exactly where model.py:170,178,185 apply ``self.dropout``.
    """
    # model.py:54-66 get_act: one activation module used after every GroupNorm and in shared_time_embed
    silu = {"swish": torch.nn.functional.silu, "elu": torch.nn.functional.elu, "relu": torch.relu,
            "lrelu": lambda v: torch.nn.functional.leaky_relu(v, negative_slope=0.2)}[nonlinearity]
    if embed_dim is None:
        embed_dim = p["shared_time_embed.0.weight"].shape[0]
    if embedding_type == "fourier":
        used_sigmas = labels                                                   # model.py:152
        temb = fourier_embedding(torch.log(used_sigmas), p["gauss_proj.W"])    # model.py:153
    elif embedding_type == "positional":
        used_sigmas = p["sigmas"].to(batch.dtype)[labels.long()]               # model.py:159
        temb = timestep_embedding(labels, embed_dim)                           # model.py:160
    else:
        raise ValueError(embedding_type)
    temb = silu(_lin(p, "shared_time_embed.0", temb))                          # model.py:164

    def drop(h, i):
        if drop_masks is None:
            return h
        return h * drop_masks[i].to(h.dtype) / (1.0 - drop_p)

    h = _lin(p, "pre_dense", batch) + _lin(p, "pre_dense_t", temb)             # model.py:166-167
    h = drop(silu(group_norm(h, p["pre_gnorm.weight"], p["pre_gnorm.bias"])), 0)  # :168-170
    di = 1
    for k in range(1, n_blocks + 1):
        h1 = _lin(p, f"b{k}_dense1", h) + _lin(p, f"b{k}_dense1_t", temb)      # :173-174
        h1 = drop(silu(group_norm(h1, p[f"b{k}_gnorm1.weight"], p[f"b{k}_gnorm1.bias"])), di)
        h2 = _lin(p, f"b{k}_dense2", h1) + _lin(p, f"b{k}_dense2_t", temb)     # :180-181
        h2 = drop(silu(group_norm(h2, p[f"b{k}_gnorm2.weight"], p[f"b{k}_gnorm2.bias"])), di + 1)
        di += 2
        h = h + h2                                                             # :187
    res = _lin(p, "post_dense", h)                                             # :189
    if scale_by_sigma:
        res = res / used_sigmas.reshape(-1, 1)                                 # :192-194
    return res


def timemlps_forward(p: Params, x: torch.Tensor, t: torch.Tensor, *, n_blocks=2, nonlinearity: str = "swish",
                     drop_masks: Optional[Sequence[torch.Tensor]] = None, drop_p: float = 0.0) -> torch.Tensor:
    """TimeMLPs.forward -- lib/algorithms/advanced/model.py:69-90: net(cat[x, t[:, None]]), net = Linear(D + 1, H), act,
    [Linear(H, H), act, Dropout] x n_blocks, Linear(H, D); parameters keyed ``net.<index in the Sequential>``.
    ``drop_masks``: optional n_blocks {0,1} keep masks [B, H] applied where model.py:82 has the Dropout modules."""
    act = {"swish": torch.nn.functional.silu, "elu": torch.nn.functional.elu, "relu": torch.relu,
           "lrelu": lambda v: torch.nn.functional.leaky_relu(v, negative_slope=0.2)}[nonlinearity]
    h = act(_lin(p, "net.0", torch.cat([x, t[:, None].to(x.dtype)], dim=1)))          # model.py:75-76, :90
    for k in range(n_blocks):
        h = act(_lin(p, f"net.{2 + 3 * k}", h))                                        # model.py:79-81
        if drop_masks is not None:
            h = h * drop_masks[k].to(h.dtype) / (1.0 - drop_p)                         # model.py:82
    return _lin(p, f"net.{2 + 3 * n_blocks}", h)                                       # model.py:85


# --------------------------------------------------------------------------------------------
# sde_lib.py
# --------------------------------------------------------------------------------------------
class SubVP:
    """lib/algorithms/advanced/sde_lib.py:184-231 (scalars only; T = 1)."""
    name = "subVPSDE"

    def __init__(self, beta_min=0.1, beta_max=20.0, N=1000, discrete=False):
        self.b0, self.b1, self.N, self.T = beta_min, beta_max, N, 1.0
        self.discrete_betas = torch.linspace(beta_min / N, beta_max / N, N)   # :197
        self.alphas = 1.0 - self.discrete_betas                              # :198
        self.sqrt_1m_alphas_cumprod = torch.sqrt(1.0 - torch.cumprod(self.alphas, dim=0))      # VPSDE :137-139
        self.discrete = discrete          # VP only: get_score_fn(..., continuous=False), utils.py:157-160

    def lmc(self, t):
        return -0.25 * t ** 2 * (self.b1 - self.b0) - 0.5 * t * self.b0     # :214

    def beta(self, t):
        return self.b0 + t * (self.b1 - self.b0)                            # :207

    def sde(self, x, t):                                                     # :206-211
        drift = -0.5 * self.beta(t)[:, None] * x
        discount = 1.0 - torch.exp(-2 * self.b0 * t - (self.b1 - self.b0) * t ** 2)
        return drift, torch.sqrt(self.beta(t) * discount)

    def marginal_prob(self, x, t):                                           # :213-217
        l = self.lmc(t)
        return torch.exp(l)[:, None] * x, 1 - torch.exp(2.0 * l)

    def alpha_sigma(self, t):                                                # :227-231
        l = self.lmc(t)
        return torch.exp(l[:, None]), 1.0 - torch.exp(2.0 * l)


class VP(SubVP):
    """lib/algorithms/advanced/sde_lib.py:122-181."""
    name = "VPSDE"

    def sde(self, x, t):                                                     # :146-150
        return -0.5 * self.beta(t)[:, None] * x, torch.sqrt(self.beta(t))

    def marginal_prob(self, x, t):                                           # :152-156
        l = self.lmc(t)
        return torch.exp(l[:, None]) * x, torch.sqrt(1.0 - torch.exp(2.0 * l))

    def alpha_sigma(self, t):                                                # :177-181
        l = self.lmc(t)
        return torch.exp(l[:, None]), torch.sqrt(1.0 - torch.exp(2.0 * l))


class VE:
    """lib/algorithms/advanced/sde_lib.py:234-292."""
    name = "VESDE"

    def __init__(self, sigma_min=0.01, sigma_max=50.0, N=1000, discrete=False):
        self.smin, self.smax, self.N, self.T = sigma_min, sigma_max, N, 1.0
        self.discrete = discrete          # get_score_fn(..., continuous=False): the network's label is round((T - t)(N - 1)), utils.py:175-178

    def sde(self, x, t):                                                     # :253-262
        sigma = self.smin * (self.smax / self.smin) ** t
        g = sigma * torch.sqrt(torch.tensor(2 * (np.log(self.smax) - np.log(self.smin))))
        return torch.zeros_like(x), g

    def marginal_prob(self, x, t):                                           # :264-267
        return x, self.smin * (self.smax / self.smin) ** t

    def alpha_sigma(self, t):                                                # :289-292
        return torch.tensor([[1.0]]), self.smin * (self.smax / self.smin) ** t


# --------------------------------------------------------------------------------------------
# utils.py : get_score_fn
# --------------------------------------------------------------------------------------------
def score_fn(p: Params, sde, x, t, **fw):
    """lib/algorithms/advanced/utils.py:127-186: VP / sub-VP (continuous :152-155, discrete VP :157-160) and VE (continuous :173, discrete :175-178)."""
    if sde.name in ("VPSDE", "subVPSDE"):
        if sde.name == "VPSDE" and getattr(sde, "discrete", False):
            labels = t * (sde.N - 1)                                         # utils.py:158
            out = scorefc_forward(p, x, labels, **fw)
            std = sde.sqrt_1m_alphas_cumprod[labels.long()]                  # :160
            return -out / std[:, None]
        labels = t * 999                                                     # utils.py:152
        out = scorefc_forward(p, x, labels, **fw)
        std = sde.marginal_prob(torch.zeros_like(x), t)[1]                   # utils.py:155
        return -out / std[:, None]                                           # utils.py:162
    if getattr(sde, "discrete", False):
        labels = sde.T - t                                                   # utils.py:176
        labels = labels * (sde.N - 1)                                        # :177
        labels = torch.round(labels).long()                                  # :178
    else:
        labels = sde.marginal_prob(torch.zeros_like(x), t)[1]                # utils.py:173
    return scorefc_forward(p, x, labels, **fw)


# --------------------------------------------------------------------------------------------
# losses.py
# --------------------------------------------------------------------------------------------
def dsm_loss(p: Params, sde, batch, t, z, *, reduce_mean=True, likelihood_weighting=False, **fw):
    """lib/algorithms/advanced/losses.py:80-137 with the random draws (t, z) injected.
    (reference: t = rand(B)*(T-eps)+eps, eps=1e-5 :110 ; z = randn_like(batch) :111)."""
    mean, std = sde.marginal_prob(batch, t)                                  # :112
    x_t = mean + std[:, None] * z                                            # :113
    score = score_fn(p, sde, x_t, t, **fw)                                   # :121
    red = (lambda a: a.mean(dim=-1)) if reduce_mean else (lambda a: 0.5 * a.sum(dim=-1))
    if not likelihood_weighting:
        losses = red(torch.square(score * std[:, None] + z))                # :124-125
    else:
        g2 = sde.sde(torch.zeros_like(batch), t)[1] ** 2                    # :127
        losses = red(torch.square(score + z / std[:, None])) * g2           # :128-129
    return losses.mean()                                                    # :131


def multi_step_denoise(p: Params, sde, x_t, t, t_end, N, **fw):
    """losses.py:91-106: N deterministic DDIM-style steps from t towards t_end on a linear time grid
    (lib/utils/misc.py:58-61 linear_interpolation); returns (score at the FIRST step, estimated clean sample)."""
    alpha_w = torch.linspace(0, 1, N + 1)[:, None]
    traj = (1 - alpha_w) * t + alpha_w * t_end                               # [N + 1, B]
    x = x_t
    score_first = None
    for i in range(N):
        a_c, s_c = sde.alpha_sigma(traj[i])                                  # alpha [B, 1], sigma [B]
        a_b, s_b = sde.alpha_sigma(traj[i + 1])
        score = score_fn(p, sde, x, traj[i], **fw)
        if i == 0:
            score_first = score
        noise = -score * s_c[:, None]                                        # :102 score -> noise prediction
        x = a_b / a_c * (x - s_c[:, None] * noise) + s_b[:, None] * noise    # :103
    return score_first, x


def aux_loss(p: Params, sde, batch, t, z, *, denormalize, body_model, denoise_steps=5, reduce_mean=True, **fw):
    """losses.py:108-119 (return_data=True) + :242-258 (auxiliary_loss=True, rot_rep='axis'): DSM loss on the score of the first
    denoising step, plus SNR-weighted vertex / joint errors of the body posed by the multi-step estimate against the batch's.
    ``body_model(pose)`` -> (vertices [B, V, 3], joints [B, J, 3]).  Returns (total, dict of the four terms)."""
    mean, std = sde.marginal_prob(batch, t)
    x_t = mean + std[:, None] * z
    alpha, sigma = sde.alpha_sigma(t)
    snr = alpha / sigma[:, None]                                             # [B, 1]
    score, est = multi_step_denoise(p, sde, x_t, t, t / (2 * denoise_steps), denoise_steps, **fw)
    red = (lambda a: a.mean(dim=-1)) if reduce_mean else (lambda a: 0.5 * a.sum(dim=-1))
    score_loss = red(torch.square(score * std[:, None] + z)).mean()
    weight = torch.log(1.0 + snr)                                            # :244
    v_gt, j_gt = body_model(denormalize(batch))
    v_pr, j_pr = body_model(denormalize(est))
    v2v = torch.mean(weight * torch.square(v_gt - v_pr).sum(dim=-1))         # :253
    j2j = torch.mean(weight * torch.square(j_gt - j_pr).sum(dim=-1))         # :254
    total = score_loss + v2v + j2j
    return total, {"step_loss": total, "score_loss": score_loss, "v2v_loss": v2v, "j2j_loss": j2j}


PARAM_ORDER_CACHE: Dict[int, List[str]] = {}


def param_names(n_blocks=2, fourier=False) -> List[str]:
    """``model.parameters()`` order of ScoreModelFC (model.py:98-139) -- the order EMA shadows
    (ema.py:28-29) and ``clip_grad_norm_`` see.  ``gauss_proj.W`` has requires_grad=False and
    sits in parameters() but is skipped by EMA/optimizer (ema.py:28, model.py:17)."""
    names = ["pre_dense", "pre_dense_t", "pre_dense_cond", "pre_gnorm"]
    out = [f"{n}.{s}" for n in names for s in ("weight", "bias")]
    if fourier:
        out.append("gauss_proj.W")
    out += ["shared_time_embed.0.weight", "shared_time_embed.0.bias"]
    for k in range(1, n_blocks + 1):
        for n in (f"b{k}_dense1", f"b{k}_dense1_t", f"b{k}_gnorm1",
                  f"b{k}_dense2", f"b{k}_dense2_t", f"b{k}_gnorm2"):
            out += [f"{n}.weight", f"{n}.bias"]
    out += ["post_dense.weight", "post_dense.bias"]
    return out


def ema_decay(num_updates_after_increment: int, decay=0.9999) -> float:
    """lib/algorithms/ema.py:43-46."""
    n = num_updates_after_increment
    return min(decay, (1 + n) / (10 + n))


class TrainState:
    """Explicit Adam + EMA state over the trainable tensors (losses.py:31-58, ema.py:10-51)."""

    def __init__(self, p: Params, names: Sequence[str], ema_rate=0.9999):
        self.names = list(names)
        self.p = p
        self.m = {n: torch.zeros_like(p[n]) for n in self.names}
        self.v = {n: torch.zeros_like(p[n]) for n in self.names}
        self.adam_t = {n: 0 for n in self.names}
        self.ema = {n: p[n].clone() for n in self.names}
        self.ema_rate = ema_rate
        self.ema_updates = 0
        self.step = 0


def train_step(st: TrainState, sde, batch, t, z, *, lr=2e-4, warmup=5000, grad_clip=1.0,
               beta1=0.9, beta2=0.999, eps=1e-8, reduce_mean=True, loss_override=None, **fw):
    """One ``step_fn`` (losses.py:220-263) with injected (t, z[, dropout masks]):
    zero_grad -> loss -> backward -> lr warm-up (:51-53) -> clip_grad_norm_ (:54-55) -> Adam
    (:56; torch.optim.Adam semantics, wd=0, amsgrad off) -> step+=1 (:262) -> EMA (:263).
    Parameters that receive no gradient (``pre_dense_cond``) are skipped by Adam (grad is None)
    and by the clip norm, but still tracked by EMA -- as in the reference."""
    leaves = {n: st.p[n].detach().clone().requires_grad_(True) for n in st.names}
    full = dict(st.p)
    full.update(leaves)
    # loss_override(full_params) -> scalar: another loss on the same step machinery (the auxiliary-loss step, losses.py:242-258)
    loss = loss_override(full) if loss_override is not None else dsm_loss(full, sde, batch, t, z, reduce_mean=reduce_mean, **fw)
    grads = torch.autograd.grad(loss, [leaves[n] for n in st.names], allow_unused=True)
    grads = dict(zip(st.names, grads))
    cur_lr = lr * min(st.step / warmup, 1.0) if warmup > 0 else lr
    live = [g for g in grads.values() if g is not None]
    total_norm = torch.sqrt(sum((g.double() ** 2).sum() for g in live)).to(live[0].dtype)
    # torch.nn.utils.clip_grad_norm_: coef = max_norm/(norm+1e-6), clamped to 1.
    coef = torch.clamp(grad_clip / (total_norm + 1e-6), max=1.0) if grad_clip >= 0 else 1.0
    for n in st.names:
        g = grads[n]
        if g is None:
            continue
        g = g * coef
        st.adam_t[n] += 1
        k = st.adam_t[n]
        st.m[n] = beta1 * st.m[n] + (1 - beta1) * g
        st.v[n] = beta2 * st.v[n] + (1 - beta2) * g * g
        bc1 = 1 - beta1 ** k
        bc2 = 1 - beta2 ** k
        denom = st.v[n].sqrt() / math.sqrt(bc2) + eps
        st.p[n] = st.p[n] - (cur_lr / bc1) * st.m[n] / denom
    st.step += 1
    st.ema_updates += 1
    d = ema_decay(st.ema_updates, st.ema_rate)
    for n in st.names:
        st.ema[n] = st.ema[n] - (1.0 - d) * (st.ema[n] - st.p[n])           # ema.py:51
    return loss.detach(), grads, float(total_norm)


# --------------------------------------------------------------------------------------------
# sampling.py
# --------------------------------------------------------------------------------------------
def rsde_sde(p, sde, x, t, probability_flow=False, **fw):
    """SDE.reverse -> RSDE.sde -- sde_lib.py:98-109."""
    drift, g = sde.sde(x, t)
    score = score_fn(p, sde, x, t, **fw)
    drift = drift - g[:, None] ** 2 * score * (0.5 if probability_flow else 1.0)
    return drift, g, score


def em_step(p, sde, x, t, z, **fw):
    """EulerMaruyamaPredictor.update_fn with injected noise -- sampling.py:182-188."""
    dt = -1.0 / sde.N
    drift, g, _ = rsde_sde(p, sde, x, t, **fw)
    x_mean = x + drift * dt
    x_new = x_mean + g[:, None] * np.sqrt(-dt) * z
    return x_new, x_mean


def em_guided_step(p, sde, x_t, t, z, observation, mask, grad_step=1.0, **fw):
    """EulerMaruyamaPredictor.update_fn_guide with injected noise -- sampling.py:191-207 around RSDE.sde(guide=True),
    sde_lib.py:98-109: the Euler-Maruyama step, then minus grad_step times the gradient w.r.t. x_t of
    || observation * mask - y0_hat * mask ||_F, y0_hat = (x_t + sigma^2 score(x_t, t)) / alpha (the Tweedie estimate,
    differentiated THROUGH the score network).  Returns (y_hat, y_mean)."""
    x = x_t.detach().clone().requires_grad_(True)
    dt = -1.0 / sde.N
    drift, g, score = rsde_sde(p, sde, x, t, **fw)
    alpha, sigma = sde.alpha_sigma(t)
    y_mean = x.detach() + drift.detach() * dt
    y_hat = y_mean + g[:, None] * np.sqrt(-dt) * z
    y0_hat = (x + (sigma ** 2)[:, None] * score) / alpha
    norm = torch.norm(observation * mask - y0_hat * mask)
    grad = torch.autograd.grad(norm, x)[0]
    return (y_hat - grad_step * grad).detach(), y_mean


def impute(sde, x, t, observation, mask, noise):
    """completion imputation -- sampling.py:416-420."""
    mean, std = sde.marginal_prob(observation, t)
    masked = mean + noise * std[:, None]
    return x * (1 - mask) + masked * mask


def langevin_step(p, sde, x, t, noise, snr=0.16, **fw):
    """LangevinCorrector.update_fn, n_steps=1, injected noise -- sampling.py:282-302."""
    if sde.name in ("VPSDE", "subVPSDE"):
        ts = (t * (sde.N - 1) / sde.T).long()
        alpha = sde.alphas.to(x.dtype)[ts]
    else:
        alpha = torch.ones_like(t)
    grad = score_fn(p, sde, x, t, **fw)
    gn = torch.norm(grad.reshape(grad.shape[0], -1), dim=-1).mean()
    nn_ = torch.norm(noise.reshape(noise.shape[0], -1), dim=-1).mean()
    step = (snr * nn_ / gn) ** 2 * 2 * alpha
    x_mean = x + step[:, None] * grad
    return x_mean + torch.sqrt(step * 2)[:, None] * noise, x_mean


def pc_sampler(p, sde, x_init, noises, *, eps=1e-3, start_step=0, observation=None, mask=None,
               impute_noises=None, corrector_noises=None, snr=0.16, denoise=True,
               keep_traj=True, **fw):
    """get_pc_sampler.pc_sampler -- sampling.py:429-466, EM predictor, corrector 'none' or
    'langevin' (when ``corrector_noises`` is given), optional completion imputation.

    ``noises[i]`` is the predictor's z at loop index i; ``impute_noises[i]`` = (after corrector,
    after predictor) imputation draws; all injected so the loop is deterministic."""
    x = x_init
    timesteps = torch.linspace(sde.T, eps, sde.N).to(x.dtype)                # :449
    trajs = []
    x_mean = x
    for i in range(start_step, sde.N):
        t = timesteps[i]
        vec_t = torch.ones(x.shape[0], dtype=x.dtype) * t                    # :458
        if corrector_noises is not None:
            x, x_mean = langevin_step(p, sde, x, vec_t, corrector_noises[i], snr=snr, **fw)
        if observation is not None:
            x = impute(sde, x, vec_t, observation, mask, impute_noises[i][0])
        x, x_mean = em_step(p, sde, x, vec_t, noises[i], **fw)               # :460
        if observation is not None:
            x = impute(sde, x, vec_t, observation, mask, impute_noises[i][1])
        if keep_traj:
            trajs.append(x)
    out = x_mean if denoise else x                                           # :466
    return (torch.stack(trajs, 0) if keep_traj else None), out


# --------------------------------------------------------------------------------------------
# prior-loss maths shared by completion.py / motion_denoising.py / smplify.py
# --------------------------------------------------------------------------------------------
def one_step_denoise(p, sde, x_t, t, **fw):
    """run/completion.py:105-110 == run/smplify.py:69-74 == run/motion_denoising.py:99-104."""
    score = score_fn(p, sde, x_t, t, **fw)
    alpha, sigma = sde.alpha_sigma(t)
    sigma2 = sigma ** 2
    x0_hat = (x_t + sigma2[:, None] * score) / alpha
    snr = alpha / torch.sqrt(sigma2)[:, None]
    return x0_hat, snr


def dposer_prior_loss(p, sde, x0, t, z, *, weighted=True, reduction="mean", batch_size=None, **fw):
    """completion.py:131-149 (reduction='mean') / smplify.py:93-107 (reduction='sum_over_batch').
    Returns (loss, analytic dloss/dx0): x0_hat is detached in the reference so the gradient is
    2*w*(x0 - x0_hat)/n."""
    mean, std = sde.marginal_prob(x0, t)
    x_t = mean + std[:, None] * z
    x0_hat, snr = one_step_denoise(p, sde, x_t, t, **fw)
    w = 0.5 * torch.sqrt(1 + snr) if weighted else torch.full_like(snr, 0.5)
    sq = w * (x0 - x0_hat) ** 2
    if reduction == "mean":
        n = x0.numel()
    else:
        n = batch_size if batch_size is not None else x0.shape[0]
    return sq.sum() / n, 2 * w * (x0 - x0_hat) / n


def completion_quan_t(step: int, total_steps: int, N: int, sample_trun=5.0) -> int:
    """time strategy '3' -- run/completion.py:189-190."""
    return N - math.floor((total_steps - step - 1) * (N / (sample_trun * total_steps))) - 2
