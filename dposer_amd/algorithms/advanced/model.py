"""ScoreModelFC / TimeMLPs -- the score networks of the reference's
lib/algorithms/advanced/model.py (ScoreModelFC :93-196, TimeMLPs :69-90, get_timestep_embedding
:37-51, GaussianFourierProjection :10-21, get_sigmas :24-34, get_act :54-66).

The module keeps the reference's constructor signature, sub-module names and therefore
``state_dict`` keys/shapes (reference checkpoints load unchanged, including the unused
``pre_dense_cond``), but owns its parameters as views into ONE flat fp32 buffer (what the fused
Adam/EMA kernel and the weight packer consume) and evaluates ``forward`` with the MFMA kernels of
``libdposer_hip.so``.  There is no torch/CPU fallback: a forward on CPU tensors raises.
"""
import contextlib
import functools
import os
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import _C
from ...engine import ScoreEngine, default_precision, param_state_key


class GaussianFourierProjection(nn.Module):
    """Fixed Gaussian random features for time steps (model.py:10-21)."""

    def __init__(self, embed_dim, scale=30.0):
        super().__init__()
        self.W = nn.Parameter(torch.randn(embed_dim // 2) * scale, requires_grad=False)

    def forward(self, x):
        proj = x[:, None] * self.W[None, :] * 2 * np.pi
        return torch.cat([torch.sin(proj), torch.cos(proj)], dim=-1)


def get_sigmas(config):
    """VE noise levels exp(linspace(ln sigma_max, ln sigma_min, num_scales)) (model.py:24-34)."""
    m = config.model
    return np.exp(np.linspace(np.log(m.sigma_max), np.log(m.sigma_min), m.num_scales))


def get_timestep_embedding(timesteps, embedding_dim, max_positions=10000):
    """Sinusoidal embedding [sin(t f_k) | cos(t f_k)], f_k = exp(-k ln(max_positions)/(half-1)) (model.py:37-51)."""
    assert len(timesteps.shape) == 1
    half = embedding_dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32, device=timesteps.device) * -(math.log(max_positions) / (half - 1)))
    arg = timesteps.float()[:, None] * freq[None, :]
    emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1), mode="constant")
    assert emb.shape == (timesteps.shape[0], embedding_dim)
    return emb


_ACTIVATIONS = {"elu": nn.ELU, "relu": nn.ReLU, "lrelu": functools.partial(nn.LeakyReLU, negative_slope=0.2), "swish": nn.SiLU}


def get_act(config):
    """model.py:54-66."""
    name = config.model.nonlinearity.lower()
    if name not in _ACTIVATIONS:
        raise NotImplementedError("activation function does not exist!")
    return _ACTIVATIONS[name]()


class _FlatParams(nn.Module):
    """Parameters as views into ONE flat fp32 buffer (what the weight packer and the fused optimizer consume).  Sub-classes set
    ``_param_list`` / ``_offsets`` / ``_num_flat`` / ``_flat`` at the end of ``__init__``."""

    def flat_params(self) -> torch.Tensor:
        """The flat fp32 buffer all parameters are views of (re-built if someone replaced ``.data``)."""
        dev = self._param_list[0].device
        ok = self._flat is not None and self._flat.device == dev
        if ok:
            base = self._flat.data_ptr()
            ok = all(p.data_ptr() == base + 4 * off for p, off in zip(self._param_list, self._offsets))
        if not ok:
            flat = torch.empty(self._num_flat, dtype=torch.float32, device=dev)
            for p, off in zip(self._param_list, self._offsets):
                flat[off:off + p.numel()].copy_(p.data.reshape(-1))
                p.data = flat[off:off + p.numel()].view(p.shape)
            self._flat = flat
        return self._flat

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._flat = None          # .to()/.cuda()/.float() replace parameter storage: re-flatten
        self.flat_params()
        return out


class _MlpEngine:
    """Handle, packed weights and workspaces of one TimeMLPs instance on one device (dposer_mlp_* of include/dposer_hip.h)."""

    def __init__(self, in_dim, out_dim, hidden_dim, n_blocks, precision, activation, dropout_p):
        import ctypes as C
        self.lib = _C.lib()
        self.h = C.c_void_p()
        d = _C.MlpDesc(in_dim, out_dim, hidden_dim, n_blocks, _C.PRECISIONS[precision], _C.ACTIVATIONS[activation], dropout_p)
        _C.check(self.lib.dposer_mlp_create(C.byref(d), C.byref(self.h)), "dposer_mlp_create")
        self.num_params = self.lib.dposer_mlp_num_params(self.h)
        self.in_dim, self.out_dim = in_dim, out_dim
        self._packed = None
        self._pack_key = None
        self._ws = {}

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.dposer_mlp_destroy(self.h)
            self.h = None

    def packed(self, flat, params):
        key = param_state_key(flat, params)
        if self._packed is None or self._packed.device != flat.device:
            self._packed = torch.empty(self.lib.dposer_mlp_packed_bytes(self.h), dtype=torch.uint8, device=flat.device)
            self._pack_key = None
        if key != self._pack_key:
            _C.check(self.lib.dposer_mlp_pack(self.h, _C.ptr(flat), _C.ptr(self._packed), _C.stream_ptr()), "dposer_mlp_pack")
            self._pack_key = key
        return self._packed

    def workspace(self, batch, device, fresh=False):
        """Inference calls share one buffer per batch size; a differentiable forward takes a buffer of its own (``fresh``), which its
        autograd node keeps until the node is freed."""
        n = self.lib.dposer_mlp_workspace_bytes(self.h, batch)
        if fresh:
            return torch.empty(n, dtype=torch.uint8, device=device)
        ws = self._ws.get((batch, device))
        if ws is None:
            ws = self._ws[(batch, device)] = torch.empty(n, dtype=torch.uint8, device=device)
        return ws


class _MlpFunction(torch.autograd.Function):
    N_FIXED = 5     # (module, xin, train_mode, seed, step) in front of the parameters

    @staticmethod
    def forward(ctx, module, xin, train_mode, seed, step, *params):
        eng = module._engine()
        flat = module._flat
        packed = eng.packed(flat, module._param_list)
        B = xin.shape[0]
        ws = eng.workspace(B, xin.device, fresh=True)
        out = torch.empty(B, eng.out_dim, dtype=torch.float32, device=xin.device)
        _C.check(eng.lib.dposer_mlp_forward(eng.h, _C.ptr(flat), _C.ptr(packed), _C.ptr(ws), _C.ptr(xin), _C.ptr(out), B,
                                            1 if train_mode else 0, 1, seed, step, _C.stream_ptr()), "dposer_mlp_forward")
        ctx.module, ctx.ws, ctx.flat = module, ws, flat
        ctx.train_mode, ctx.seed, ctx.step = train_mode, seed, step
        ctx.versions = tuple(p._version for p in module._param_list)
        return out

    @staticmethod
    def backward(ctx, dout):
        m = ctx.module
        eng = m._engine()
        if m._flat is not ctx.flat or tuple(p._version for p in m._param_list) != ctx.versions:
            raise RuntimeError("TimeMLPs: parameters were modified in place between forward and backward")
        B = dout.shape[0]
        dout = dout.contiguous().float()
        want = ctx.needs_input_grad[_MlpFunction.N_FIXED:]
        need_dw, need_dx = any(want), ctx.needs_input_grad[1]
        flat_grad = torch.empty(eng.num_params, dtype=torch.float32, device=dout.device) if need_dw else None
        dx = torch.empty(B, eng.in_dim, dtype=torch.float32, device=dout.device) if need_dx else None
        packed = eng.packed(ctx.flat, m._param_list)       # (same parameter versions: a no-op unless another device's call re-packed)
        _C.check(eng.lib.dposer_mlp_backward(eng.h, _C.ptr(ctx.flat), _C.ptr(packed), _C.ptr(ctx.ws), _C.ptr(dout), _C.ptr(flat_grad),
                                             _C.ptr(dx), B, 1 if ctx.train_mode else 0, ctx.seed, ctx.step, _C.stream_ptr()),
                 "dposer_mlp_backward")
        ctx.ws = None
        grads = [flat_grad[off:off + p.numel()].view_as(p) if w else None for p, off, w in zip(m._param_list, m._offsets, want)]
        return (None, dx, None, None, None, *grads)


class TimeMLPs(_FlatParams):
    """Plain MLP on [x, t] (model.py:69-90), the reference's secondary score model (run/train.py:163-170).

    Same constructor, sub-module names and ``state_dict`` keys as the reference; the parameters are views into one flat fp32 buffer
    and ``forward`` runs on the GEMM kernels of ``libdposer_hip.so`` (one launch per Linear + activation, dropout drawn in the
    epilogue; dposer_amd/csrc/mlp.hip).  Like ScoreModelFC there is no torch/CPU fallback: a forward on CPU tensors raises."""

    def __init__(self, config, n_poses=21, pose_dim=6, hidden_dim=64, n_blocks=2):
        super().__init__()
        dim = n_poses * pose_dim
        self.config = config
        self.data_dim, self.hidden_dim, self.n_blocks = dim, hidden_dim, n_blocks
        self.act = get_act(config)
        layers = [nn.Linear(dim + 1, hidden_dim), self.act]
        for _ in range(n_blocks):
            layers += [nn.Linear(hidden_dim, hidden_dim), self.act, nn.Dropout(p=config.model.dropout)]
        layers.append(nn.Linear(hidden_dim, dim))
        self.net = nn.Sequential(*layers)

        # Precision: what the configuration / DPOSER_PRECISION asks for; with neither, "bf16x3" -- the raw time label t * 999 is an INPUT
        # COLUMN of this model (model.py:89-90), and one bf16 operand keeps 8 bits of it (above 512 neighbouring labels collapse: 0.4 %);
        # the split operand keeps 16 (7.6e-6), on the bf16 matrix pipe (ScoreModelFC embeds t in fp32 before any GEMM and defaults to bf16).
        explicit = os.environ.get("DPOSER_PRECISION") is not None or (hasattr(config.model, "get") and config.model.get("precision", None) is not None)
        self.precision = default_precision(config) if explicit else "bf16x3"
        self._engines = {}
        self._flat = None
        self._param_list = list(self.parameters())
        self._offsets = [0]
        for p in self._param_list:
            self._offsets.append(self._offsets[-1] + p.numel())
        self._num_flat = self._offsets.pop()
        self.flat_params()
        self._rng_seed = int(getattr(config, "seed", 0) or 0) * 1000003 + 54321
        self._rng_step = 0

    def _engine(self) -> _MlpEngine:
        eng = self._engines.get(self.precision)
        if eng is None:
            eng = _MlpEngine(self.data_dim + 1, self.data_dim, self.hidden_dim, self.n_blocks, self.precision,
                             self.config.model.nonlinearity.lower(), float(self.config.model.dropout))
            assert eng.num_params == self._num_flat, "flat layout disagrees with parameters()"
            self._engines[self.precision] = eng
        return eng

    def forward(self, x, t, condition=None, mask=None):
        """x [B, dim], t [B] -> [B, dim]: net(cat[x, t[:, None]]) (model.py:89-90)."""
        _C.require_gpu(x, "TimeMLPs input")
        if self._param_list[0].device != x.device:
            raise _C.DPoserHipError("TimeMLPs parameters and input are on different devices")
        xin = torch.cat([x.float(), t.float()[:, None]], dim=1).contiguous()
        if xin.shape[0] == 0:
            return xin[:, :self.data_dim] * 0.0
        flat = self.flat_params()
        eng = self._engine()
        needs_grad = torch.is_grad_enabled() and (xin.requires_grad or any(p.requires_grad for p in self._param_list))
        if needs_grad or self.training:
            self._rng_step += 1
            return _MlpFunction.apply(self, xin, bool(self.training), self._rng_seed, self._rng_step, *self._param_list)
        B = xin.shape[0]
        out = torch.empty(B, self.data_dim, dtype=torch.float32, device=xin.device)
        _C.check(eng.lib.dposer_mlp_forward(eng.h, _C.ptr(flat), _C.ptr(eng.packed(flat, self._param_list)),
                                            _C.ptr(eng.workspace(B, xin.device)), _C.ptr(xin), _C.ptr(out), B, 0, 0, 0, 0, _C.stream_ptr()),
                 "dposer_mlp_forward")
        return out


class _ScoreFCFunction(torch.autograd.Function):
    """Differentiable forward: d/d params and d/d x through dposer_scorefc_backward.

    The activations of the forward pass stay in a WS_TRAIN workspace that this autograd node LEASES from the engine (one
    buffer per live node, returned when the node is freed), so any number of differentiable forwards may be outstanding.
    ``backward`` re-reads the packed (transposed) weights: if they were re-packed in between it packs them again, and if the
    parameters themselves were modified in place it raises, like torch does for a saved tensor."""

    N_FIXED = 6     # (module, x, labels, train_mode, seed, step) in front of the parameters

    @staticmethod
    def forward(ctx, module, x, labels, train_mode, seed, step, *params):
        eng = module._engine()
        flat = module._flat
        packed = eng.packed(flat, with_backward=True, force=not module.freeze_packed)
        lease = eng.lease_train_workspace(x.shape[0], x.device)
        out = torch.empty_like(x)
        freq = eng.freq(x.device, module._fourier_W())
        _C.check(eng.lib.dposer_scorefc_forward_train(eng.h, _C.ptr(flat), _C.ptr(packed), _C.ptr(lease.ws), _C.ptr(x), _C.ptr(labels),
                                                      _C.ptr(freq), _C.ptr(module.sigmas), _C.ptr(out), x.shape[0],
                                                      1 if train_mode else 0, seed, step, _C.stream_ptr()),
                 "dposer_scorefc_forward_train")
        ctx.module, ctx.train_mode, ctx.seed, ctx.step = module, train_mode, seed, step
        ctx.labels = labels
        ctx.lease = lease
        ctx.flat, ctx.pack_gen = flat, eng._pack_gen
        ctx.versions = tuple(p._version for p in module._param_list)
        return out

    @staticmethod
    def backward(ctx, dout):
        m = ctx.module
        eng = m._engine()
        if ctx.lease is None or ctx.lease.ws is None:
            raise _C.DPoserHipError("ScoreModelFC backward: the activations of this forward pass have been released")
        if m._flat is not ctx.flat or tuple(p._version for p in m._param_list) != ctx.versions:
            raise RuntimeError("ScoreModelFC: parameters were modified in place between forward and backward "
                               "(their packed copies no longer match the activations saved by the forward pass)")
        if eng._pack_gen != ctx.pack_gen or not eng._packed_bwd:
            # another forward re-packed the weights (possibly without the transposed copies): same values, pack them again
            eng.packed(ctx.flat, with_backward=True, force=True)
            ctx.pack_gen = eng._pack_gen
        B = dout.shape[0]
        dout = dout.contiguous().float()
        need_dx = ctx.needs_input_grad[1]
        # parameter gradients only when autograd asks for one (a Hutchinson / guidance VJP w.r.t. x alone then skips the wgrad
        # GEMMs, the bucket reductions and the 33 MB flat gradient)
        want = ctx.needs_input_grad[_ScoreFCFunction.N_FIXED:]
        need_dw = any(want)
        flat_grad = torch.empty(eng.num_params, dtype=torch.float32, device=dout.device) if need_dw else None
        dx = torch.empty(B, eng.D, dtype=torch.float32, device=dout.device) if need_dx else None
        _C.check(eng.lib.dposer_scorefc_backward(eng.h, _C.ptr(ctx.flat), _C.ptr(eng._packed), _C.ptr(ctx.lease.ws), _C.ptr(ctx.labels),
                                                 _C.ptr(m.sigmas), _C.ptr(dout), _C.ptr(flat_grad), _C.ptr(dx), B,
                                                 1 if ctx.train_mode else 0, ctx.seed, ctx.step, _C.stream_ptr()),
                 "dposer_scorefc_backward")
        grads = []
        for p, off, w in zip(m._param_list, eng.offsets, want):
            if w and not m._is_nograd(off):
                grads.append(flat_grad[off:off + p.numel()].view_as(p))
            else:
                grads.append(None)
        return (None, dx, None, None, None, None, *grads)


class ScoreModelFC(_FlatParams):
    """Time-embedded residual MLP with independent time projections per layer (model.py:93-196)."""

    def __init__(self, config, n_poses=21, pose_dim=6, hidden_dim=64, embed_dim=32, n_blocks=2):
        super().__init__()
        self.config = config
        self.n_poses = n_poses
        self.joint_dim = pose_dim
        self.n_blocks = n_blocks
        self.hidden_dim = hidden_dim
        self.embed_dim = embed_dim
        self.act = get_act(config)
        data_dim = n_poses * pose_dim

        # registration order == reference (model.py:109-139) => identical parameters() order / init stream
        self.pre_dense = nn.Linear(data_dim, hidden_dim)
        self.pre_dense_t = nn.Linear(embed_dim, hidden_dim)
        self.pre_dense_cond = nn.Linear(hidden_dim, hidden_dim)     # never used in forward (kept for checkpoints)
        self.pre_gnorm = nn.GroupNorm(32, num_channels=hidden_dim)
        self.dropout = nn.Dropout(p=config.model.dropout)
        self.time_embedding_type = config.model.embedding_type.lower()
        if self.time_embedding_type == "fourier":
            self.gauss_proj = GaussianFourierProjection(embed_dim=embed_dim, scale=config.model.fourier_scale)
        elif self.time_embedding_type == "positional":
            self.posit_proj = functools.partial(get_timestep_embedding, embedding_dim=embed_dim)
        else:
            assert 0
        self.shared_time_embed = nn.Sequential(nn.Linear(embed_dim, embed_dim), self.act)
        self.register_buffer("sigmas", torch.tensor(get_sigmas(config), dtype=torch.float))
        for idx in range(1, n_blocks + 1):
            for j in (1, 2):
                setattr(self, f"b{idx}_dense{j}", nn.Linear(hidden_dim, hidden_dim))
                setattr(self, f"b{idx}_dense{j}_t", nn.Linear(embed_dim, hidden_dim))
                setattr(self, f"b{idx}_gnorm{j}", nn.GroupNorm(32, num_channels=hidden_dim))
        self.post_dense = nn.Linear(hidden_dim, data_dim)

        # ---- MI355X engine state (not part of the state_dict) ----
        self.precision = default_precision(config)
        self.freeze_packed = False      # True: caller promises the weights do not change between forwards
        self._input_grad_only = False   # see input_grad_only()
        self._engines = {}
        self._flat = None
        self._param_list = list(self.parameters())
        self._offsets = [0]
        for p in self._param_list:
            self._offsets.append(self._offsets[-1] + p.numel())
        self._num_flat = self._offsets.pop()
        self.flat_params()
        self._rng_seed = int(getattr(config, "seed", 0) or 0) * 1000003 + 12345
        self._rng_step = 0

    @contextlib.contextmanager
    def input_grad_only(self):
        """Forwards inside this context are differentiable w.r.t. the INPUT only (Hutchinson divergence of the likelihood,
        likelihood.py:29-35; guided sampling, sampling.py:191-207): their backward skips the weight-gradient GEMMs, the bucket
        reductions and the 33 MB flat gradient.  The reference gets the same saving for free from torch's per-edge pruning."""
        prev, self._input_grad_only = self._input_grad_only, True
        try:
            yield self
        finally:
            self._input_grad_only = prev

    # ---- flat parameter storage -------------------------------------------------------------------
    def _engine(self) -> ScoreEngine:
        eng = self._engines.get(self.precision)
        if eng is None:
            act = self.config.model.nonlinearity.lower()          # get_act (model.py:54-66): elu / relu / lrelu / swish
            if act not in _C.ACTIVATIONS:
                raise NotImplementedError("activation function does not exist!")
            eng = ScoreEngine(activation=act, data_dim=self.n_poses * self.joint_dim, hidden_dim=self.hidden_dim, embed_dim=self.embed_dim,
                              n_blocks=self.n_blocks, embedding=self.time_embedding_type,
                              scale_by_sigma=bool(self.config.model.scale_by_sigma), num_scales=int(self.sigmas.numel()),
                              dropout_p=float(self.config.model.dropout), precision=self.precision)
            assert eng.offsets == self._offsets and eng.num_params == self._num_flat, "flat layout disagrees with parameters()"
            self._engines[self.precision] = eng
        return eng

    def _is_nograd(self, off):
        return any(lo <= off < hi for lo, hi in self._engine().nograd)

    def _fourier_W(self):
        return self.gauss_proj.W if self.time_embedding_type == "fourier" else None

    # ---- forward -----------------------------------------------------------------------------------
    def forward(self, batch, t, condition=None, mask=None):
        """batch [B, j*3|j*6], t [B] (the *labels*, i.e. t*999 for the continuous sub-VP model) -> [B, same].
        model.py:141-196."""
        _C.require_gpu(batch, "ScoreModelFC input")
        if self.sigmas.device != batch.device or self._param_list[0].device != batch.device:
            raise _C.DPoserHipError("ScoreModelFC parameters and input are on different devices")
        if batch.shape[0] == 0:
            # torch's layers return an empty [0, D] for an empty batch (model.py:141-196 has no special case); the kernels take B >= 1
            return batch.float() * 0.0
        flat = self.flat_params()
        eng = self._engine()
        x = batch.contiguous().float()
        labels = t.contiguous().float()
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self._param_list))
        if needs_grad or self.training:
            self._rng_step += 1
            # inside input_grad_only() the parameters are not handed to autograd at all: the backward then has no parameter
            # gradient to produce (ctx.needs_input_grad only reflects requires_grad flags, not what a given backward call wants)
            params = () if self._input_grad_only else self._param_list
            return _ScoreFCFunction.apply(self, x, labels, bool(self.training), self._rng_seed, self._rng_step, *params)
        packed = eng.packed(flat, with_backward=False, force=not self.freeze_packed)
        ws = eng.workspace(x.shape[0], _C.WS_INFER, 0, x.device)
        out = torch.empty_like(x)
        freq = eng.freq(x.device, self._fourier_W())
        _C.check(eng.lib.dposer_scorefc_forward(eng.h, _C.ptr(flat), _C.ptr(packed), _C.ptr(ws), _C.ptr(x), _C.ptr(labels),
                                                _C.ptr(freq), _C.ptr(self.sigmas), _C.ptr(out), x.shape[0], _C.stream_ptr()),
                 "dposer_scorefc_forward")
        return out

