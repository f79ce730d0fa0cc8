"""Per-module owner of the C handle, packed weights and scratch workspaces of the HIP score path.

PyTorch is used for device memory and streams only: every buffer is a torch tensor, every kernel
is enqueued on ``torch.cuda.current_stream()`` through the C ABI (dposer_amd/_C.py).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Dict, Optional

import torch

from . import _C


def positional_freq(embed_dim: int, max_positions: int = 10000) -> torch.Tensor:
    """Frequencies of the sinusoidal embedding, evaluated with the reference's exact fp32 expression
    (lib/algorithms/advanced/model.py:39-43) on the CPU so they are bit-identical to the reference's."""
    half = embed_dim // 2
    scale = math.log(max_positions) / (half - 1)
    return torch.exp(torch.arange(half, dtype=torch.float32) * -scale)


class ScoreEngine:
    """One engine = one (module, precision) pair."""

    def __init__(self, *, data_dim, hidden_dim, embed_dim, n_blocks, embedding, scale_by_sigma, num_scales,
                 dropout_p, precision, activation="swish"):
        self.lib = _C.lib()
        self.precision = precision
        desc = _C.ScoreFCDesc(data_dim, hidden_dim, embed_dim, n_blocks,
                              _C.EMB_FOURIER if embedding == "fourier" else _C.EMB_POSITIONAL,
                              1 if scale_by_sigma else 0, num_scales,
                              _C.PRECISIONS[precision], float(dropout_p), _C.ACTIVATIONS[activation])
        h = C.c_void_p()
        _C.check(self.lib.dposer_scorefc_create(C.byref(desc), C.byref(h)), "dposer_scorefc_create")
        self.h = h
        self.D, self.H, self.E = data_dim, hidden_dim, embed_dim
        self.num_params = self.lib.dposer_scorefc_num_params(h)
        n = self.lib.dposer_scorefc_num_tensors(h)
        self.offsets = [self.lib.dposer_scorefc_tensor_offset(h, i) for i in range(n)]
        self.numels = [self.lib.dposer_scorefc_tensor_numel(h, i) for i in range(n)]
        lo = (C.c_int64 * 2)()
        hi = (C.c_int64 * 2)()
        k = self.lib.dposer_scorefc_nograd_ranges(h, lo, hi)
        self.nograd = [(lo[i], hi[i]) for i in range(k)]
        nb = self.lib.dposer_scorefc_grad_buckets(h, None, None, 0)
        blo, bhi = (C.c_int64 * nb)(), (C.c_int64 * nb)()
        self.lib.dposer_scorefc_grad_buckets(h, blo, bhi, nb)
        self.grad_buckets = [(blo[i], bhi[i]) for i in range(nb)]        # flat ranges in backward completion order
        self._bucket_events = None
        self._packed: Optional[torch.Tensor] = None
        self._packed_bwd = False
        self._pack_gen = 0                 # bumped by every (re)pack: lets a backward notice that the packed buffer was rewritten
        self._fresh_key = None             # state key under which the fused optimizer step last wrote the packed copies itself
        self._key_flat = None
        self._ws: Dict[int, torch.Tensor] = {}
        self._train_pool = []              # free WS_TRAIN buffers (see lease_train_workspace)
        self.freq_cpu = positional_freq(embed_dim)
        self._freq: Optional[torch.Tensor] = None

    def bucket_events(self):
        """One HIP event per gradient bucket (created once), as a ctypes array for dposer_dsm_loss_fwd_bwd_bucketed."""
        if self._bucket_events is None:
            arr = (C.c_void_p * len(self.grad_buckets))()
            for i in range(len(self.grad_buckets)):
                ev = C.c_void_p()
                _C.check(self.lib.dposer_event_create(C.byref(ev)), "dposer_event_create")
                arr[i] = ev.value
            self._bucket_events = arr
        return self._bucket_events

    def __del__(self):
        try:
            if getattr(self, "_bucket_events", None) is not None:
                for ev in self._bucket_events:
                    self.lib.dposer_event_destroy(ev)
                self._bucket_events = None
            if getattr(self, "h", None):
                self.lib.dposer_scorefc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- buffers -------------------------------------------------------------------------------
    def freq(self, device, fourier_W=None):
        if fourier_W is not None:
            return fourier_W
        if self._freq is None or self._freq.device != device:
            self._freq = self.freq_cpu.to(device)
        return self._freq

    def packed(self, flat: torch.Tensor, with_backward: bool, force: bool = True, state_key=None):
        """(Re)pack the fp32 master weights into MFMA fragment order.  ``force=False`` reuses the
        current packing (caller guarantees the weights did not change).  ``state_key`` (see ``param_state_key``): when it equals the
        key under which the fused optimizer step wrote the packed copies itself (dposer_scorefc_adam_pack_step), they ARE the
        parameters' and nothing is launched."""
        need = self.lib.dposer_scorefc_packed_bytes(self.h, 1)
        self._key_flat = flat              # the tensor OBJECT whose version counter the freshness key reads (see mark_packed_by_optimizer)
        if self._packed is None or self._packed.device != flat.device:
            self._packed = torch.empty(need, dtype=torch.uint8, device=flat.device)
            force = True
            self._fresh_key = None
        if force and state_key is not None and state_key == self._fresh_key and (self._packed_bwd or not with_backward):
            return self._packed
        if force or (with_backward and not self._packed_bwd):
            _C.check(self.lib.dposer_scorefc_pack(self.h, _C.ptr(flat), _C.ptr(self._packed), 1 if with_backward else 0,
                                                  _C.stream_ptr()), "dposer_scorefc_pack")
            self._packed_bwd = bool(with_backward)
            self._pack_gen += 1
            self._fresh_key = None
        return self._packed

    def repack_target(self, flat: torch.Tensor):
        """The packed buffer the fused optimizer step may write into: it exists, lives on ``flat``'s device and holds the backward
        copies (so every zero-padded region has been written once); else None."""
        if self._packed is None or self._packed.device != flat.device or not self._packed_bwd or self.precision == "bf16x3":
            return None                    # (bf16x3: three column groups per weight segment -- packed by dposer_scorefc_pack only)
        return self._packed

    def mark_packed_by_optimizer(self, flat, params):
        """The fused optimizer step has just rewritten the packed copies from the parameters it updated.  The freshness key must be
        built from the SAME tensor object the next ``packed()`` call will present (the model's flat buffer): the optimizer reaches the
        storage through an alias of its own (losses.flat_base), whose version counter is a different one."""
        key_flat = getattr(self, "_key_flat", None)
        if key_flat is None or key_flat.data_ptr() != flat.data_ptr() or key_flat.numel() != flat.numel():
            self._fresh_key = None         # never packed from this buffer: the next step packs
        else:
            self._fresh_key = param_state_key(key_flat, params)
        self._pack_gen += 1

    def workspace(self, batch: int, mode: int, n_steps: int, device):
        need = self.lib.dposer_scorefc_workspace_bytes(self.h, batch, mode, n_steps)
        if need < 0:
            raise _C.DPoserHipError("dposer_scorefc_workspace_bytes failed")
        ws = self._ws.get(mode)
        if ws is None or ws.numel() < need or ws.device != device:
            ws = torch.empty(need, dtype=torch.uint8, device=device)
            self._ws[mode] = ws
        return ws

    def lease_train_workspace(self, batch: int, device) -> "TrainWorkspaceLease":
        """A WS_TRAIN workspace that belongs to ONE forward/backward pair until the lease object dies.

        The training workspace holds every activation the backward pass needs, so two differentiable forwards that are alive
        at the same time (two score evaluations in one graph, gradient accumulation, an eval forward between loss and
        ``.backward()``) must not share one: each autograd node keeps its lease, a node that is freed returns the buffer to
        the pool.  With 288 GB of HBM the pool simply grows to the number of simultaneously live graphs (3.4 GB each at
        B = 65536)."""
        need = self.lib.dposer_scorefc_workspace_bytes(self.h, batch, _C.WS_TRAIN, 0)
        if need < 0:
            raise _C.DPoserHipError("dposer_scorefc_workspace_bytes failed")
        best = None
        for i, ws in enumerate(self._train_pool):
            if ws.device == device and ws.numel() >= need and (best is None or ws.numel() < self._train_pool[best].numel()):
                best = i
        if best is not None:
            ws = self._train_pool.pop(best)
        else:
            # drop free buffers that are too small instead of keeping them next to the new, larger one
            self._train_pool[:] = [w for w in self._train_pool if w.device != device or w.numel() >= need]   # (in place: leases hold this list)
            ws = torch.empty(need, dtype=torch.uint8, device=device)
        return TrainWorkspaceLease(self._train_pool, ws)


def param_state_key(flat: torch.Tensor, params):
    """What the packed weights were derived from: the flat buffer, the library's parameter epoch (bumped by every routine of this
    package that writes ``.data``) the version counters of the parameters (bumped by every in-place torch operation on them) and the flat buffer's own
    counter (the parameters are ``.data`` views of it with counters of their own: ``flat.copy_()`` bumps only this one).  Collectives
    that write the buffer without any counter (RCCL broadcast / all-gather) bump the epoch: distributed.broadcast_ / all_gather_flat_."""
    v = flat._version          # writes THROUGH the flat buffer (flat.copy_(...), c10d collectives' in-place ops) never reach the views' counters
    for p in params:
        v += p._version
    return (flat.data_ptr(), _C.PARAM_EPOCH[0], v)


class TrainWorkspaceLease:
    """Owner token of one WS_TRAIN buffer; returns it to the engine's pool when garbage-collected (kernels that still use it
    were enqueued on the current stream before that, and any later user is enqueued behind them)."""

    __slots__ = ("_pool", "ws")

    def __init__(self, pool, ws):
        self._pool, self.ws = pool, ws

    def release(self):
        if self.ws is not None:
            self._pool.append(self.ws)
            self.ws = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def default_precision(config=None) -> str:
    p = os.environ.get("DPOSER_PRECISION")
    if p is None and config is not None:
        try:
            p = config.model.get("precision", None) if hasattr(config.model, "get") else None
        except Exception:
            p = None
    p = (p or "bf16").lower()
    if p not in _C.PRECISIONS:
        raise ValueError(f"unknown precision {p!r} (bf16 | fp32 | bf16x3)")
    return p
