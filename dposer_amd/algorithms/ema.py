"""Exponential moving average of the parameters -- counterpart of the reference's
lib/algorithms/ema.py:10-98 (same constructor, methods and ``state_dict`` layout
``{decay, num_updates, shadow_params}``).

When the tracked parameters are views of one flat buffer (``ScoreModelFC.flat_params()``), the
shadow parameters are views of one flat shadow buffer too, so the fused Adam/EMA kernel
(``dposer_adam_ema_clip_step``) can update them in the same pass as the optimizer.
"""
import torch


def _bump_param_epoch():
    from .. import _C
    _C.bump_param_epoch()


def flat_base(params):
    """(flat tensor, offsets) if ``params`` are consecutive fp32 views of one storage, else (None, None)."""
    params = list(params)
    if not params:
        return None, None
    st = params[0].untyped_storage()
    offs = []
    for p in params:
        if p.dtype != torch.float32 or p.untyped_storage().data_ptr() != st.data_ptr() or not p.is_contiguous():
            return None, None
        offs.append(p.storage_offset())
    n = st.nbytes() // 4
    flat = torch.empty(0, dtype=torch.float32, device=params[0].device).set_(st, 0, (n,))
    return flat, offs


class ExponentialMovingAverage:
    """ema.py:10-98.  ``decay`` is warmed up as min(decay, (1+n)/(10+n)) when ``use_num_updates``."""

    def __init__(self, parameters, decay=0.999, use_num_updates=True):
        if decay < 0.0 or decay > 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        params = list(parameters)
        self._flat_shadow = None
        flat, offs = flat_base(params)
        if flat is not None:
            self._flat_shadow = flat.clone().detach()
            self.shadow_params = [self._flat_shadow[o:o + p.numel()].view(p.shape) for p, o in zip(params, offs) if p.requires_grad]
        else:
            self.shadow_params = [p.clone().detach() for p in params if p.requires_grad]
        self.collected_params = []

    def next_one_minus_decay(self):
        """Advance the update counter and return (1 - decay) for this update (ema.py:43-47)."""
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        return 1.0 - decay

    def flat_shadow_for(self, flat_params):
        """The flat shadow buffer when it mirrors ``flat_params`` element for element, else None."""
        fs = self._flat_shadow
        if fs is not None and fs.numel() == flat_params.numel() and fs.device == flat_params.device:
            return fs
        return None

    def update(self, parameters):
        """s -= (1 - decay) * (s - p) for every tracked parameter (ema.py:32-51)."""
        one_minus_decay = self.next_one_minus_decay()
        with torch.no_grad():
            params = [p for p in parameters if p.requires_grad]
            for s, p in zip(self.shadow_params, params):
                s.sub_(one_minus_decay * (s - p))

    def copy_to(self, parameters):
        """Write the averaged values into ``parameters`` (ema.py:53-64)."""
        params = [p for p in parameters if p.requires_grad]
        for s, p in zip(self.shadow_params, params):
            p.data.copy_(s.data)
        _bump_param_epoch()                    # (.data writes are invisible to version counters: packed weights are stale now)

    def store(self, parameters):
        """Remember the current parameter values (ema.py:66-75)."""
        self.collected_params = [p.clone() for p in parameters]

    def restore(self, parameters):
        """Write back what ``store`` remembered (ema.py:77-89)."""
        for c, p in zip(self.collected_params, parameters):
            p.data.copy_(c.data)
        _bump_param_epoch()

    def state_dict(self):
        return dict(decay=self.decay, num_updates=self.num_updates, shadow_params=self.shadow_params)

    def load_state_dict(self, state_dict):
        self.decay = state_dict["decay"]
        self.num_updates = state_dict["num_updates"]
        loaded = state_dict["shadow_params"]
        if self._flat_shadow is not None and len(loaded) == len(self.shadow_params):
            for s, l in zip(self.shadow_params, loaded):     # keep the flat backing
                s.copy_(l.to(s.device))
        else:
            self.shadow_params = loaded
            self._flat_shadow = None
