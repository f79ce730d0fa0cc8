"""Autograd semantics of the HIP score network that torch users rely on: any number of differentiable forwards may be
alive at once (each autograd node leases its own training workspace), an inference / fused-step call between a forward and
its backward is harmless, in-place parameter updates in between raise like torch does for a saved tensor, and parameter
gradients are only computed when autograd asks for them.  (model.py:141-196 is an ordinary nn.Module in the reference, so all
of this is implied by "ScoreModelFC.forward must be differentiable")"""
import numpy as np
import pytest
import torch

from gpu_common import DEV, make_model, t2n
from helpers import rel_err

pytestmark = pytest.mark.gpu


def _inputs(seed, B=96):
    rs = np.random.RandomState(seed)
    x = torch.tensor(rs.standard_normal((B, 63)).astype(np.float32), device=DEV)
    t = torch.tensor(rs.uniform(1e-3, 1.0, B).astype(np.float32), device=DEV) * 999
    w = torch.tensor(rs.standard_normal((B, 63)).astype(np.float32), device=DEV)
    return x, t, w


def _grads_alone(m, x, t, w):
    m.zero_grad(set_to_none=True)
    xx = x.clone().requires_grad_(True)
    (m(xx, t) * w).sum().backward()
    return xx.grad.clone(), torch.cat([p.grad.reshape(-1) if p.grad is not None else torch.zeros(p.numel(), device=DEV) for p in m.parameters()])


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_two_outstanding_forwards_keep_their_own_activations(precision):
    cfg, m, p = make_model(4, precision=precision, dropout=0.0)
    xa, ta, wa = _inputs(1)
    xb, tb, wb = _inputs(2, B=160)                        # a different batch size: the second forward needs a larger workspace
    dxa, dwa = _grads_alone(m, xa, ta, wa)
    dxb, dwb = _grads_alone(m, xb, tb, wb)
    m.zero_grad(set_to_none=True)
    a = xa.clone().requires_grad_(True)
    b = xb.clone().requires_grad_(True)
    oa = m(a, ta)                                         # forward A ...
    ob = m(b, tb)                                         # ... forward B while A's graph is alive
    with torch.no_grad():
        m(xb, tb)                                         # and an inference call (re-packs without the transposed weights)
    ((oa * wa).sum() + (ob * wb).sum()).backward()        # backward runs B's node, then A's
    assert torch.equal(a.grad, dxa) and torch.equal(b.grad, dxb)          # bit-identical to the runs on their own
    dw = torch.cat([q.grad.reshape(-1) if q.grad is not None else torch.zeros(q.numel(), device=DEV) for q in m.parameters()])
    assert rel_err(t2n(dw), t2n(dwa + dwb)) < 1e-6
    del oa, ob, a, b                                      # the autograd nodes die with their outputs ...
    assert len(m._engine()._train_pool) == 2              # ... and both leases went back to the pool


def test_gradient_accumulation_over_micro_batches_and_fused_step_in_between():
    from dposer_amd.algorithms.advanced import losses, sde_lib
    cfg, m, p = make_model(4, precision="fp32", dropout=0.0)
    x, t, w = _inputs(3, B=128)
    dx_full, dw_full = _grads_alone(m, x, t, w)
    m.zero_grad(set_to_none=True)
    outs = [m(x[i:i + 64].clone(), t[i:i + 64]) for i in (0, 64)]          # two live graphs, nothing backpropagated yet
    # a fused DSM evaluation (its own workspace) between the forwards and their backward must not disturb them
    flat_grad = torch.empty(m._engine().num_params, device=DEV)
    losses.fused_dsm_grad(m, sde_lib.subVPSDE(0.1, 20.0, 1000), x, flat_grad=flat_grad, seed=1, step=0)
    for o, i in zip(outs, (0, 64)):
        (o * w[i:i + 64]).sum().backward()
    dw = torch.cat([q.grad.reshape(-1) if q.grad is not None else torch.zeros(q.numel(), device=DEV) for q in m.parameters()])
    assert rel_err(t2n(dw), t2n(dw_full)) < 2e-6


def test_in_place_parameter_update_between_forward_and_backward_raises():
    cfg, m, p = make_model(4, precision="fp32", dropout=0.0)
    x, t, w = _inputs(5)
    out = m(x.clone().requires_grad_(True), t)
    with torch.no_grad():
        m.post_dense.bias.add_(1.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        (out * w).sum().backward()


def test_backward_twice_with_retain_graph():
    cfg, m, p = make_model(4, precision="fp32", dropout=0.0)
    x, t, w = _inputs(6)
    xx = x.clone().requires_grad_(True)
    loss = (m(xx, t) * w).sum()
    g1, = torch.autograd.grad(loss, xx, retain_graph=True)
    g2, = torch.autograd.grad(loss, xx)
    assert torch.equal(g1, g2)


def test_input_gradient_only_skips_the_parameter_gradients():
    """A VJP w.r.t. x alone (Hutchinson divergence, likelihood.py:29-35; guidance, sampling.py:191-207) inside
    ScoreModelFC.input_grad_only() must not run the wgrad side: the backward call gets flat_grad = NULL."""
    from dposer_amd import _C
    cfg, m, p = make_model(4, precision="fp32", dropout=0.0)
    x, t, w = _inputs(7)
    lib = m._engine().lib
    seen = []
    real = lib.dposer_scorefc_backward

    class Spy:
        def __call__(self, *a):
            seen.append(a[7])          # flat_grad argument
            return real(*a)
    m._engine().lib = type("L", (), {"__getattr__": lambda s, n: Spy() if n == "dposer_scorefc_backward" else getattr(lib, n)})()
    try:
        xx = x.clone().requires_grad_(True)
        with m.input_grad_only():
            g, = torch.autograd.grad((m(xx, t) * w).sum(), xx)
        assert seen and (seen[-1] is None or getattr(seen[-1], "value", seen[-1]) in (None, 0))
    finally:
        m._engine().lib = lib
    dx_ref, _ = _grads_alone(m, x, t, w)
    assert torch.equal(g, dx_ref)


def test_fused_adam_load_state_dict_after_stepping():
    """Restoring a checkpoint into an optimizer that has already stepped (in-process rollback) must replace the flat moments and
    the step count, not keep the live ones (train.py:393-403 restore path)."""
    from dposer_amd.algorithms.advanced import losses, sde_lib
    from dposer_amd.algorithms.ema import ExponentialMovingAverage
    cfg, m, p = make_model(2, precision="fp32", dropout=0.0)
    cfg.optim.warmup = 0
    sde = sde_lib.subVPSDE(0.1, 20.0, 1000)
    opt = losses.get_optimizer(cfg, m.parameters())
    ema = ExponentialMovingAverage(m.parameters(), decay=0.9999)
    step_fn = losses.get_step_fn(sde, True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=True, continuous=True)
    state = dict(model=m, optimizer=opt, ema=ema, step=0)
    x, _, _ = _inputs(8, B=64)
    step_fn(state, x)
    import copy
    snap = copy.deepcopy(opt.state_dict())
    flat_snap = m.flat_params().clone()
    m_snap, v_snap = opt._flat_m.clone(), opt._flat_v.clone()
    for _ in range(3):
        step_fn(state, x)
    assert not torch.equal(opt._flat_m, m_snap)
    opt.load_state_dict(snap)
    assert torch.equal(opt._flat_m, m_snap) and torch.equal(opt._flat_v, v_snap) and opt._step_count == 1
    # and the next step continues from the restored state: same update as the original second step
    with torch.no_grad():
        m.flat_params().copy_(flat_snap)
    state["step"] = 1
    step_fn(state, x)
    sd = opt.state_dict()["state"]
    assert int(next(iter(sd.values()))["step"]) == 2
