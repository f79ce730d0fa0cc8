#!/usr/bin/env python3
"""Upper bound of what a hipGraph replay of the training step could win (round 4):
    python tools/graph_probe.py [batch ...]
The fused step (dposer_dsm_loss_fwd_bwd + dposer_scorefc_adam_pack_step) is captured ONCE into a HIP graph -- with the per-step
scalars (Philox step, Adam step, learning rate, EMA decay) frozen at their captured values, which a real training loop could not do:
they are kernel arguments today -- and replayed; eager steps of the same library run next to it.  If replay is not faster than eager,
moving those scalars to device memory to make the graph legal buys nothing: the step is bound by its kernels, not by the host."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from dposer_amd.algorithms.advanced import losses, sde_lib  # noqa: E402
from dposer_amd.algorithms.advanced.model import ScoreModelFC  # noqa: E402
from dposer_amd.algorithms.ema import ExponentialMovingAverage  # noqa: E402
from dposer_amd.configs import load_config  # noqa: E402


def main():
    batches = [int(a) for a in sys.argv[1:]] or [1280, 4096, 8192]
    dev = torch.device("cuda", 0)
    cfg = load_config("configs.subvp.amass_scorefc_continuous.get_config")
    for B in batches:
        torch.manual_seed(42)
        model = ScoreModelFC(cfg, n_poses=21, pose_dim=3, hidden_dim=cfg.model.HIDDEN_DIM, embed_dim=cfg.model.EMBED_DIM, n_blocks=cfg.model.N_BLOCKS)
        model.precision = "bf16"
        model.to(dev)
        sde = sde_lib.subVPSDE(beta_min=cfg.model.beta_min, beta_max=cfg.model.beta_max, N=cfg.model.num_scales)
        opt = losses.get_optimizer(cfg, model.parameters())
        ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
        state = dict(optimizer=opt, model=model, ema=ema, step=6000)
        step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=True, continuous=True)
        x = torch.randn(B, 63, device=dev)
        for _ in range(10):
            step_fn(state, x)
        torch.cuda.synchronize()

        def timed(fn, n=300):
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        eager = timed(lambda: step_fn(state, x))
        # host time alone: enqueue without waiting for the GPU (the queue is deep enough for 50 steps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            step_fn(state, x)
        host = (time.perf_counter() - t0) / 50 * 1e3
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                step_fn(state, x)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            out = step_fn(state, x)
        torch.cuda.synchronize()
        replay = timed(g.replay)
        eager2 = timed(lambda: step_fn(state, x))
        print(f"B = {B:6d}: eager {eager:.4f} / {eager2:.4f} ms per step, host enqueue alone {host:.4f} ms, graph replay (frozen scalars) {replay:.4f} ms, "
              f"loss {float(out['step_loss']):.3f}", flush=True)


if __name__ == "__main__":
    main()
