#!/usr/bin/env python3
"""A/B timing of the B = 65536 training step with two builds of the library on ONE box (boxes differ by +-5 %):
    python tools/step_time.py tools/bin/libdposer_hip_r02.so dposer_amd/libdposer_hip.so [--rounds 3] [--steps 30]
Each (library, round) runs in a fresh child process, alternating, and prints ms per step; the parent prints medians."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(lib_path, steps, batch):
    sys.path.insert(0, ROOT)
    import torch
    from dposer_amd import _C
    _C.LIB_PATH = os.path.abspath(lib_path)
    from dposer_amd.algorithms.advanced import losses, sde_lib
    from dposer_amd.algorithms.advanced.model import ScoreModelFC
    from dposer_amd.algorithms.ema import ExponentialMovingAverage
    from dposer_amd.configs import load_config
    cfg = load_config("configs.subvp.amass_scorefc_continuous.get_config")
    torch.manual_seed(42)
    dev = torch.device("cuda", 0)
    model = ScoreModelFC(cfg, n_poses=21, pose_dim=3, hidden_dim=cfg.model.HIDDEN_DIM, embed_dim=cfg.model.EMBED_DIM, n_blocks=cfg.model.N_BLOCKS)
    model.precision = "bf16"
    model.to(dev)
    sde = sde_lib.subVPSDE(beta_min=cfg.model.beta_min, beta_max=cfg.model.beta_max, N=cfg.model.num_scales)
    state = dict(optimizer=losses.get_optimizer(cfg, model.parameters()), model=model, ema=ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate), step=0)
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg), reduce_mean=True, continuous=True)
    batch_x = torch.randn(batch, 63, device=dev)
    for _ in range(8):
        out = step_fn(state, batch_x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step_fn(state, batch_x)
    torch.cuda.synchronize()
    print(f"MS {(time.perf_counter() - t0) / steps * 1e3:.4f} loss {float(out['step_loss']):.4f}")


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
        sys.exit(0)
    libs = [a for a in sys.argv[1:] if not a.startswith("--") and a.endswith(".so")]
    rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 3
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 30
    batch = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 65536
    res = {l: [] for l in libs}
    for r in range(rounds):
        for l in libs:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", l, str(steps), str(batch)], capture_output=True, text=True)
            line = [x for x in out.stdout.splitlines() if x.startswith("MS ")]
            if not line:
                print(l, "FAILED", out.stderr[-500:])
                continue
            res[l].append(float(line[0].split()[1]))
            print(f"round {r} {l}: {line[0]}", flush=True)
    for l, v in res.items():
        v = sorted(v)
        if v:
            print(f"{l}: median {v[len(v) // 2]:.4f} ms  min {v[0]:.4f}  ({len(v)} runs, B = {batch})")
