"""A/B of the Euler-Maruyama fast path: per-layer launches vs the persistent kernels (DPOSER_SAMPLER_PERSISTENT = 0: launches,
1: one workgroup per sample block, 2: clusters of four workgroups per sample block, 3: the cluster walk WITHOUT its waits -- a
timing bound, garbage samples), one box, interleaved child processes:
    python tools/sampler_ab.py [--batch 65536] [--steps 200] [--modes 0,2,3]
Every child also prints a checksum of the samples: modes 0, 1, 2 must agree bit for bit."""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from dposer_amd.algorithms.advanced import sampling, sde_lib
    from dposer_amd.algorithms.advanced.model import ScoreModelFC
    from dposer_amd.configs import load_config
    B, N, prec = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    cfg = load_config("configs.subvp.amass_scorefc_continuous.get_config")
    torch.manual_seed(42)
    dev = torch.device("cuda", 0)
    model = ScoreModelFC(cfg, n_poses=21, pose_dim=3, hidden_dim=1024, embed_dim=512, n_blocks=2)
    model.precision = prec
    model.to(dev).eval()
    sde = sde_lib.subVPSDE(beta_min=0.1, beta_max=20.0, N=N)
    fn = sampling.get_sampling_fn(cfg, sde, (B, 63), lambda v: v, 1e-3, device=dev)
    z = torch.randn(B, 63, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    fn(model, z=z, traj_stride=0)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        _, xs = fn(model, z=z, traj_stride=0)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    h = hashlib.sha1(xs.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"persistent={os.environ.get('DPOSER_SAMPLER_PERSISTENT', '1')} {prec} B={B} N={N}: {min(ts) * 1e3:8.2f} ms = {min(ts) / N * 1e6:7.1f} us / step  "
          f"({B / (min(ts) * 1000 / N):,.0f} samples/s at 1000 steps)  sha1 {h} finite {bool(torch.isfinite(xs).all())}", flush=True)
else:
    B = sys.argv[sys.argv.index("--batch") + 1] if "--batch" in sys.argv else "65536"
    N = sys.argv[sys.argv.index("--steps") + 1] if "--steps" in sys.argv else "200"
    prec = sys.argv[sys.argv.index("--precision") + 1] if "--precision" in sys.argv else "bf16"
    modes = sys.argv[sys.argv.index("--modes") + 1].split(",") if "--modes" in sys.argv else ["0", "1"]
    for rnd in range(2):
        for flag in modes:
            subprocess.run([sys.executable, __file__, "child", B, N, prec], env=dict(os.environ, DPOSER_SAMPLER_PERSISTENT=flag), check=False)
