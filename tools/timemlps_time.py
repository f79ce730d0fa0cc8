"""TimeMLPs (model.py:69-90) forward + backward on the GEMM family vs the same layers as torch modules on the same GPU (fp32 and bf16 autocast):
    python tools/timemlps_time.py [--batch 65536]
Prints ms per forward + backward (median of 5 runs of 10) and the relative difference of the outputs / a weight gradient."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn as nn
from dposer_amd.algorithms.advanced.model import TimeMLPs
from dposer_amd.configs import load_config

B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 65536
dev = "cuda:0"
cfg = load_config("configs.subvp.amass_scorefc_continuous.get_config")
cfg.model.dropout = 0.0
torch.manual_seed(0)
m = TimeMLPs(cfg, n_poses=21, pose_dim=3, hidden_dim=1024, n_blocks=2).to(dev)
ref = nn.Sequential(*[type(l)(l.in_features, l.out_features) if isinstance(l, nn.Linear) else (nn.SiLU() if isinstance(l, nn.SiLU) else nn.Dropout(0.0)) for l in m.net]).to(dev)
ref.load_state_dict(m.net.state_dict())
x = torch.randn(B, 63, device=dev)
t = torch.rand(B, device=dev) * 999
c = torch.randn(B, 63, device=dev)


def run(fn, params):
    for p in params:
        p.grad = None
    y = fn()
    (y * c).sum().backward()
    return y


def timed(fn, params):
    for _ in range(3):
        run(fn, params)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(fn, params)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10)
    return sorted(ts)[2]


def ref_fn():
    return ref(torch.cat([x, t[:, None]], 1))


def ref_bf16():
    with torch.autocast("cuda", dtype=torch.bfloat16):
        return ref(torch.cat([x, t[:, None]], 1)).float()


m.train()
ref.train()
y_ref = run(ref_fn, list(ref.parameters()))
g_ref = ref[2].weight.grad.clone()
for prec in ("bf16", "bf16x3", "fp32"):
    m.precision = prec
    y = run(lambda: m(x, t), list(m.parameters()))
    g = list(m.parameters())[2].grad
    ms = timed(lambda: m(x, t), list(m.parameters()))
    print(f"dposer_mlp {prec:6s} B={B}: {ms:7.3f} ms fwd+bwd   out rel {float((y - y_ref).norm() / y_ref.norm()):.2e}  dW rel {float((g - g_ref).norm() / g_ref.norm()):.2e}", flush=True)
print(f"torch modules fp32 B={B}: {timed(ref_fn, list(ref.parameters())):7.3f} ms fwd+bwd", flush=True)
print(f"torch modules bf16 autocast B={B}: {timed(ref_bf16, list(ref.parameters())):7.3f} ms fwd+bwd", flush=True)
