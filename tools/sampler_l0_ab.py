import os, sys, time, torch
import os as _os; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, _os.path.join(_R, 'tests')); sys.path.insert(0, _os.path.join(_R, 'tests', 'golden'))
from gpu_common import *  # noqa
from test_gpu_score import make_model, _sampler, _dev
import numpy as np
cfg, m, p = make_model(1, precision="bf16")
B, N = 65536, 200
sde, fn = _sampler(m, cfg, N, B)
z0 = torch.randn(B, 63, device="cuda")
for tag, env in (("fused", None), ("unfused", "1"), ("fused", None), ("unfused", "1")):
    if env: os.environ["DPOSER_NO_L0_FUSION"] = env
    else: os.environ.pop("DPOSER_NO_L0_FUSION", None)
    fn(m, z=z0, seed=1, traj_stride=0); torch.cuda.synchronize()
    t = time.perf_counter()
    fn(m, z=z0, seed=1, traj_stride=0); torch.cuda.synchronize()
    dt = time.perf_counter() - t
    print(tag, "%.1f us/step" % (dt / N * 1e6), flush=True)
