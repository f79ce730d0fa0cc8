"""Round-4 debugging aid: the Euler-Maruyama sampler's three forms (DPOSER_SAMPLER_PERSISTENT = 0 launches, 1 one workgroup per sample
block, 2 clusters of four) on one set of inputs, for a scan of step counts -- they must agree bit for bit.  (What it found: not a
race but `m2b0 * t - db * (t * t)` contracted into an fma in two of the three kernels; sde_dev.h now switches contraction off.)"""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import torch
from gpu_common import make_model
from dposer_amd import _C
from dposer_amd.algorithms.advanced import sampling, sde_lib
def run(mode, B, N, prec):
    os.environ['DPOSER_SAMPLER_PERSISTENT'] = str(mode)
    _C.lib().dposer_scorefc_tuning_reload()
    cfg, m, p = make_model(3, precision=prec)
    m.eval()
    sde = sde_lib.subVPSDE(0.1, 20.0, N)
    fn = sampling.get_sampling_fn(cfg, sde, (B, 63), lambda v: v, 1e-3, device='cuda:0')
    z = torch.randn(B, 63, device='cuda:0', generator=torch.Generator(device='cuda:0').manual_seed(5))
    _, x = fn(m, z=z, seed=11, traj_stride=0)
    return x.clone()
bad = 0
for prec in ('bf16', 'fp32'):
    for B, Ns in ((256, list(range(1, 21)) + [40, 100]), (2048, (4, 7, 13, 40)), (16384, (13, 200)), (65536, (200,))):
        for N in Ns:
            a = run(0, B, N, prec)
            for mode in (1, 2):
                d = float((a - run(mode, B, N, prec)).abs().max())
                bad += d > 0
                if d > 0: print(prec, B, N, 'mode', mode, 'max diff', d, flush=True)
print('mismatches:', bad)
