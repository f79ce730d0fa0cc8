"""APD of the cfg-3 sampler run (500 samples, 1000 steps) in bf16 against fp32 for several z seeds and library builds (run on the GPU box):
    python tools/cfg3_apd_spread.py tools/bin/libdposer_hip_r03a.so dposer_amd/libdposer_hip.so
Shows the noise floor of the bf16 APD that tests/test_gpu_configs.py bounds."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def child(lib):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import numpy as np, torch
    from dposer_amd import _C
    _C.LIB_PATH = os.path.abspath(lib)
    from gpu_common import DEV, make_model, t2n
    from helpers import rel_err, load
    from dposer_amd.algorithms.advanced import sampling, sde_lib
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    from dposer_amd.utils.metric import average_pairwise_distance
    N, B, seed = 1000, 500, 2024
    g = load("g10_normalizer")
    mean, std = g["stats/axis_normalize2/mean_poses"], g["stats/axis_normalize2/std_poses"]
    bm = BodyModel(make_synthetic_smplx_asset(seed=0)).to(DEV)
    for zs in (500, 501, 502):
        z0 = np.random.RandomState(zs).standard_normal((B, 63)).astype(np.float32)
        out, apd = {}, {}
        for prec in ("bf16", "fp32"):
            cfg, m, p = make_model(5, precision=prec)
            sde = sde_lib.subVPSDE(0.1, 20.0, N)
            fn = sampling.get_sampling_fn(cfg, sde, (B, 63), lambda v: v, 1e-3, device=DEV)
            _, x = fn(m, z=torch.tensor(z0, device=DEV), seed=seed + zs, traj_stride=0)
            out[prec] = t2n(x)
            pose = torch.tensor(out[prec] * std + mean, dtype=torch.float32, device=DEV)
            apd[prec] = float(average_pairwise_distance(bm.fk_joints(pose)))
        d = out["bf16"] - out["fp32"]
        print(f"RES {os.path.basename(lib)} z{zs}: rel {rel_err(out['bf16'], out['fp32']):.3e} APD fp32 {apd['fp32']:.5f} bf16 {apd['bf16']:.5f} ({(apd['bf16']/apd['fp32']-1)*100:+.3f} %) mean d {d.mean():+.2e} norm ratio {np.linalg.norm(out['bf16'])/np.linalg.norm(out['fp32'])-1:+.3e}", flush=True)
if sys.argv[1] == "--child":
    child(sys.argv[2]); sys.exit(0)
for l in sys.argv[1:]:
    o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", l], capture_output=True, text=True)
    print("\n".join(x for x in o.stdout.splitlines() if x.startswith("RES")) or o.stderr[-800:], flush=True)
