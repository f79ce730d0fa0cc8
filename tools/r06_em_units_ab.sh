#!/bin/bash
# Run on the GPU box:  bash tools/r06_em_units_ab.sh -- the sampler's fused post_dense + Euler-Maruyama launch: Philox normals drawn in front of the K loop
# (tools/bin/libdposer_hip_r8.so: the build before) vs one Philox call at a time BETWEEN the K loop's stages (shipped); 65536 x 200 steps and cfg 3 (500 x 1000)
cd "$(dirname "$0")/.."
cat > /tmp/em_units.py <<'PY'
import sys, time, torch
sys.path.insert(0, ".")
from dposer_amd.algorithms.advanced import sampling, sde_lib
from dposer_amd.algorithms.advanced.model import ScoreModelFC
from dposer_amd.configs import load_config
from dposer_amd import _C
cfg = load_config("configs.subvp.amass_scorefc_continuous.get_config")
torch.manual_seed(0)
m = ScoreModelFC(cfg, n_poses=21, pose_dim=3, hidden_dim=1024, embed_dim=512, n_blocks=2).to("cuda:0")
m.precision = "bf16"; m.eval()
for B, N in ((65536, 200), (500, 1000)):
    sde = sde_lib.subVPSDE(0.1, 20.0, N)
    fn = sampling.get_sampling_fn(cfg, sde, (B, 63), lambda v: v, 1e-3, device="cuda:0")
    z = torch.randn(B, 63, device="cuda:0")
    fn(m, z=z, traj_stride=0); torch.cuda.synchronize()
    t0 = time.perf_counter(); _, x = fn(m, z=z, traj_stride=0); torch.cuda.synchronize(); e = time.perf_counter() - t0
    _C.profile_enable(True); fn(m, z=z, traj_stride=0); torch.cuda.synchronize(); pr = _C.profile_collect(); _C.profile_enable(False)
    em = [(k, v[0] / v[1] * 1e3) for k, v in pr.items() if "em_step" in k]
    print(f"B={B} N={N}: {e / N * 1e6:.1f} us/step  em-step launch {em[0][1]:.1f} us  checksum {float(x.double().sum()):.6f}")
PY
for r in 1 2 3; do
  echo "before  (run $r): $(DPOSER_LIB_PATH=$PWD/tools/bin/libdposer_hip_r8.so python /tmp/em_units.py 2>/dev/null | tr '\n' ' ')"
  echo "shipped (run $r): $(python /tmp/em_units.py 2>/dev/null | tr '\n' ' ')"
done
