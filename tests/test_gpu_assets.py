"""The model-file door on the GPU: ``BodyModel(bm_path=<file>)`` / ``SMPLX(<file>)`` -> HIP kernels, against the numpy oracle on
the arrays the file was written from (reference lib/body_model/body_model.py:14-66, smpl.py:49-52; files in the official key
layouts come from tests/asset_files.py), and assets with MORE than four skinning influences per vertex through every LBS entry
point (forward, the small-batch and the streaming backward, the one-call motion-denoising loop)."""
import numpy as np
import pytest
import torch

from asset_files import write_npz, write_pkl
from gpu_common import DEV, make_model, t2n
from helpers import _log_measured, load, rel_err
from oracle import fk_ref

pytestmark = pytest.mark.gpu


def _inputs(model_type, B, nb, ne, seed):
    rs = np.random.RandomState(seed)
    nbody = 23 if model_type == "smpl" else 21
    d = dict(root_orient=(rs.standard_normal((B, 3)) * 0.5).astype(np.float32), pose_body=(rs.standard_normal((B, nbody * 3)) * 0.4).astype(np.float32),
             betas=(rs.standard_normal((B, nb)) * 0.7).astype(np.float32), trans=rs.standard_normal((B, 3)).astype(np.float32))
    full = [d["root_orient"], d["pose_body"]]
    if model_type == "smplx":
        d["pose_jaw"] = (rs.standard_normal((B, 3)) * 0.2).astype(np.float32)
        d["pose_eye"] = (rs.standard_normal((B, 6)) * 0.2).astype(np.float32)
        d["expression"] = (rs.standard_normal((B, ne)) * 0.7).astype(np.float32)
        full += [d["pose_jaw"], d["pose_eye"]]
    if model_type in ("smplh", "smplx"):
        d["pose_hand"] = (rs.standard_normal((B, 90)) * 0.3).astype(np.float32)
        full.append(d["pose_hand"])
    shape = d["betas"] if model_type != "smplx" else np.concatenate([d["betas"], d["expression"]], axis=1)
    return d, np.concatenate(full, axis=1), shape


@pytest.mark.parametrize("model_type,layout,ext,nb,ne", [
    ("smplx", "smplx_v1.1", "npz", 10, 10), ("smplx", "smplx_v1.0", "npz", 10, 10), ("smplx", "smplx_v1.1", "pkl", 16, 50),
    ("smplh", "smplh_amass", "npz", 16, 0), ("smpl", "smpl", "pkl", 10, 0)])
def test_body_model_from_a_model_file_vs_oracle(tmp_path, model_type, layout, ext, nb, ne):
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_asset
    asset = make_synthetic_asset(model_type, seed=11, num_betas=nb, num_expressions=ne)
    path = str(tmp_path / f"model.{ext}")
    (write_npz if ext == "npz" else write_pkl)(asset, path, layout)
    bm = BodyModel(bm_path=path, num_betas=nb, num_expressions=ne, model_type=model_type).to(DEV)
    assert bm.num_joints == {"smpl": 23, "smplh": 51, "smplx": 54}[model_type]
    assert bm.J_regressor.shape == asset["J_regressor"].shape
    B = 33
    d, full, shape = _inputs(model_type, B, nb, ne, seed=3)
    o = bm(**{k: torch.tensor(v, device=DEV) for k, v in d.items()})
    v_ref, j_ref, _ = fk_ref.model_forward(asset, full.astype(np.float64), shape=shape.astype(np.float64), transl=d["trans"].astype(np.float64))
    ev, ej = np.abs(t2n(o.v) - v_ref).max(), np.abs(t2n(o.Jtr) - j_ref).max()
    _log_measured(f"BodyModel from {layout}.{ext}: vertices / joints max abs", max(ev, ej))
    assert ev < 1e-5 and ej < 1e-5
    assert o.Jtr.shape[1] == {"smpl": 45, "smplh": 73, "smplx": 127}[model_type]


def test_clamped_shape_space_builds_the_module_smplx_would(tmp_path):
    """A 10 + 10 SMPL-X file asked for 16 betas / 50 expression coefficients: smplx warns and builds with 10 / 10; so does this."""
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_asset
    asset = make_synthetic_asset("smplx", seed=12)
    path = write_npz(asset, str(tmp_path / "SMPLX_NEUTRAL.npz"), "smplx_v1.0")
    bm = BodyModel(bm_path=str(tmp_path), num_betas=16, num_expressions=50).to(DEV)            # (a directory)
    assert (bm.bm.num_betas, bm.bm.num_expression_coeffs) == (10, 10)
    d, full, shape = _inputs("smplx", 5, 10, 10, seed=4)
    o = bm(**{k: torch.tensor(v, device=DEV) for k, v in d.items()})
    v_ref, _, _ = fk_ref.model_forward(asset, full.astype(np.float64), shape=shape.astype(np.float64), transl=d["trans"].astype(np.float64))
    assert np.abs(t2n(o.v) - v_ref).max() < 1e-5
    with pytest.raises(ValueError, match="shape coefficients"):
        bm(pose_body=torch.zeros(2, 63, device=DEV), betas=torch.zeros(2, 16, device=DEV))


def test_smplx_wrapper_from_a_model_file(tmp_path):
    from dposer_amd.body_model import constants
    from dposer_amd.body_model.smpl import SMPLX
    from dposer_amd.body_model.synthetic import make_synthetic_asset
    asset = make_synthetic_asset("smplx", seed=13)
    path = write_pkl(asset, str(tmp_path / "SMPLX_NEUTRAL.pkl"), "smplx_v1.1")
    m = SMPLX(path).to(DEV)
    rs = np.random.RandomState(0)
    pose = (rs.standard_normal((4, 63)) * 0.3).astype(np.float32)
    o = m(body_pose=torch.tensor(pose, device=DEV))
    _, j_ref, _, _ = fk_ref.smplx_forward(asset, pose.astype(np.float64), dtype=np.float64)
    jm = [constants.JOINT_MAP[i] for i in constants.JOINT_NAMES]
    jm[:25] = constants.SMPLX_OPENPOSE_25
    assert np.abs(t2n(o.joints) - j_ref[:, jm]).max() < 1e-5


# ---- more than four influences per vertex ------------------------------------------------------------------------------------------
@pytest.fixture(scope="module", params=[5, 8])
def wide(request):
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    asset = make_synthetic_smplx_asset(seed=20 + request.param, nnz_per_vertex=request.param)
    bm = BodyModel(asset).to(DEV)
    assert int(bm.bm.skin_idx.shape[1]) == request.param
    return asset, bm


@pytest.mark.parametrize("B", [3, 400])          # the small-batch kernels / the sizes where the 4-wide assets take the run / streaming kernels
def test_wide_skinning_forward_and_backward_vs_oracle(wide, B):
    from oracle import fk_torch
    asset, bm = wide
    rs = np.random.RandomState(B)
    pose = (rs.standard_normal((B, 63)) * 0.4).astype(np.float32)
    root = (rs.standard_normal((B, 3)) * 0.4).astype(np.float32)
    p = torch.tensor(pose, device=DEV, requires_grad=True)
    r = torch.tensor(root, device=DEV, requires_grad=True)
    o = bm(pose_body=p, root_orient=r)
    v_ref, j_ref, _, _ = fk_ref.smplx_forward(asset, pose.astype(np.float64), global_orient=root.astype(np.float64), dtype=np.float64)
    assert np.abs(t2n(o.v) - v_ref).max() < 1e-5 and np.abs(t2n(o.Jtr) - j_ref).max() < 1e-5
    with torch.no_grad():                                                         # the non-differentiable entry (dposer_lbs_forward on the cached workspace)
        o2 = bm(pose_body=p.detach(), root_orient=r.detach())
    assert torch.equal(o2.v, o.v.detach())
    nref = min(B, 24)                                                             # fp64 autograd reference on a slice (per-pose independent)
    wv = rs.standard_normal((B, 10475, 3)).astype(np.float32) / 50.0
    wj = rs.standard_normal((B, 127, 3)).astype(np.float32)
    wv[nref:] = 0
    wj[nref:] = 0
    ((o.v * torch.tensor(wv, device=DEV)).sum() + (o.Jtr * torch.tensor(wj, device=DEV)).sum()).backward()
    p_r = torch.tensor(pose[:nref], dtype=torch.float64, requires_grad=True)
    r_r = torch.tensor(root[:nref], dtype=torch.float64, requires_grad=True)
    v, j = fk_torch.smplx_forward(asset, p_r, global_orient=r_r)
    ((v * torch.tensor(wv[:nref], dtype=torch.float64)).sum() + (j * torch.tensor(wj[:nref], dtype=torch.float64)).sum()).backward()
    for name, got, want in (("pose", p.grad, p_r.grad), ("root", r.grad, r_r.grad)):
        err = rel_err(t2n(got)[:nref], want.numpy())
        _log_measured(f"wide skinning ({bm.bm.skin_idx.shape[1]} influences) backward, B = {B}, d {name}", err)
        assert err < 2e-4, (name, err)
        assert not t2n(got)[nref:].any()                                          # zero incoming gradient -> zero outgoing, no stray writes


def test_wide_skinning_through_the_one_call_motion_denoising_loop(wide, monkeypatch):
    """dposer_motion_denoise_optimize on an asset whose ELL width is not 4: the loop falls back to the general skinning kernels + the
    two-kernel temporal gradient (also when the fused form is forced by the A/B switch) and lands where the autograd loop lands."""
    from dposer_amd.dataset.AMASS import Posenormalizer
    from dposer_amd.tasks.motion_denoising import MotionDenoise
    asset, bm = wide
    cfg, m, p = make_model(63, precision="fp32")
    g = load("g10_normalizer")
    stats = {k.split("/")[-1]: torch.tensor(g[k]) for k in g.files if k.startswith("stats/axis_normalize")}
    F, S, iters, spi = 8, 4, 1, 3
    T = F * S
    gt = np.tile(g["raw"], (2, 1))[:T].astype(np.float32)
    rs = np.random.RandomState(1)
    init = (gt + rs.standard_normal(gt.shape) * 0.05).astype(np.float32)
    _, jgt, _, _ = fk_ref.smplx_forward(asset, gt.astype(np.float64), dtype=np.float64)
    joints3d = (jgt[:, :22] + rs.standard_normal((T, 22, 3)) * 0.04).astype(np.float32)

    class Args:
        device = DEV
    nz = Posenormalizer(stats, device=DEV, normalize=True, min_max=False, rot_rep="axis")
    md = MotionDenoise(cfg, Args(), m, bm, sde_N=500, batch_size=F, normalizer=nz)
    assert md._fused_supported()
    dev = lambda a: torch.tensor(a, device=DEV)
    noise = dev(rs.standard_normal((iters * spi, T, 63)).astype(np.float32))
    kw = dict(time_strategy="3", iterations=iters, steps_per_iter=spi)
    args = (dev(joints3d).reshape(S, F, 22, 3), dev(gt).reshape(S, F, 63))
    res = md.optimize_sequences(*args, noise=noise, init_poses=dev(init).reshape(S, F, 63), **kw)
    monkeypatch.setenv("DPOSER_MD_FUSED_TEMPORAL", "1")                           # forcing the 4-wide kernel must not refuse a wider asset
    res_forced = md.optimize_sequences(*args, noise=noise, init_poses=dev(init).reshape(S, F, 63), **kw)
    monkeypatch.delenv("DPOSER_MD_FUSED_TEMPORAL")
    assert torch.equal(res["pose_body"], res_forced["pose_body"])
    md.batch_size = F
    for i in range(S):
        sl = slice(i * F, (i + 1) * F)
        one = md.optimize(dev(joints3d[sl]), gt_poses=dev(gt[sl]), noise=noise[:, sl].contiguous(), init_poses=dev(init[sl]), fused=False, **kw)
        err = rel_err(t2n(res["pose_body"][i]), t2n(one["pose_body"]))
        assert err < 2e-5, (i, err)


# ---- the smplx pin (tests/golden/pin_fk_parity.py) -----------------------------------------------------------------------------------------
def _pin_cases():
    import os
    from helpers import GOLDEN
    path = os.path.join(GOLDEN, "g23_smplx_pin.npz")
    if not os.path.exists(path):
        pytest.skip("FK / LBS parity unpinned: tests/golden/g23_smplx_pin.npz is written by tests/golden/pin_fk_parity.py wherever smplx==0.1.28 can be installed")
    return np.load(path, allow_pickle=False)


def test_lbs_against_smplx_golden(tmp_path):
    """The HIP body model against outputs of the reference's own dependency (smplx objects built the way body_model.py:30-62 builds them)
    on files in the official layout.  Needs the golden written by tests/golden/pin_fk_parity.py; skips (loudly) until then."""
    from pin_cases import build_case
    g = _pin_cases()
    for i, layout in enumerate([str(c) for c in g["cases"]]):
        bm, kw, _ = build_case(g, i, layout, tmp_path)
        bm = bm.to(DEV)
        o = bm(**{k: torch.tensor(v, device=DEV) for k, v in kw.items()})
        ev, ej = np.abs(t2n(o.v) - g[f"{layout}/vertices"]).max(), np.abs(t2n(o.Jtr) - g[f"{layout}/joints"]).max()
        _log_measured(f"HIP LBS vs smplx golden, {layout}", max(ev, ej))
        assert ev < 1e-5 and ej < 1e-5, (layout, ev, ej)
