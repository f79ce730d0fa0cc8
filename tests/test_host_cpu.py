"""CPU tests of the host logic: index tables (bit-exact vs the reference's goldens), masks, normaliser,
config entry points, shard arithmetic, the C-ABI library's symbols, the flat-parameter module."""
import ctypes
import math
import os
import re
import sys

import numpy as np
import pytest
import torch

from helpers import load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PARTS = ["left_leg", "right_leg", "left_arm", "right_arm", "trunk", "hands", "legs", "arms"]


def test_index_tables_bit_exact():
    from dposer_amd.body_model import constants, utils
    g = load("g9_tables")
    for part in PARTS:
        assert np.array_equal(np.array(getattr(utils.BodyPartIndices, part)), g[f"part/{part}"])
        assert np.array_equal(np.array(getattr(utils.BodySegIndices, part)), g[f"seg/{part}"])
    assert list(g["joint_names"]) == constants.JOINT_NAMES
    assert np.array_equal(np.array([constants.JOINT_MAP[n] for n in constants.JOINT_NAMES]), g["joint_map"])
    for mt in ("smpl", "smplh", "smplx"):
        assert np.array_equal(utils.smpl_to_openpose(mt), g[f"openpose/{mt}"])
    assert np.array_equal(utils.skeleton_parents(22), g["parents22"])
    assert np.array_equal(utils.get_smpl_skeleton(), g["skeleton"])
    for perm in ("SMPL_JOINTS_FLIP_PERM", "SMPL_POSE_FLIP_PERM", "J24_FLIP_PERM", "J49_FLIP_PERM", "H36M_TO_J17", "H36M_TO_J14",
                 "J24_TO_J17", "J24_TO_J14"):
        assert np.array_equal(np.array(getattr(constants, perm)), g[f"const/{perm}"])
    from dposer_amd.body_model.synthetic import SMPLX_PARENTS
    assert np.array_equal(SMPLX_PARENTS[:22], g["parents22"])


def test_smpl_to_openpose_every_argument_combination_bit_exact():
    """lib/body_model/utils.py:68-177 over model_type x use_hands x use_face x use_face_contour x format (golden g24 = the
    reference's own outputs), plus its error behaviour."""
    from dposer_amd.body_model import utils
    g = load("g24_openpose_maps")
    assert len(g.files) == 48
    for key in g.files:
        fmt, mt, flags = key.split("/")
        got = utils.smpl_to_openpose(mt, use_hands=flags[0] == "1", use_face=flags[1] == "1", use_face_contour=flags[2] == "1", openpose_format=fmt)
        assert got.dtype == np.int32 and np.array_equal(got, g[key]), key
    assert np.array_equal(utils.smpl_to_openpose("smplx", openpose_format="COCO25"), g["coco25/smplx/110"])
    with pytest.raises(ValueError, match="Unknown model type"):
        utils.smpl_to_openpose("mano")
    with pytest.raises(ValueError, match="Unknown joint format"):
        utils.smpl_to_openpose("smplx", openpose_format="COCO19")
    with pytest.raises(ValueError, match="Unknown joint format"):
        utils.smpl_to_openpose("smplx", openpose_format="coco17")


def test_smplx_joint_map_bit_exact():
    from dposer_amd.body_model import constants
    g = load("g9_tables")
    jm = [constants.JOINT_MAP[i] for i in constants.JOINT_NAMES]
    jm[:25] = constants.SMPLX_OPENPOSE_25
    assert np.array_equal(np.array(jm), g["smplx_joint_map"])


@pytest.mark.parametrize("rot_n", [3, 6])
def test_create_mask(rot_n):
    from dposer_amd.utils.misc import create_mask
    g = load("g9_tables")
    for part in PARTS:
        poses = torch.zeros(4, 21 * rot_n)
        mask, obs = create_mask(poses, part=part)
        assert np.array_equal(mask[0].numpy().astype(np.uint8), g[f"mask/{part}/{rot_n}"])
        assert torch.equal(obs * mask, poses * mask)


def test_eval_sampler_shards():
    from dposer_amd.dataset.EvaSampler import DistributedEvalSampler
    g = load("g9_tables")

    class DS:
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

    for total, world in ((103, 4), (16, 8), (7, 2), (100, 1)):
        seen = []
        for rank in range(world):
            s = DistributedEvalSampler(DS(total), num_replicas=world, rank=rank, shuffle=False)
            idx = np.array(list(iter(s)), dtype=np.int64)
            assert np.array_equal(idx, g[f"eva/{total}/{world}/{rank}"])
            assert len(s) == len(idx)
            seen += list(idx)
        assert sorted(seen) == list(range(total))


def test_normalizer_and_misc():
    from dposer_amd.dataset.AMASS import Posenormalizer
    from dposer_amd.utils import misc
    g = load("g10_normalizer")
    stats = {k.split("/")[-1]: torch.tensor(g[k]) for k in g.files if k.startswith("stats/axis_normalize")}
    raw = torch.tensor(g["raw"])
    for mm in (False, True):
        nz = Posenormalizer(stats, device="cpu", normalize=True, min_max=mm, rot_rep="axis")
        n = nz.offline_normalize(raw)
        assert np.array_equal(n.numpy(), g[f"norm_minmax{int(mm)}"])
        assert np.array_equal(nz.offline_denormalize(n).numpy(), g[f"denorm_minmax{int(mm)}"])
        assert np.array_equal(nz.offline_normalize(raw.reshape(4, 16, 63)).numpy(), g[f"norm3d_minmax{int(mm)}"])
    g2 = load("g11_rot6d")
    ts = torch.linspace(0.1, 0.9, 7)
    assert np.array_equal(misc.linear_interpolation(ts, ts / 10, 6).numpy(), g2["lin_interp"])
    assert np.allclose(misc.gaussian_smoothing(torch.tensor(g2["smooth_in"]), 5, 2.0).numpy(), g2["smooth_out"], atol=1e-6)


def test_config_entry_points():
    from dposer_amd.configs import load_config
    for spec in ("configs/subvp/amass_scorefc_continuous.py", "configs.subvp.amass_scorefc_continuous.get_config"):
        c = load_config(spec)
        assert (c.model.HIDDEN_DIM, c.model.EMBED_DIM, c.model.N_BLOCKS, c.model.dropout) == (1024, 512, 2, 0.1)
        assert (c.training.sde, c.sampling.predictor, c.sampling.corrector) == ("subvpsde", "euler_maruyama", "none")
        assert (c.optim.lr, c.optim.warmup, c.optim.grad_clip, c.training.batch_size) == (2e-4, 5000, 1.0, 1280)


def test_capi_exports_every_declared_symbol():
    from dposer_amd import _C
    hdr = open(os.path.join(ROOT, "include", "dposer_hip.h")).read()
    declared = set(re.findall(r"\b(dposer_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = ctypes.CDLL(_C.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/dposer_hip.h but not exported"
    assert declared == set(_C.SIGNATURES), declared ^ set(_C.SIGNATURES)
    assert _C.lib().dposer_abi_version() == 1


def test_shipped_library_has_no_dropout_mask_hook():
    """The injected-dropout-mask test hook lives in the test-hook build only (libdposer_hip_testhooks.so, -DDPOSER_TEST_HOOKS): the
    shipped library keeps the symbol (one ABI) but refuses masks, and its training epilogues reference no `ext_keep` load -- both
    libraries export the same symbols."""
    import ctypes as C
    from dposer_amd import _C
    lib = _C.lib()
    h = C.c_void_p()
    d = _C.ScoreFCDesc(63, 1024, 512, 2, _C.EMB_POSITIONAL, 1, 1000, _C.PREC_BF16, 0.1)
    assert lib.dposer_scorefc_create(C.byref(d), C.byref(h)) == 0
    fake = C.c_void_p(0x1000)
    assert lib.dposer_scorefc_debug_set_dropout_masks(h, fake, 32) < 0 and b"DPOSER_TEST_HOOKS" in lib.dposer_last_error()
    assert lib.dposer_scorefc_debug_set_dropout_masks(h, None, 0) == 0           # clearing is always fine
    lib.dposer_scorefc_destroy(h)
    th = os.path.join(os.path.dirname(_C.LIB_PATH), "libdposer_hip_testhooks.so")
    assert os.path.exists(th), "make -C dposer_amd/csrc builds it beside libdposer_hip.so"
    hooks = ctypes.CDLL(th)
    for name in _C.SIGNATURES:
        assert hasattr(hooks, name), name


def test_model_surface_and_flat_params():
    from dposer_amd import _C
    from dposer_amd.algorithms.advanced.model import ScoreModelFC
    from dposer_amd.algorithms.ema import ExponentialMovingAverage
    from dposer_amd.configs import load_config
    from weights import scorefc_shapes
    cfg = load_config("configs.subvp.amass_scorefc_continuous.get_config")
    m = ScoreModelFC(cfg, n_poses=21, pose_dim=3, hidden_dim=1024, embed_dim=512, n_blocks=2)
    sd = m.state_dict()
    expect = dict(scorefc_shapes())
    expect["sigmas"] = (1000,)
    assert {k: tuple(v.shape) for k, v in sd.items()} == expect
    assert [n for n, _ in m.named_parameters()] == [n for n, _ in scorefc_shapes()]
    assert sum(p.numel() for p in m.parameters()) == 8277567
    flat = m.flat_params()
    assert flat.numel() == 8277567 and all(p.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for p in m.parameters())
    m.post_dense.bias.data.fill_(3.0)                     # in-place writes go through to the flat buffer
    assert float(flat[-63:].min()) == 3.0
    ema = ExponentialMovingAverage(m.parameters(), 0.9999)
    assert len(ema.shadow_params) == 36 and ema.flat_shadow_for(flat) is not None
    sd2 = {k: torch.zeros_like(v) for k, v in sd.items()}
    m.load_state_dict(sd2)                                # reference checkpoints load into the views
    assert float(m.flat_params().abs().sum()) == 0.0
    with pytest.raises(_C.DPoserHipError):                # no CPU fallback
        m(torch.zeros(4, 63), torch.zeros(4))
    with pytest.raises(_C.DPoserHipError):
        ScoreModelFC(cfg, hidden_dim=64, embed_dim=32)._engine()


def test_reference_import_paths():
    import dposer_amd
    dposer_amd.install_reference_aliases()
    from lib.algorithms.advanced import likelihood, losses, sampling, sde_lib  # noqa: F401
    from lib.algorithms.advanced import utils as mutils  # noqa: F401
    from lib.utils import metric  # noqa: F401
    assert callable(likelihood.get_likelihood_fn) and callable(sampling.get_ode_sampler)
    from lib.algorithms.advanced.model import ScoreModelFC  # noqa: F401
    from lib.algorithms.ema import ExponentialMovingAverage  # noqa: F401
    from lib.body_model.body_model import BodyModel  # noqa: F401
    from lib.dataset.AMASS import N_POSES, Posenormalizer  # noqa: F401
    from lib.utils.misc import create_mask  # noqa: F401
    from lib.utils.generic import import_configs
    assert import_configs("configs.subvp.amass_scorefc_continuous.get_config").model.HIDDEN_DIM == 1024
    assert sampling.get_predictor("euler_maruyama") is sampling.EulerMaruyamaPredictor


def test_average_pairwise_distance_matches_reference_golden():
    from dposer_amd.utils.metric import average_pairwise_distance, self_intersections_percentage
    g = load("g12_likelihood_ode")
    j = torch.tensor(g["apd/joints"])
    assert abs(float(average_pairwise_distance(j)) - float(g["apd/value"])) < 1e-6
    assert abs(float(average_pairwise_distance(j, chunk=3)) - float(g["apd/value"])) < 1e-6     # chunking is invisible
    assert np.isnan(self_intersections_percentage(np.zeros((2, 4, 3)), np.zeros((2, 3), dtype=np.int64))).all()


def test_philox_known_answer():
    # Random123 known-answer test for Philox4x32-10: counter = key = 0, and the all-ones / pi vectors
    from oracle.philox import philox4x32_10
    r = philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(v) for v in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = philox4x32_10(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)
    assert [int(v) for v in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)
    assert [int(v) for v in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_amass_dataset_matches_reference_golden(tmp_path):
    """The AMASS .pt data format: sampling, statistics written / re-read, both normalisations, Denormalize (axis-angle; the
    rot6d variant needs the GPU rotation kernels)."""
    from dposer_amd.dataset.AMASS import AMASSDataset
    g = load("g13_dataset")
    toy = torch.tensor(load("g10_normalizer")["toy_pose_samples"])
    betas = torch.tensor(g["betas"])
    for sub in ("train", "valid"):
        os.makedirs(tmp_path / "v" / sub)
        torch.save(toy if sub == "train" else toy[:100] * 0.5, tmp_path / "v" / sub / "pose_body.pt")
        torch.save(betas if sub == "train" else betas[:100], tmp_path / "v" / sub / "betas.pt")
    for tag, mm in (("minmax", True), ("zscore", False)):
        ds = AMASSDataset(str(tmp_path), version="v", subset="train", sample_interval=3, rot_rep="axis", return_shape=True, normalize=True,
                          min_max=mm)
        assert len(ds) == int(g[f"{tag}/len"])
        assert np.array_equal(ds.poses.numpy(), g[f"{tag}/poses"]) and np.array_equal(ds.shapes.numpy(), g[f"{tag}/shapes"])
        assert np.array_equal(ds[7]["poses"].numpy(), g[f"{tag}/item7_poses"])
        assert np.array_equal(ds.Denormalize(ds.poses[:5]).numpy(), g[f"{tag}/denorm"])
        dp, dsh = ds.Denormalize(ds.poses[None, :5], ds.shapes[None, :5])
        assert np.array_equal(dp.numpy(), g[f"{tag}/denorm3_poses"]) and np.array_equal(dsh.numpy(), g[f"{tag}/denorm3_shapes"])
        dv = AMASSDataset(str(tmp_path), version="v", subset="valid", rot_rep="axis", normalize=True, min_max=mm)
        assert np.array_equal(dv.poses.numpy(), g[f"{tag}/valid_poses"])
        assert os.path.exists(tmp_path / "v" / "train" / ("axis_normalize1.pt" if mm else "axis_normalize2.pt"))


def test_capi_handle_layout_and_error_codes():
    """Host-only entry points (no GPU): parameter layout = ScoreModelFC.parameters() order, gradient buckets partition the flat
    buffer, argument errors come back as negative codes with a message in dposer_last_error()."""
    import ctypes as C
    from dposer_amd import _C
    from oracle import score_ref as R
    lib = _C.lib()
    h = C.c_void_p()
    bad = _C.ScoreFCDesc(63, 1000, 512, 2, _C.EMB_POSITIONAL, 1, 1000, _C.PREC_BF16, 0.1)
    rc = lib.dposer_scorefc_create(C.byref(bad), C.byref(h))
    assert rc < 0 and b"hidden_dim" in lib.dposer_last_error()
    with pytest.raises(_C.DPoserHipError, match="hidden_dim"):
        _C.check(rc, "dposer_scorefc_create")
    assert lib.dposer_scorefc_create(None, C.byref(h)) < 0
    for width, ok in ((256, False), (512, True), (2048, True), (4096, False)):            # GroupNorm(32, H): groups of 16 / 32 / 64 channels are built
        d = _C.ScoreFCDesc(63, width, 512, 2, _C.EMB_POSITIONAL, 1, 1000, _C.PREC_BF16, 0.1)
        hh = C.c_void_p()
        rc = lib.dposer_scorefc_create(C.byref(d), C.byref(hh))
        assert (rc == 0) == ok, width
        if ok:
            lib.dposer_scorefc_destroy(hh)
        else:
            assert b"hidden_dim" in lib.dposer_last_error()

    good = _C.ScoreFCDesc(63, 1024, 512, 2, _C.EMB_POSITIONAL, 1, 1000, _C.PREC_BF16, 0.1)
    assert lib.dposer_scorefc_create(C.byref(good), C.byref(h)) == 0
    try:
        assert lib.dposer_scorefc_num_params(h) == 8277567
        shapes = {"pre_dense.weight": 1024 * 63, "pre_dense_cond.weight": 1024 * 1024, "shared_time_embed.0.weight": 512 * 512,
                  "b1_dense1_t.weight": 1024 * 512, "b2_gnorm2.bias": 1024, "post_dense.weight": 63 * 1024, "post_dense.bias": 63}
        names = R.param_names()
        assert lib.dposer_scorefc_num_tensors(h) == len(names) == 36
        off = 0
        for i, n in enumerate(names):
            assert lib.dposer_scorefc_tensor_offset(h, i) == off
            if n in shapes:
                assert lib.dposer_scorefc_tensor_numel(h, i) == shapes[n], n
            off += lib.dposer_scorefc_tensor_numel(h, i)
        assert off == 8277567
        lo, hi = (C.c_int64 * 2)(), (C.c_int64 * 2)()
        assert lib.dposer_scorefc_nograd_ranges(h, lo, hi) == 1                  # the dead pre_dense_cond
        i_cond = names.index("pre_dense_cond.weight")
        assert lo[0] == lib.dposer_scorefc_tensor_offset(h, i_cond) and hi[0] - lo[0] == 1024 * 1024 + 1024
        nb = lib.dposer_scorefc_grad_buckets(h, None, None, 0)
        blo, bhi = (C.c_int64 * nb)(), (C.c_int64 * nb)()
        assert lib.dposer_scorefc_grad_buckets(h, blo, bhi, nb) == nb == 6          # layers 4..1, front A, front B
        assert bhi[0] == 8277567 and blo[4] == 0 and all(bhi[b + 1] == blo[b] for b in range(3))
        assert bhi[4] == lo[0] and blo[5] == hi[0] and bhi[5] == blo[3]            # the dead range lies between front A and front B: in no bucket
        assert blo[0] == lib.dposer_scorefc_tensor_offset(h, names.index("b2_dense2.weight"))
        assert lib.dposer_scorefc_workspace_bytes(h, 0, _C.WS_TRAIN, 0) < 0       # batch must be positive
        assert lib.dposer_scorefc_workspace_bytes(h, 65536, _C.WS_TRAIN, 0) > lib.dposer_scorefc_workspace_bytes(h, 65536, _C.WS_INFER, 0) > 0
        # NULL tensors are rejected before anything is launched
        assert lib.dposer_scorefc_forward(h, None, None, None, None, None, None, None, None, 8, None) < 0
        assert b"null" in lib.dposer_last_error()
    finally:
        lib.dposer_scorefc_destroy(h)


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """The structs of include/dposer_hip.h as the C compiler lays them out (gcc, the header is plain C) against their ctypes
    mirrors in dposer_amd/_C.py: size and the offset of every field."""
    import ctypes as C
    import shutil
    import subprocess
    from dposer_amd import _C
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    pairs = {"dposer_scorefc_desc": _C.ScoreFCDesc, "dposer_sde_desc": _C.SdeDesc, "dposer_body_desc": _C.BodyDesc,
             "dposer_motion_denoise_args": _C.MotionDenoiseArgs, "dposer_lbs_joint_fold": _C.LbsJointFold, "dposer_mlp_desc": _C.MlpDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "dposer_hip.h"', 'int main(void) {']
    for cname, ct in pairs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "abi_probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "abi_probe"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = {}
    for ln in subprocess.check_output([str(exe)], text=True).splitlines():
        a, b, c = ln.split()
        got[(a, b)] = int(c)
    for cname, ct in pairs.items():
        assert got[(cname, "size")] == C.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert got[(cname, fname)] == getattr(ct, fname).offset, (cname, fname)


def test_device_rk45_driver_reproduces_scipy_step_for_step():
    """ode_device.solve_rk45 is scipy's RK45 controller on torch tensors: on the same right-hand side it must make the same
    accept / reject decisions (identical nfev) and land on the same state, forwards and backwards in time, including a stiff-ish
    problem that forces rejected steps."""
    from scipy import integrate
    from dposer_amd.algorithms.advanced.ode_device import solve_fixed, solve_rk45
    rs = np.random.RandomState(0)
    A = torch.tensor(rs.standard_normal((12, 12)) * 0.7, dtype=torch.float64)

    def fun_t(t, y):
        return torch.tanh(A @ y) * (1.0 + 3.0 * math.sin(5 * t)) - 0.3 * y ** 3

    def fun_np(t, y):
        return fun_t(t, torch.from_numpy(y)).numpy()

    y0 = rs.standard_normal(12)
    for (a, b), rtol, atol in (((0.0, 2.0), 1e-5, 1e-5), ((0.6, 0.2), 1e-4, 1e-4), ((0.0, 6.0), 1e-8, 1e-10)):
        sol = integrate.solve_ivp(fun_np, (a, b), y0, rtol=rtol, atol=atol, method="RK45")
        y, nfev = solve_rk45(fun_t, a, b, torch.from_numpy(y0), rtol=rtol, atol=atol)
        assert nfev == sol.nfev, (a, b, nfev, sol.nfev)
        assert np.abs(y.numpy() - sol.y[:, -1]).max() < 1e-12
    # fixed-step variant: RK4 converges with order 4 to the adaptive solution
    ref = integrate.solve_ivp(fun_np, (0.0, 1.0), y0, rtol=1e-11, atol=1e-12, method="RK45").y[:, -1]
    e1 = np.abs(solve_fixed(fun_t, 0.0, 1.0, torch.from_numpy(y0), 40)[0].numpy() - ref).max()
    e2 = np.abs(solve_fixed(fun_t, 0.0, 1.0, torch.from_numpy(y0), 80)[0].numpy() - ref).max()
    assert e2 < e1 / 12 and e2 < 1e-6


def test_product_never_touches_the_oracle_or_the_reference():
    """The oracle is test infrastructure: nothing under dposer_amd/ may import it (or read /root/reference), and bench.py /
    __graft_entry__.py may use it only in cpu_baseline / smoke."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle|from\s+\.+oracle)|/root/reference", re.M)
    for dirpath, _, files in os.walk(os.path.join(root, "dposer_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not pat.search(src), os.path.join(dirpath, f)
    bench = open(os.path.join(root, "bench.py")).read()
    body = bench.split("def cpu_baseline", 1)[1].split("\ndef main", 1)[0]
    assert "oracle" in body                                              # the CPU leg is the oracle ...
    assert "oracle" not in bench.replace(body, "").split('"""', 2)[2]    # ... and nothing else in bench.py is (docstring aside)
    entry = open(os.path.join(root, "__graft_entry__.py")).read()
    build_body = entry.split("def build", 1)[1].split("def smoke", 1)[0]
    assert "score_ref." not in build_body and "fk_ref." not in build_body   # build() imports the oracle modules, never calls them


def test_rot6d_to_axis_angle_autograd_helper_vs_scipy():
    """The differentiable 6D -> axis-angle map of the auxiliary-loss step (rot_rep = 'rot6d', losses.py:247-249) is plain torch:
    values against scipy, gradient against finite differences (CPU)."""
    from scipy.spatial.transform import Rotation
    from dposer_amd.algorithms.advanced.losses import _rot6d_to_axis_angle_autograd as f
    rs = np.random.RandomState(2)
    rv = rs.standard_normal((40, 3)) * 0.8
    rv[0] = 1e-6                                                # small-angle branch
    R = Rotation.from_rotvec(rv).as_matrix()
    x = torch.tensor(R[:, :, :2].reshape(-1, 6) * rs.uniform(0.5, 2.0, (40, 1)), dtype=torch.float64, requires_grad=True)
    out = f(x)
    assert np.abs(out.detach().numpy() - Rotation.from_matrix(R).as_rotvec()).max() < 1e-6
    assert torch.autograd.gradcheck(f, (x[1:9],), eps=1e-6, atol=1e-5)
    # the whole range of angles, in fp32 as the training step runs it: tiny angles, angles up to and AT pi (where a log map through
    # acos / (2 sin) returns ~0 instead of pi n -- the reference's quaternion route is exact there), every axis octant
    axes = rs.standard_normal((64, 3))
    axes /= np.linalg.norm(axes, axis=1, keepdims=True)
    for ang in (1e-7, 1e-5, 3e-4, 0.1, 1.0, 2.5, 3.0, 3.1, 3.14, np.pi - 1e-4, np.pi - 1e-6, np.pi):
        rv = axes * ang
        Rm = Rotation.from_rotvec(rv).as_matrix()
        got = f(torch.tensor(Rm[:, :, :2].reshape(-1, 6), dtype=torch.float32)).numpy().astype(np.float64)
        # compare as rotations (at pi the vectors pi n and -pi n are the same rotation)
        err = np.abs(Rotation.from_rotvec(got).as_matrix() - Rm).max()
        assert err < 2e-3 if ang > 3.1 else err < 2e-5, (ang, err)
        assert np.abs(np.linalg.norm(got, axis=1) - ang).max() < (2e-3 if ang > 3.1 else 1e-5), ang
    x64 = torch.tensor(Rotation.from_rotvec(axes[:6] * 3.05).as_matrix()[:, :, :2].reshape(-1, 6), dtype=torch.float64, requires_grad=True)
    assert torch.autograd.gradcheck(f, (x64,), eps=1e-6, atol=1e-4)           # the near-pi branch is differentiable too


def test_dropout_decisions_philox7_statistics():
    """The dropout streams run Philox4x32-7 (rng.h, oracle/philox.py): keep rate, independence across channels / samples / sites /
    steps of the decisions the kernels draw (the 10-round generator is pinned by the Random123 vectors above; 7 rounds are its first 7)."""
    from oracle.philox import dropout_keep_mask
    B, H, p = 4096, 1024, 0.1
    m = dropout_keep_mask(B, H, 0, 5, 42, p)
    n = m.size
    sigma = np.sqrt(p * (1 - p) / n)
    assert abs(m.mean() - (1 - p)) < 5 * sigma + 2e-5                      # thr = floor(0.9 * 65536): keep rate 0.89999
    assert np.abs(m.mean(axis=0) - 0.9).max() < 6 * np.sqrt(0.09 / B)       # every channel
    assert np.abs(m.mean(axis=1) - 0.9).max() < 6 * np.sqrt(0.09 / H)       # every sample
    c = m - m.mean()
    for a, b in ((c[:, :-1], c[:, 1:]), (c[:-1], c[1:]), (c[:, :-32], c[:, 32:]), (c[:, ::2], c[:, 1::2])):
        assert abs((a * b).mean() / 0.09) < 5 / np.sqrt(a.size)             # lag correlations: neighbours, next group, lane pairs
    for other in (dropout_keep_mask(B, H, 1, 5, 42, p), dropout_keep_mask(B, H, 0, 6, 42, p), dropout_keep_mask(B, H, 0, 5, 43, p)):
        assert abs(((other - other.mean()) * c).mean() / 0.09) < 5 / np.sqrt(n)   # other site / step / seed: uncorrelated
        assert (other != m).mean() > 0.15


def test_generated_wgrad_k_loop_is_in_sync_with_its_generator():
    """dposer_amd/csrc/gemm_wgrad_tr_asm.inc (the hand-placed K loop of the sample-major wgrad kernel, one asm statement) is the committed
    output of tools/gen_wgrad_asm.py; every ring stage issues its 16 MFMAs, 24 transposing reads and -- when it fetches -- 4 DMA pieces."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "gen_wgrad_asm.py")], capture_output=True, text=True, check=True).stdout
    with open(os.path.join(root, "dposer_amd", "csrc", "gemm_wgrad_tr_asm.inc")) as f:
        assert f.read() == out
    stages = out.split("// ---- stage on slot")[1:]
    assert len(stages) == 3 + 4 + 3                                        # leading stages 1..3, the group of four, the three tail stages
    for st in stages:
        mode = int(st.split("mode")[1].split()[0])
        assert st.count("v_mfma_f32_32x32x16_bf16") == 16
        assert st.count("ds_read_b64_tr_b16") == (12 if mode == 3 else 24)
        assert st.count("global_load_lds_dwordx4") == (4 if mode == 0 else 0)
        assert st.count("s_barrier") == (0 if mode == 3 else 1)


def test_emitted_isa_passes_the_asm_audits():
    """tools/check_isa.py on the ISA hipcc emits for EVERY kernel of the library (cross-compiles the six .hip files in parallel, ~1-2 min).
    The compiler cannot see inside the hand-placed asm statements, so (1) nothing it puts between two neighbouring K-loop stage
    statements may touch an accumulator register and those kernels must not spill, and (2) M0 -- compiler-reserved, "m0" clobbers are
    not honoured -- must have been written by the compiler itself in front of every compiler-issued M0 reader on every path, and every
    asm-issued reader must follow an M0 write of its own statement."""
    import shutil
    import subprocess
    import sys
    if shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_isa.py")], capture_output=True, text=True, timeout=1200)
    lines = [l for l in r.stdout.splitlines() if l.startswith(("ok", "FAIL"))]
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert len(lines) >= 100 and not any(l.startswith("FAIL") for l in lines)
    assert sum("stage statements" in l and l.split("stage statements")[0].split()[-1].isdigit() for l in lines) >= 20
    assert any("k_fk_joints_dma" in l for l in lines) and any("gemm_wgrad_tr_batch_kernel" in l for l in lines)


def test_isa_audit_catches_a_stale_m0():
    """The M0 analysis on hand-made ISA: a compiler-issued global_load_lds behind an asm statement that wrote M0 is flagged, on a
    straight line and through a branch; a compiler s_mov m0 in between, or an asm statement that saves / restores M0, clears it."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(root, "tools", "check_isa.py"))
    ci = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ci)
    asm_w = ";;#ASMSTART\n\ts_mov_b32 m0, s4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v1, off\n;;#ASMEND\n"
    asm_keep = ";;#ASMSTART\n\ts_mov_b32 s9, m0\n\ts_mov_b32 m0, s4\n\tglobal_load_lds_dwordx4 v1, off\n\ts_mov_b32 m0, s9\n;;#ASMEND\n"
    rd = "\tglobal_load_lds_dwordx4 v[2:3], off\n"
    mv = "\ts_mov_b32 m0, s7\n"
    assert ci.audit_m0(mv + rd)[0] == []
    assert len(ci.audit_m0(mv + asm_w + rd)[0]) == 1
    assert ci.audit_m0(mv + asm_w + mv + rd)[0] == []
    assert ci.audit_m0(mv + asm_keep + rd)[0] == []
    branchy = mv + "\ts_cbranch_scc1 .LBB0_2\n" + asm_w + ".LBB0_2:\n" + rd          # one of the two paths into the reader is stale
    assert len(ci.audit_m0(branchy)[0]) == 1
    loop = mv + ".LBB0_1:\n" + rd + asm_w + "\ts_cbranch_scc1 .LBB0_1\n"             # stale on the back edge
    assert len(ci.audit_m0(loop)[0]) == 1
    bare = ";;#ASMSTART\n\tglobal_load_lds_dwordx4 v1, off\n;;#ASMEND\n"            # asm reader without an M0 write of its own
    assert len(ci.audit_m0(mv + bare)[0]) == 1


def test_capi_argument_paths_under_address_sanitizer():
    """the host side of the C ABI -- argument checks, handle construction (parameter layout, pack-job and optimizer tables,
    bucket ranges), error strings, struct layouts -- built with -fsanitize=address (device code untouched, `make asan`) and driven by
    the C-API tests of this file in a child process with the ASan runtime preloaded.  Sanitizers run on the CPU build only."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    if os.environ.get("DPOSER_ASAN_CHILD"):
        pytest.skip("child process")
    rt = subprocess.run([hipcc, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        cand = [os.path.join(dp, f) for dp, _, fs in os.walk("/opt/rocm/lib/llvm/lib/clang") for f in fs if f == "libclang_rt.asan-x86_64.so"]
        if not cand:
            pytest.skip("no ASan runtime in this toolchain")
        rt = cand[0]
    csrc = os.path.join(ROOT, "dposer_amd", "csrc")
    b = subprocess.run(["make", "-C", csrc, "-j", str(min(8, os.cpu_count() or 1)), "asan"], capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, PATH=os.environ.get("PATH", "") + ":/opt/rocm/bin"))
    assert b.returncode == 0, b.stdout[-2000:] + b.stderr[-2000:]
    lib = os.path.join(csrc, "build", "asan", "libdposer_hip_asan.so")
    env = dict(os.environ, DPOSER_LIB_PATH=lib, LD_PRELOAD=rt, DPOSER_ASAN_CHILD="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23:protect_shadow_gap=0:verify_asan_link_order=0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider", "-k",
                        "capi_exports_every_declared_symbol or capi_handle_layout_and_error_codes or ctypes_structs_match"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert "AddressSanitizer" not in r.stdout + r.stderr, (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_profile_summariser_names_the_precision_modes_apart():
    """tools/rocpd_summary.py folds kernel names into `gemm_ft_kernel<precision,tile,epilogue>`: the bf16x3 instantiations (bf16 operand
    planes under a fp32-storage epilogue) must not share a name with the bf16 ones -- the bench looks its `traffic` / MFMA counters up by
    these names (a first round-5 pass averaged the two modes' launches together) -- and absent counters print as n/a, not as zero."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import rocpd_summary as rs
    finally:
        sys.path.pop(0)
    assert rs.short("void gemm_ft_kernel<__bf16, 2, 4, 4, 2, 2, EpiGN<__bf16, true, -1, false>, 4>(GemmArgs, GNParams)") == "gemm_ft_kernel<bf16,256x256,EpiGN<train>>"
    assert rs.short("void gemm_ft_kernel<__bf16, 2, 4, 4, 2, 2, EpiGN<float, true, -1, false>, 4>(GemmArgs, GNParams)") == "gemm_ft_kernel<bf16x3,256x256,EpiGN<train>>"
    assert rs.short("void gemm_ft_kernel<float, 2, 4, 4, 2, 2, EpiGNBwd<float, 0, false>, 4>(GemmArgs, GNBwdParams)") == "gemm_ft_kernel<fp32,256x256,EpiGNBwd>"
    assert rs.short("void gemm_ft_kernel<bool _Accum, int, E, 4, 2, 1, 2, EpiEmStep<float>, 4>(GemmArgs)") == "gemm_ft_kernel<bf16x3,64x128,EpiEmStep>"
