"""GPU parity tests of the body-model half: rotation conversions, FK joints, full LBS vs the numpy oracle
(oracle/fk_ref.py; smplx LBS parity is UNPINNED -- see oracle/__init__.py) on the synthetic SMPL-X-shaped asset."""
import numpy as np
import pytest
import torch

from gpu_common import DEV, t2n
from helpers import _log_measured, load
from oracle import fk_ref


def _reload_tuning():
    """the body-model A/B switches are read from the environment once per process: re-read them after changing one"""
    from dposer_amd import _C
    _C.lib().dposer_body_tuning_reload()

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def asset():
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    return make_synthetic_smplx_asset(seed=0)


@pytest.fixture(scope="module")
def bm(asset):
    from dposer_amd.body_model.body_model import BodyModel
    return BodyModel(asset, num_betas=10, num_expressions=10, model_type="smplx").to(DEV)


def _poses(B, seed=0, scale=0.4):
    return (np.random.RandomState(seed).standard_normal((B, 63)) * scale).astype(np.float32)


def test_rot6d_matches_reference_golden():
    from dposer_amd.utils.transforms import rot6d_to_mat3x3
    g = load("g11_rot6d")
    out = rot6d_to_mat3x3(torch.tensor(g["rot6d"], device=DEV))
    assert np.abs(t2n(out) - g["rotmat"]).max() < 2e-6
    assert rot6d_to_mat3x3(torch.zeros(0, 6, device=DEV)).shape == (0, 3, 3)


@pytest.mark.parametrize("n", [1, 255, 256, 257, 100000])
def test_rodrigues_vs_oracle(n):
    from dposer_amd.utils.transforms import batch_rodrigues
    rs = np.random.RandomState(n)
    aa = (rs.standard_normal((n, 3)) * 1.5).astype(np.float32)
    aa[0] = 0.0                                                  # the singular direction the 1e-8 offset guards
    out = t2n(batch_rodrigues(torch.tensor(aa, device=DEV)))
    ref = fk_ref.batch_rodrigues(aa.astype(np.float64))
    assert np.abs(out - ref).max() < 1e-5
    assert np.abs(out @ out.transpose(0, 2, 1) - np.eye(3)).max() < 1e-5   # orthonormal


@pytest.mark.parametrize("B", [1, 63, 64, 65, 1000])
def test_fk_body_joints_vs_oracle(bm, asset, B):
    pose = _poses(B, seed=B)
    j = t2n(bm.fk_joints(torch.tensor(pose, device=DEV), n_joints=22))
    _, ref, _, _ = fk_ref.smplx_forward(asset, pose.astype(np.float64), dtype=np.float64)
    assert j.shape == (B, 22, 3)
    assert np.abs(j - ref[:, :22]).max() < 1e-5                 # north_star: fp32 pose/vertex error <= 1e-5


def test_fk_all_segments_transl_and_55_joints(bm, asset):
    B = 130
    rs = np.random.RandomState(9)
    segs = dict(global_orient=rs.standard_normal((B, 3)) * 0.5, body_pose=_poses(B, 1), jaw_pose=rs.standard_normal((B, 3)) * 0.2,
                leye_pose=rs.standard_normal((B, 3)) * 0.1, reye_pose=rs.standard_normal((B, 3)) * 0.1,
                left_hand_pose=rs.standard_normal((B, 45)) * 0.3, right_hand_pose=rs.standard_normal((B, 45)) * 0.3)
    transl = rs.standard_normal((B, 3))
    dev = {k: torch.tensor(v.astype(np.float32), device=DEV) for k, v in segs.items()}
    out = bm.bm(transl=torch.tensor(transl.astype(np.float32), device=DEV), joints_only=True, **dev)
    _, ref, _, _ = fk_ref.smplx_forward(asset, segs["body_pose"].astype(np.float64), transl=transl,
                                        **{k: v.astype(np.float64) for k, v in segs.items() if k != "body_pose"}, dtype=np.float64)
    assert np.abs(t2n(out.joints) - ref[:, :55]).max() < 2e-5


def test_fk_full_batch_property():
    """B = 65536: rows agree with the same rows evaluated in a small batch (bit-exact), zero pose gives rest joints."""
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    a = make_synthetic_smplx_asset(seed=0)
    bm = BodyModel(a).to(DEV)
    pose = torch.tensor(_poses(65536, 5), device=DEV)
    big = bm.fk_joints(pose)
    small = bm.fk_joints(pose[777:777 + 100].contiguous())
    assert torch.equal(big[777:877], small)
    rest = bm.fk_joints(torch.zeros(3, 63, device=DEV))
    assert np.abs(t2n(rest)[0] - (a["J_regressor"] @ a["v_template"])[:22]).max() < 1e-6


@pytest.mark.parametrize("B", [1, 7, 130])
def test_lbs_vertices_and_127_joints_vs_oracle(bm, asset, B):
    pose = _poses(B, seed=100 + B)
    with torch.no_grad():
        out = bm(pose_body=torch.tensor(pose, device=DEV))
    verts, joints, full_pose, _ = fk_ref.smplx_forward(asset, pose.astype(np.float64), dtype=np.float64)
    assert out.v.shape == (B, 10475, 3) and out.Jtr.shape == (B, 127, 3)
    assert np.abs(t2n(out.v) - verts).max() < 1e-5
    assert np.abs(t2n(out.Jtr) - joints).max() < 1e-5
    assert np.array_equal(t2n(out.full_pose), full_pose.astype(np.float32))
    assert np.array_equal(t2n(out.body_joints), t2n(out.Jtr)[:22])          # batch-axis slice quirk (body_model.py:95)


def test_lbs_with_betas_root_trans(bm, asset):
    B = 5
    rs = np.random.RandomState(4)
    pose, betas = _poses(B, 3), rs.standard_normal((B, 10)).astype(np.float32)
    root, trans = (rs.standard_normal((B, 3)) * 0.3).astype(np.float32), rs.standard_normal((B, 3)).astype(np.float32)
    with torch.no_grad():
        out = bm(pose_body=torch.tensor(pose, device=DEV), betas=torch.tensor(betas, device=DEV),
                 root_orient=torch.tensor(root, device=DEV), trans=torch.tensor(trans, device=DEV))
    verts, joints, _, _ = fk_ref.smplx_forward(asset, pose.astype(np.float64), betas=betas.astype(np.float64),
                                               global_orient=root.astype(np.float64), transl=trans.astype(np.float64), dtype=np.float64)
    assert np.abs(t2n(out.v) - verts).max() < 2e-5
    assert np.abs(t2n(out.Jtr) - joints).max() < 2e-5


def test_smplx_wrapper_joint_map(asset):
    from dposer_amd.body_model.smpl import SMPLX
    g = load("g9_tables")
    sm = SMPLX(asset).to(DEV)
    assert np.array_equal(sm.joint_map.numpy(), g["smplx_joint_map"])
    pose = torch.tensor(_poses(4, 8), device=DEV)
    with torch.no_grad():
        o = sm(body_pose=pose)
        full = sm.bm(body_pose=pose)
    assert o.joints.shape == (4, 49, 3)
    assert torch.equal(o.joints, full.joints[:, sm.joint_map.to(DEV)])


def test_body_model_backward_vs_torch_autograd(bm, asset):
    """d(loss)/d(pose_body, root_orient, betas, trans) of a random linear functional of vertices and the 127 joints:
    HIP LBS backward vs torch fp64 autograd on the restated algorithm (oracle/fk_torch.py)."""
    from oracle import fk_torch
    B = 6
    rs = np.random.RandomState(12)
    pose, root = _poses(B, 21), (rs.standard_normal((B, 3)) * 0.4).astype(np.float32)
    betas, trans = (rs.standard_normal((B, 10)) * 0.5).astype(np.float32), rs.standard_normal((B, 3)).astype(np.float32)
    wv = rs.standard_normal((B, 10475, 3)).astype(np.float32) / 100.0
    wj = rs.standard_normal((B, 127, 3)).astype(np.float32)
    dev = lambda a: torch.tensor(a, device=DEV, requires_grad=True)
    p_d, r_d, b_d, t_d = dev(pose), dev(root), dev(betas), dev(trans)
    out = bm(pose_body=p_d, root_orient=r_d, betas=b_d, trans=t_d)
    loss = (out.v * torch.tensor(wv, device=DEV)).sum() + (out.Jtr * torch.tensor(wj, device=DEV)).sum()
    loss.backward()
    ref = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    p_r, r_r, b_r, t_r = ref(pose), ref(root), ref(betas), ref(trans)
    v, j = fk_torch.smplx_forward(asset, p_r, betas=b_r, global_orient=r_r, transl=t_r)
    lref = (v * torch.tensor(wv, dtype=torch.float64)).sum() + (j * torch.tensor(wj, dtype=torch.float64)).sum()
    lref.backward()
    assert abs(float(loss.detach()) - float(lref.detach())) / abs(float(lref.detach())) < 1e-5
    for name, got, want in (("pose", p_d.grad, p_r.grad), ("root", r_d.grad, r_r.grad), ("betas", b_d.grad, b_r.grad), ("trans", t_d.grad, t_r.grad)):
        err = float(np.linalg.norm(t2n(got) - want.numpy()) / np.linalg.norm(want.numpy()))
        assert err < 2e-4, (name, err)


def test_body_model_backward_pose_only_shared_shape(bm, asset):
    """The motion-denoising configuration: betas constant (shared v_shaped), gradient w.r.t. pose_body only, data term on
    Jtr[:, :22] and a temporal term on vertices (run/motion_denoising.py:255-261)."""
    from oracle import fk_torch
    B = 12
    pose = _poses(B, 33, scale=0.3)
    p_d = torch.tensor(pose, device=DEV, requires_grad=True)
    out = bm(pose_body=p_d)
    tgt = torch.tensor(np.random.RandomState(5).standard_normal((B, 22, 3)).astype(np.float32), device=DEV)
    loss = ((out.v[:-1] - out.v[1:]) ** 2).mean() + ((out.Jtr[:, :22] - tgt) ** 2).mean()
    loss.backward()
    p_r = torch.tensor(pose, dtype=torch.float64, requires_grad=True)
    v, j = fk_torch.smplx_forward(asset, p_r)
    lref = ((v[:-1] - v[1:]) ** 2).mean() + ((j[:, :22] - tgt.cpu().double()) ** 2).mean()
    lref.backward()
    err = float(np.linalg.norm(t2n(p_d.grad) - p_r.grad.numpy()) / np.linalg.norm(p_r.grad.numpy()))
    assert err < 2e-4, err


_SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
_SMPLH_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 22, 23, 20, 25, 26, 20, 28, 29, 20, 31, 32,
                  20, 34, 35, 21, 37, 38, 21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50]


@pytest.mark.parametrize("parents,segs", [(_SMPL_PARENTS, (1, 23)), (_SMPLH_PARENTS, (1, 21, 15, 15))])
def test_fk_smpl_and_smplh_trees_through_the_c_abi(parents, segs):
    """The Python wrapper only builds SMPL-X (what the reference's scripts use); the library also unrolls the SMPL (24 joints)
    and SMPL-H (52) chains.  Posed joints and skinning transforms straight through dposer_fk_joints vs the chain oracle."""
    import ctypes as C
    from dposer_amd import _C
    lib = _C.lib()
    J, B = len(parents), 130
    desc = _C.BodyDesc(J, 100, 10, 0, 0)
    h = C.c_void_p()
    par = (C.c_int32 * J)(*parents)
    _C.check(lib.dposer_body_create(C.byref(desc), par, C.byref(h)), "dposer_body_create")
    try:
        rs = np.random.RandomState(J)
        full = (rs.standard_normal((B, J, 3)) * 0.5).astype(np.float32)
        jrest = rs.standard_normal((J, 3)).astype(np.float32)
        transl = rs.standard_normal((B, 3)).astype(np.float32)
        parts, first = [], 0
        for n in segs:
            parts.append(torch.tensor(full[:, first:first + n].reshape(B, n * 3).copy(), device=DEV))
            first += n
        segp = (C.c_void_p * len(segs))(*[t.data_ptr() for t in parts])
        segj = (C.c_int32 * len(segs))(*segs)
        jr, tr = torch.tensor(jrest, device=DEV), torch.tensor(transl, device=DEV)
        joints = torch.empty(B, J, 3, device=DEV)
        rel = torch.empty(B, J, 12, device=DEV)
        _C.check(lib.dposer_fk_joints(h, segp, segj, len(segs), _C.ptr(jr), 0, _C.ptr(tr), _C.ptr(joints), _C.ptr(rel), J, B, _C.stream_ptr()),
                 "dposer_fk_joints")
        R = fk_ref.batch_rodrigues(full.reshape(-1, 3).astype(np.float64)).reshape(B, J, 3, 3)
        posed, A = fk_ref.batch_rigid_transform(R, np.broadcast_to(jrest.astype(np.float64), (B, J, 3)).copy(), np.array(parents))
        assert np.abs(t2n(joints) - (posed + transl[:, None].astype(np.float64))).max() < 1e-5
        assert np.abs(t2n(rel).reshape(B, J, 3, 4) - A[:, :, :3, :]).max() < 1e-5
        # body-only query (22 joints; the lean specialisation) on the same trees
        j22 = torch.empty(B, 22, 3, device=DEV)
        _C.check(lib.dposer_fk_joints(h, segp, segj, len(segs), _C.ptr(jr), 0, None, _C.ptr(j22), None, 22, B, _C.stream_ptr()), "dposer_fk_joints")
        assert np.abs(t2n(j22) - posed[:, :22]).max() < 1e-5
    finally:
        lib.dposer_body_destroy(h)
    # an unknown kinematic tree is refused, not mis-evaluated
    bad = list(parents)
    bad[5] = 1
    h2 = C.c_void_p()
    assert lib.dposer_body_create(C.byref(desc), (C.c_int32 * J)(*bad), C.byref(h2)) < 0
    assert b"kinematic tree" in lib.dposer_last_error()


@pytest.mark.parametrize("V,nb,ne,nnz", [(6890, 16, 10, 8), (1000, 10, 0, 2), (10475, 300, 100, 4)])
def test_lbs_other_asset_shapes_vs_oracle(V, nb, ne, nnz):
    """Assets other than the 10475-vertex / 10+10 shape-coefficient / 4-weights-per-vertex default: vertex counts that are not a
    multiple of the GEMM padding, wider shape spaces, denser and sparser skinning rows.  Forward and pose / betas gradients."""
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    from oracle import fk_torch
    asset = make_synthetic_smplx_asset(seed=V, num_vertices=V, num_betas=nb, num_expressions=ne, nnz_per_vertex=nnz)
    bm = BodyModel(asset, num_betas=nb, num_expressions=ne, model_type="smplx").to(DEV)
    B = 9
    rs = np.random.RandomState(V)
    pose = _poses(B, seed=V)
    betas = (rs.standard_normal((B, nb)) * 0.3).astype(np.float32)
    p_d = torch.tensor(pose, device=DEV, requires_grad=True)
    b_d = torch.tensor(betas, device=DEV, requires_grad=True)
    out = bm(pose_body=p_d, betas=b_d)
    verts, joints, _, _ = fk_ref.smplx_forward(asset, pose.astype(np.float64), betas=betas.astype(np.float64), dtype=np.float64)
    assert out.v.shape == (B, V, 3)
    assert np.abs(t2n(out.v) - verts).max() < 2e-5 and np.abs(t2n(out.Jtr) - joints).max() < 2e-5
    wv = rs.standard_normal((B, V, 3)).astype(np.float32) / 50.0
    wj = rs.standard_normal(tuple(out.Jtr.shape)).astype(np.float32)
    ((out.v * torch.tensor(wv, device=DEV)).sum() + (out.Jtr * torch.tensor(wj, device=DEV)).sum()).backward()
    p_r = torch.tensor(pose, dtype=torch.float64, requires_grad=True)
    b_r = torch.tensor(betas, dtype=torch.float64, requires_grad=True)
    v, j = fk_torch.smplx_forward(asset, p_r, betas=b_r)
    ((v * torch.tensor(wv, dtype=torch.float64)).sum() + (j * torch.tensor(wj, dtype=torch.float64)).sum()).backward()
    for name, got, want in (("pose", p_d.grad, p_r.grad), ("betas", b_d.grad, b_r.grad)):
        err = float(np.linalg.norm(t2n(got) - want.numpy()) / np.linalg.norm(want.numpy()))
        assert err < 2e-4, (name, err)


def test_body_model_rejects_out_of_range_vertex_tables():
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    asset = dict(make_synthetic_smplx_asset(seed=1, num_vertices=500))
    bad = dict(asset)
    bad["extra_joint_vertex_ids"] = np.array(asset["extra_joint_vertex_ids"]).copy()
    bad["extra_joint_vertex_ids"][3] = 500
    with pytest.raises(ValueError, match="extra_joint_vertex_ids"):
        BodyModel(bad, model_type="smplx")


def test_body_model_backward_all_pose_segments(bm, asset):
    """Gradients w.r.t. every pose segment BodyModel.forward accepts (hands, jaw, eyes besides body / root) and the expression
    coefficients -- the smplify-style fitting configuration."""
    from oracle import fk_torch
    B = 4
    rs = np.random.RandomState(77)
    mk = lambda n, s=0.3: (rs.standard_normal((B, n)) * s).astype(np.float32)
    pose, root, hand, jaw, eye, expr = mk(63), mk(3), mk(90, 0.2), mk(3, 0.2), mk(6, 0.1), mk(10, 0.5)
    dev = lambda a: torch.tensor(a, device=DEV, requires_grad=True)
    d = [dev(a) for a in (pose, root, hand, jaw, eye, expr)]
    out = bm(pose_body=d[0], root_orient=d[1], pose_hand=d[2], pose_jaw=d[3], pose_eye=d[4], expression=d[5])
    wv = rs.standard_normal((B, 10475, 3)).astype(np.float32) / 100.0
    wj = rs.standard_normal((B, 127, 3)).astype(np.float32)
    ((out.v * torch.tensor(wv, device=DEV)).sum() + (out.Jtr * torch.tensor(wj, device=DEV)).sum()).backward()
    ref = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    r = [ref(a) for a in (pose, root, hand, jaw, eye, expr)]
    v, j = fk_torch.smplx_forward(asset, r[0], global_orient=r[1], left_hand_pose=r[2][:, :45], right_hand_pose=r[2][:, 45:], jaw_pose=r[3],
                                  leye_pose=r[4][:, :3], reye_pose=r[4][:, 3:], expression=r[5])
    ((v * torch.tensor(wv, dtype=torch.float64)).sum() + (j * torch.tensor(wj, dtype=torch.float64)).sum()).backward()
    for name, got, want in zip(("pose", "root", "hand", "jaw", "eye", "expression"), d, r):
        err = float(np.linalg.norm(t2n(got.grad) - want.grad.numpy()) / np.linalg.norm(want.grad.numpy()))
        assert err < 2e-4, (name, err)


@pytest.mark.parametrize("B", [7, 300])
def test_pose_blend_bf16x3_against_exact_fp32_and_the_oracle(bm, asset, B, monkeypatch):
    """The pose-blend GEMMs (offsets = pose_feature @ posedirs and its transpose in the backward pass) run by default as three
    bf16 products of two-term splits (hi*hi + hi*lo + lo*hi, fp32 accumulation); DPOSER_LBS_BLEND=fp32 selects the exact-fp32 MFMA
    chain.  Both must sit inside the 1e-5 vertex bar against the fp64 oracle; measured max abs vertex
    error [m] at 7 / 300 poses of 0.5 rad: bf16x3 6.2e-7 / 1.04e-6, fp32 4.5e-7 / 8.8e-7 (fp32 rounding of the skinning sum dominates both)."""
    from oracle import fk_torch
    rs = np.random.RandomState(B)
    pose = (rs.standard_normal((B, 63)) * 0.5).astype(np.float32)
    hand = (rs.standard_normal((B, 90)) * 0.3).astype(np.float32)
    wv = torch.tensor(rs.standard_normal((B, 10475, 3)).astype(np.float32) / 100.0, device=DEV)
    v_ref, j_ref, _, _ = fk_ref.smplx_forward(asset, pose.astype(np.float64), left_hand_pose=hand[:, :45].astype(np.float64),
                                              right_hand_pose=hand[:, 45:].astype(np.float64), dtype=np.float64)

    def run():
        p = torch.tensor(pose, device=DEV, requires_grad=True)
        hd = torch.tensor(hand, device=DEV, requires_grad=True)
        out = bm(pose_body=p, pose_hand=hd)
        (out.v * wv).sum().backward()
        return t2n(out.v), t2n(p.grad), t2n(hd.grad)

    v3, gp3, gh3 = run()
    monkeypatch.setenv("DPOSER_LBS_BLEND", "fp32")
    _reload_tuning()
    v1, gp1, gh1 = run()
    monkeypatch.delenv("DPOSER_LBS_BLEND")
    _reload_tuning()
    e3, e1 = np.abs(v3 - v_ref).max(), np.abs(v1 - v_ref).max()
    _log_measured("pose blend bf16x3 max abs vertex error", e3)
    _log_measured("pose blend fp32 max abs vertex error", e1)
    assert e1 < 1e-6 and e3 < 2e-6, (e1, e3)
    assert not np.array_equal(v3, v1)                                            # (the two modes really are different arithmetic)
    for a, b in ((gp3, gp1), (gh3, gh1)):
        assert np.linalg.norm(a - b) / np.linalg.norm(b) < 2e-5


def test_small_batch_fk_kernels_return_the_bits_of_the_large_batch_ones(bm, asset, monkeypatch):
    """Below DPOSER_FK_SMALL_MAX poses (default 8192) forward kinematics and its backward run one wave per pose / one lane per
    joint (k_fk_small, k_fk_bwd_small: the chain walked by tree depth through LDS) instead of one lane per pose.  Same
    per-joint expressions, children summed in the unrolled chain's order => identical bits: joints-only query, full LBS
    (vertices, 127 joints) and every gradient of the all-segments backward."""
    B = 37
    rs = np.random.RandomState(91)
    mk = lambda n, s=0.3: (rs.standard_normal((B, n)) * s).astype(np.float32)
    arrs = [mk(63), mk(3), mk(90, 0.2), mk(3, 0.2), mk(6, 0.1), mk(10, 0.5), mk(10, 0.5), mk(3, 1.0)]
    wv = torch.tensor(rs.standard_normal((B, 10475, 3)).astype(np.float32) / 100.0, device=DEV)
    wj = torch.tensor(rs.standard_normal((B, 127, 3)).astype(np.float32), device=DEV)

    def run():
        d = [torch.tensor(a, device=DEV, requires_grad=True) for a in arrs]
        out = bm(pose_body=d[0], root_orient=d[1], pose_hand=d[2], pose_jaw=d[3], pose_eye=d[4], expression=d[5], betas=d[6], trans=d[7])
        ((out.v * wv).sum() + (out.Jtr * wj).sum()).backward()
        with torch.no_grad():
            jo = bm.fk_joints(d[0].detach(), root_orient=d[1].detach(), trans=d[7].detach())
        return [out.v.detach(), out.Jtr.detach(), jo] + [t.grad for t in d]

    small = run()
    monkeypatch.setenv("DPOSER_FK_SMALL_MAX", "0")
    _reload_tuning()
    large = run()
    monkeypatch.delenv("DPOSER_FK_SMALL_MAX")
    _reload_tuning()
    names = ["v", "Jtr", "fk_joints", "d pose", "d root", "d hand", "d jaw", "d eye", "d expression", "d betas", "d trans"]
    for n, a, b in zip(names, small, large):
        assert torch.equal(a, b), n


def test_joint_gradient_kernels_agree(bm, monkeypatch):
    """d loss / d (skinning transforms): the (pose, joint)-parallel gather kernel (batches below DPOSER_LBS_JOINT_STREAM_MIN = 320
    poses) and the streaming per-pose kernel over chunk-major lists (above) sum the same terms in different orders."""
    B = 9
    pose = (np.random.RandomState(2).standard_normal((B, 63)) * 0.4).astype(np.float32)
    wv = torch.tensor(np.random.RandomState(3).standard_normal((B, 10475, 3)).astype(np.float32) / 50.0, device=DEV)

    def run():
        p = torch.tensor(pose, device=DEV, requires_grad=True)
        out = bm(pose_body=p)
        ((out.v * wv).sum() + (out.Jtr ** 2).sum()).backward()
        return t2n(p.grad)

    g_gather = run()
    monkeypatch.setenv("DPOSER_LBS_JOINT_STREAM_MIN", "1")
    _reload_tuning()
    g_stream = run()
    monkeypatch.delenv("DPOSER_LBS_JOINT_STREAM_MIN")
    _reload_tuning()
    assert np.linalg.norm(g_gather - g_stream) / np.linalg.norm(g_gather) < 1e-5
    assert not np.array_equal(g_gather, g_stream)


def test_blend_gradient_terms_side_by_side_equal_one_after_the_other(bm, monkeypatch):
    """From 2048 poses the three product terms of the bf16 x 3 blend-gradient GEMM run side by side on three streams of the body handle,
    with a split count chosen for the three launches together; DPOSER_LBS_BWD_TERMS_PARALLEL=0 runs them one after the other on the
    caller's stream.  Same products, another number of split-K slabs: the pose gradient agrees to fp32 summation noise, the call leaves
    the caller's stream joined (a second backward right behind the first reads finished buffers), and repeated calls are bit-identical."""
    B = 2048
    gen = torch.Generator(device=DEV).manual_seed(5)
    pose = (torch.randn(B, 63, device=DEV, generator=gen) * 0.3)
    gv = torch.randn(B, 10475, 3, device=DEV, generator=gen) * 0.01
    gj = torch.randn(B, 127, 3, device=DEV, generator=gen)

    def run():
        _reload_tuning()
        p = pose.clone().requires_grad_(True)
        out = bm(pose_body=p)
        torch.autograd.backward([out.v, out.Jtr], [gv, gj])
        return p.grad.clone()

    par = run()
    par2 = run()
    monkeypatch.setenv("DPOSER_LBS_BWD_TERMS_PARALLEL", "0")
    ser = run()
    monkeypatch.delenv("DPOSER_LBS_BWD_TERMS_PARALLEL")
    _reload_tuning()
    assert torch.equal(par, par2)
    err = float((par - ser).norm() / ser.norm())
    _log_measured("blend-gradient terms on three streams vs one", err)
    assert err < 1e-6 and bool(torch.isfinite(par).all())


@pytest.mark.parametrize("segments", ["body", "all"])
def test_row_concatenated_blend_gradient_launch_keeps_every_bit(bm, tuning_env, segments):
    """Round 6: the two blend-gradient product terms that read the high plane of d_offsets are ONE launch against the row-concatenated
    [posedirs high ; posedirs low] -- the packed 256-row prefix panel when only the body is posed (pe = 256 of 512 rows), the natural
    layout when every pose segment wants a gradient -- and the slabs are added in the three-launch form's order: the pose gradients must
    carry the bits of DPOSER_LBS_BWD_ROWCAT=0, with the terms side by side and one after the other."""
    B = 2304
    gen = torch.Generator(device=DEV).manual_seed(11)
    pose = (torch.randn(B, 63, device=DEV, generator=gen) * 0.3)
    hand = (torch.randn(B, 90, device=DEV, generator=gen) * 0.2)
    jaw = (torch.randn(B, 3, device=DEV, generator=gen) * 0.2)
    gv = torch.randn(B, 10475, 3, device=DEV, generator=gen) * 0.01
    gj = torch.randn(B, 127, 3, device=DEV, generator=gen)

    def run():
        p = pose.clone().requires_grad_(True)
        kw = dict(pose_body=p)
        leaves = [p]
        if segments == "all":
            h, jw = hand.clone().requires_grad_(True), jaw.clone().requires_grad_(True)
            kw.update(pose_hand=h, pose_jaw=jw)
            leaves += [h, jw]
        out = bm(**kw)
        torch.autograd.backward([out.v, out.Jtr], [gv, gj])
        return [t.grad.clone() for t in leaves]

    got = {}
    for par in ("1", "0"):
        for rowcat in ("1", "0"):
            tuning_env(DPOSER_LBS_BWD_TERMS_PARALLEL=par, DPOSER_LBS_BWD_ROWCAT=rowcat)
            got[(par, rowcat)] = run()
    for par in ("1", "0"):
        for a, b in zip(got[(par, "1")], got[(par, "0")]):
            assert torch.equal(a, b), (segments, par, float((a - b).abs().max()))
            assert bool(torch.isfinite(a).all()) and float(a.abs().max()) > 0


def test_fused_skinning_backward_agrees_with_the_two_kernel_path(bm, monkeypatch):
    """The one-pass skinning-backward kernels -- k_skin_bwd_mfma (default: the joint reduction as dense 16x16x32 MFMAs on bf16 hi / lo
    planes, a workgroup per four poses) and k_skin_bwd_fused (one streaming pass per pose: d_verts read once, v_posed never in HBM, joint
    lists in balanced segments) -- against k_skin_bwd + k_skin_bwd_joints: same terms, other summation orders.  B = 70: the first 64 poses go through the permuted block -> pose map, the rest through the identity tail; betas require a
    gradient, so d v_posed is written too."""
    B = 70
    rs = np.random.RandomState(12)
    pose = (rs.standard_normal((B, 63)) * 0.4).astype(np.float32)
    betas = (rs.standard_normal((B, 10)) * 0.5).astype(np.float32)
    wv = torch.tensor(rs.standard_normal((B, 10475, 3)).astype(np.float32) / 50.0, device=DEV)
    monkeypatch.setenv("DPOSER_LBS_JOINT_STREAM_MIN", "1")

    def run():
        _reload_tuning()
        p = torch.tensor(pose, device=DEV, requires_grad=True)
        b = torch.tensor(betas, device=DEV, requires_grad=True)
        out = bm(pose_body=p, betas=b)
        ((out.v * wv).sum() + (out.Jtr ** 2).sum()).backward()
        return t2n(p.grad), t2n(b.grad)

    mfma = run()                                       # default: k_skin_bwd_mfma (joint reduction on the matrix pipe, four poses per workgroup)
    monkeypatch.setenv("DPOSER_SKIN_BWD_MFMA", "2")
    mfma2 = run()                                      # ... two poses per workgroup
    monkeypatch.setenv("DPOSER_SKIN_BWD_MFMA", "0")
    fused = run()                                      # k_skin_bwd_fused (joint lists walked through LDS)
    monkeypatch.setenv("DPOSER_SKIN_BWD_FUSED", "0")
    two = run()
    monkeypatch.delenv("DPOSER_SKIN_BWD_FUSED")
    monkeypatch.delenv("DPOSER_SKIN_BWD_MFMA")
    monkeypatch.delenv("DPOSER_LBS_JOINT_STREAM_MIN")
    _reload_tuning()
    for tag, got in (("mfma", mfma), ("mfma, 2 poses", mfma2), ("fused", fused)):
        for name, a, b in (("d pose", got[0], two[0]), ("d betas", got[1], two[1])):
            err = np.linalg.norm(a - b) / np.linalg.norm(b)
            _log_measured(f"{tag} skinning backward vs two kernels, {name}", err)
            assert err < 1e-5, (tag, name, err)        # measured: fused 2e-7 / 1e-7, mfma 3e-6 (bf16 hi / lo split products)
            assert np.isfinite(a).all()
    assert not np.array_equal(fused[0], two[0]) and not np.array_equal(mfma[0], fused[0])        # (they really are other kernels)
    assert np.array_equal(mfma[0], mfma2[0]) and np.array_equal(mfma[1], mfma2[1])               # the workgroup's pose count changes no sum


@pytest.mark.parametrize("stream_min", ["1", None])       # the fused streaming kernel forced onto the small batch / the small-batch kernels
def test_landmark_and_extra_joint_gradients_are_folded_inside_the_library(bm, asset, monkeypatch, stream_min):
    """d_joints[:, J:] (vertex-selected extra joints, barycentric landmarks) reaches the pose through dposer_lbs_backward_fold: a loss on
    those rows ONLY against the fp64 oracle, and the incoming vertex gradient -- autograd's tensor -- is bit-unchanged while the call
    runs (copied out on a second stream, unsynchronised with the first) and after it.  (Rounds 3-4 added the fold to that tensor in
    place with torch ops and restored it afterwards.)"""
    from oracle import fk_torch
    if stream_min:
        monkeypatch.setenv("DPOSER_LBS_JOINT_STREAM_MIN", stream_min)
    _reload_tuning()
    try:
        B, J = 12, 55
        rs = np.random.RandomState(31)
        pose = (rs.standard_normal((B, 63)) * 0.4).astype(np.float32)
        wj = rs.standard_normal((B, 127, 3)).astype(np.float32)
        wj[:, :J] = 0.0
        p = torch.tensor(pose, device=DEV, requires_grad=True)
        out = bm(pose_body=p)
        gv = torch.tensor(rs.standard_normal((B, 10475, 3)).astype(np.float32) / 50.0, device=DEV)
        gj = torch.tensor(wj, device=DEV)
        gv0 = gv.clone()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        torch.autograd.backward([out.v, out.Jtr], [gv, gj])          # enqueues the backward kernels on the current stream ...
        with torch.cuda.stream(side):
            during = gv.clone()                                      # ... while this copy races them on another one
        torch.cuda.synchronize()
        assert torch.equal(during, gv0) and torch.equal(gv, gv0)
        p_r = torch.tensor(pose, dtype=torch.float64, requires_grad=True)
        v, j = fk_torch.smplx_forward(asset, p_r)
        ((v * gv0.cpu().double()).sum() + (j * torch.tensor(wj, dtype=torch.float64)).sum()).backward()
        err = float(np.linalg.norm(t2n(p.grad) - p_r.grad.numpy()) / np.linalg.norm(p_r.grad.numpy()))
        assert err < 1e-5, err
        # and the landmark rows alone (no vertex gradient at all): everything the pose receives came through the fold
        p2 = torch.tensor(pose, device=DEV, requires_grad=True)
        out2 = bm(pose_body=p2)
        torch.autograd.backward([out2.v, out2.Jtr], [torch.zeros_like(gv), gj])
        p_r2 = torch.tensor(pose, dtype=torch.float64, requires_grad=True)
        _, j2 = fk_torch.smplx_forward(asset, p_r2)
        (j2 * torch.tensor(wj, dtype=torch.float64)).sum().backward()
        err2 = float(np.linalg.norm(t2n(p2.grad) - p_r2.grad.numpy()) / np.linalg.norm(p_r2.grad.numpy()))
        assert err2 < 1e-5 and float(np.linalg.norm(p_r2.grad.numpy())) > 0, err2
    finally:
        if stream_min:
            monkeypatch.delenv("DPOSER_LBS_JOINT_STREAM_MIN")
        _reload_tuning()


def test_dma_staged_fk_joints_returns_the_bits_of_the_general_kernel(bm, monkeypatch):
    """k_fk_joints_dma (full blocks of 64 poses of the 22-joint query: pose rows enter LDS by global_load_lds, joints leave as the
    [64][66] image) against k_fk_joints, with and without root orientation / translation; 209 poses = 3 full blocks + a tail of 17
    through the general kernel."""
    B = 209
    rs = np.random.RandomState(44)
    pose = torch.tensor((rs.standard_normal((B, 63)) * 0.5).astype(np.float32), device=DEV)
    root = torch.tensor((rs.standard_normal((B, 3)) * 0.8).astype(np.float32), device=DEV)
    tr = torch.tensor(rs.standard_normal((B, 3)).astype(np.float32), device=DEV)
    monkeypatch.setenv("DPOSER_FK_SMALL_MAX", "0")          # (batches up to 8192 poses would take the one-wave-per-pose kernel)

    def run():
        _reload_tuning()
        return [bm.fk_joints(pose), bm.fk_joints(pose, root_orient=root, trans=tr)]

    dma = run()
    monkeypatch.setenv("DPOSER_FK_DMA", "0")
    gen = run()
    monkeypatch.delenv("DPOSER_FK_DMA")
    monkeypatch.delenv("DPOSER_FK_SMALL_MAX")
    _reload_tuning()
    for a, b in zip(dma, gen):
        assert a.shape == (B, 22, 3) and torch.equal(a, b)
    assert not torch.equal(dma[0], dma[1])


@pytest.mark.parametrize("model_type", ["smpl", "smplh"])
def test_small_batch_fk_kernels_other_trees(model_type, monkeypatch):
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_asset
    bm2 = BodyModel(make_synthetic_asset(model_type, seed=3), model_type=model_type).to(DEV)
    B = 9
    nb = bm2.bm.NUM_BODY_JOINTS * 3
    pose = (np.random.RandomState(4).standard_normal((B, nb)) * 0.3).astype(np.float32)

    def run():
        p = torch.tensor(pose, device=DEV, requires_grad=True)
        out = bm2(pose_body=p)
        (out.v.sum() + (out.Jtr ** 2).sum()).backward()
        return out.v.detach(), out.Jtr.detach(), p.grad

    small = run()
    monkeypatch.setenv("DPOSER_FK_SMALL_MAX", "0")
    _reload_tuning()
    large = run()
    monkeypatch.delenv("DPOSER_FK_SMALL_MAX")
    _reload_tuning()
    for a, b in zip(small, large):
        assert torch.equal(a, b)


# ------------------------------------------------------------------------------------------------
# Pins that need neither smplx nor the restatement in oracle/fk_ref.py: closed-form consequences of the LBS definition
# (rotations from scipy.spatial.transform, everything else from the asset arrays).
# ------------------------------------------------------------------------------------------------
def _rot(aa):
    from scipy.spatial.transform import Rotation
    return Rotation.from_rotvec(np.asarray(aa, dtype=np.float64)).as_matrix()


def _descendants(parents, k):
    out = []
    for j in range(len(parents)):
        a = j
        while a != -1 and a != k:
            a = int(parents[a])
        if a == k and j != k:
            out.append(j)
    return out


def test_zero_pose_returns_the_template_and_its_regressed_joints(bm, asset):
    """theta = 0, beta = 0: every transform is the identity, the pose feature vanishes, so v == v_template and the 55 skeleton
    joints == J_regressor @ v_template (lbs.py definition, no arithmetic to restate)."""
    z = torch.zeros(3, 63, device=DEV)
    o = bm(pose_body=z)
    vt = asset["v_template"].astype(np.float64)
    assert np.abs(t2n(o.v) - vt[None]).max() < 2e-6
    assert np.abs(t2n(o.Jtr)[:, :55] - (asset["J_regressor"].astype(np.float64) @ vt)[None]).max() < 2e-6


@pytest.mark.parametrize("joint", [1, 4, 9, 16, 20])       # left hip, left knee, spine3, left shoulder, left wrist (SMPL numbering)
def test_single_joint_rotation_closed_form(bm, asset, joint):
    """Rotating ONE joint k by R moves its descendants rigidly about J_k and nothing else:
    J_d' = J_k + R (J_d - J_k) for d below k, J_j' = J_j otherwise."""
    rs = np.random.RandomState(joint)
    aa = rs.standard_normal(3) * 0.7
    pose = np.zeros((1, 63), np.float32)
    pose[0, 3 * (joint - 1):3 * joint] = aa
    j = t2n(bm.fk_joints(torch.tensor(pose, device=DEV), n_joints=55))[0].astype(np.float64)
    J0 = asset["J_regressor"].astype(np.float64) @ asset["v_template"].astype(np.float64)
    want = J0.copy()
    R = _rot(pose[0, 3 * (joint - 1):3 * joint])
    for d in _descendants(asset["parents"], joint):
        want[d] = J0[joint] + R @ (J0[d] - J0[joint])
    assert np.abs(j - want).max() < 2e-6


def test_two_joint_chain_closed_form(bm, asset):
    """Hip then knee: J_ankle' = J_hip + R1 (J_knee - J_hip) + R1 R2 (J_ankle - J_knee)."""
    rs = np.random.RandomState(3)
    a1, a2 = rs.standard_normal(3) * 0.6, rs.standard_normal(3) * 0.9
    pose = np.zeros((1, 63), np.float32)
    pose[0, 0:3], pose[0, 9:12] = a1, a2                      # joints 1 (left hip) and 4 (left knee)
    j = t2n(bm.fk_joints(torch.tensor(pose, device=DEV), n_joints=22))[0].astype(np.float64)
    J0 = asset["J_regressor"].astype(np.float64) @ asset["v_template"].astype(np.float64)
    R1, R2 = _rot(pose[0, 0:3]), _rot(pose[0, 9:12])
    knee = J0[1] + R1 @ (J0[4] - J0[1])
    ankle = knee + R1 @ R2 @ (J0[7] - J0[4])
    foot = ankle + R1 @ R2 @ (J0[10] - J0[7])
    assert np.abs(j[4] - knee).max() < 2e-6 and np.abs(j[7] - ankle).max() < 2e-6 and np.abs(j[10] - foot).max() < 2e-6
    assert np.abs(j[2] - J0[2]).max() < 2e-6                 # the other leg does not move


def test_root_orientation_and_translation_act_rigidly(bm, asset):
    """global_orient = R, transl = t:  x' = J_root + R (x - J_root) + t for every vertex and joint (the pose-blend feature
    excludes the root joint, so the body does not deform)."""
    B = 5
    pose = _poses(B, seed=77)
    rs = np.random.RandomState(78)
    root = (rs.standard_normal((B, 3)) * 0.8).astype(np.float32)
    tr = rs.standard_normal((B, 3)).astype(np.float32)
    dev = lambda a: torch.tensor(a, device=DEV)
    o0 = bm(pose_body=dev(pose))
    o1 = bm(pose_body=dev(pose), root_orient=dev(root), trans=dev(tr))
    Jr = (asset["J_regressor"].astype(np.float64) @ asset["v_template"].astype(np.float64))[0]
    R = _rot(root)
    for a0, a1 in ((t2n(o0.v), t2n(o1.v)), (t2n(o0.Jtr), t2n(o1.Jtr))):
        want = np.einsum("bij,bnj->bni", R, a0.astype(np.float64) - Jr) + Jr + tr[:, None].astype(np.float64)
        assert np.abs(a1 - want).max() < 1e-5


def test_lbs_backward_against_finite_differences(bm):
    """Directional derivative of a random linear functional of (vertices, joints) w.r.t. the body pose, root orientation and
    translation: autograd through the HIP backward kernels vs central differences of the HIP forward."""
    B = 6
    rs = np.random.RandomState(9)
    dev = lambda a: torch.tensor(a.astype(np.float32), device=DEV)
    pose, root, tr = dev(rs.standard_normal((B, 63)) * 0.3), dev(rs.standard_normal((B, 3)) * 0.3), dev(rs.standard_normal((B, 3)))
    wv, wj = dev(rs.standard_normal((B, 10475, 3)) / 100), dev(rs.standard_normal((B, 127, 3)))

    def L(p, r, t):
        o = bm(pose_body=p, root_orient=r, trans=t)
        return ((o.v * wv).sum() + (o.Jtr * wj).sum()).double()

    p, r, t = pose.clone().requires_grad_(True), root.clone().requires_grad_(True), tr.clone().requires_grad_(True)
    gp, gr, gt = torch.autograd.grad(L(p, r, t), [p, r, t])
    eps = 1e-2
    for k in range(3):
        dp, dr, dt = dev(rs.standard_normal((B, 63))), dev(rs.standard_normal((B, 3))), dev(rs.standard_normal((B, 3)))
        with torch.no_grad():
            fd = float(L(pose + eps * dp, root + eps * dr, tr + eps * dt) - L(pose - eps * dp, root - eps * dr, tr - eps * dt)) / (2 * eps)
        an = float((gp * dp).sum() + (gr * dr).sum() + (gt * dt).sum())
        assert abs(fd - an) / max(abs(an), 1e-3) < 5e-3, (fd, an)


# ------------------------------------------------------------------------------------------------
# rest shape kernel (blend shapes + joint regression) and the SMPL / SMPL-H wrappers
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B", [1, 7, 8, 9, 300])
def test_shape_blend_and_joint_regression_vs_oracle(bm, asset, B):
    """v_shaped = v_template + shapedirs . [betas | expression], J = J_regressor @ v_shaped (lbs.py), per pose."""
    rs = np.random.RandomState(B)
    betas = rs.standard_normal((B, 10)).astype(np.float32) * 2
    expr = rs.standard_normal((B, 10)).astype(np.float32)
    vs, jr, batched = bm.bm.rest_shape(torch.tensor(betas, device=DEV), torch.tensor(expr, device=DEV))
    assert batched and vs.shape == (B, 10475, 3) and jr.shape == (B, 55, 3)
    shape = np.concatenate([betas, expr], axis=1).astype(np.float64)
    ref_v = asset["v_template"].astype(np.float64)[None] + np.einsum("bl,mkl->bmk", shape, asset["shapedirs"].astype(np.float64))
    ref_j = np.einsum("bik,ji->bjk", ref_v, asset["J_regressor"].astype(np.float64))
    assert np.abs(t2n(vs) - ref_v).max() < 1e-6 and np.abs(t2n(jr) - ref_j).max() < 1e-6
    # full forward with betas: vertices / joints vs the oracle
    pose = _poses(B, seed=B + 1)
    o = bm(pose_body=torch.tensor(pose, device=DEV), betas=torch.tensor(betas, device=DEV), expression=torch.tensor(expr, device=DEV))
    v_ref, j_ref, _, _ = fk_ref.smplx_forward(asset, pose.astype(np.float64), betas=betas.astype(np.float64), expression=expr.astype(np.float64),
                                              dtype=np.float64)
    assert np.abs(t2n(o.v) - v_ref).max() < 1e-5 and np.abs(t2n(o.Jtr) - j_ref).max() < 1e-5


def test_constant_betas_are_blended_once(bm):
    """The fitting loops pass the same betas tensor every step (motion_denoising.py:64,217): the rest shape is computed on the
    first call and reused while the tensor is unchanged; an in-place change or another tensor recomputes it."""
    betas = torch.zeros(6, 10, device=DEV)
    a = bm.bm.rest_shape(betas, None)
    b = bm.bm.rest_shape(betas, None)
    assert a[0] is b[0] and a[1] is b[1]
    betas.add_(1.0)
    c = bm.bm.rest_shape(betas, None)
    assert c[0] is not a[0] and float((c[0] - a[0]).abs().max()) > 0
    d = bm.bm.rest_shape(betas.clone(), None)
    assert d[0] is not c[0] and torch.equal(d[0], c[0])


def test_betas_gradient_through_shape_blend_and_lbs(bm, asset):
    """d loss / d betas, d expression (run/smplify.py:200-260 optimises betas) vs the differentiable fp64 restatement."""
    from oracle import fk_torch
    B = 5
    rs = np.random.RandomState(12)
    dev = lambda a: torch.tensor(a.astype(np.float32), device=DEV)
    pose, betas, expr = rs.standard_normal((B, 63)) * 0.3, rs.standard_normal((B, 10)), rs.standard_normal((B, 10)) * 0.5
    wv, wj = rs.standard_normal((B, 10475, 3)) / 100, rs.standard_normal((B, 127, 3))
    p, b, e = dev(pose).requires_grad_(True), dev(betas).requires_grad_(True), dev(expr).requires_grad_(True)
    o = bm(pose_body=p, betas=b, expression=e)
    ((o.v * dev(wv)).sum() + (o.Jtr * dev(wj)).sum()).backward()
    t64 = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    p2, b2, e2 = t64(pose.astype(np.float32)), t64(betas.astype(np.float32)), t64(expr.astype(np.float32))
    v, j = fk_torch.smplx_forward(asset, p2, betas=b2, expression=e2)
    ((v * torch.tensor(wv.astype(np.float32)).double()).sum() + (j * torch.tensor(wj.astype(np.float32)).double()).sum()).backward()
    for got, ref in ((p.grad, p2.grad), (b.grad, b2.grad), (e.grad, e2.grad)):
        assert np.abs(t2n(got) - ref.numpy()).max() / np.abs(ref.numpy()).max() < 1e-5


@pytest.mark.parametrize("model_type", ["smpl", "smplh"])
def test_smpl_and_smplh_wrappers_vs_oracle(model_type):
    """BodyModel(model_type='smpl' / 'smplh') (body_model.py:38-62): full-pose layouts global(1) body(23) and
    global(1) body(21) lhand(15) rhand(15); joints = LBS joints + 21 vertex-selected extras (45 / 73)."""
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_asset
    a = make_synthetic_asset(model_type, seed=3)
    m = BodyModel(a, num_betas=10, model_type=model_type).to(DEV)
    B = 37
    rs = np.random.RandomState(5)
    nb = 23 if model_type == "smpl" else 21
    body = (rs.standard_normal((B, nb * 3)) * 0.4).astype(np.float32)
    root = (rs.standard_normal((B, 3)) * 0.5).astype(np.float32)
    betas = rs.standard_normal((B, 10)).astype(np.float32)
    tr = rs.standard_normal((B, 3)).astype(np.float32)
    hand = (rs.standard_normal((B, 90)) * 0.3).astype(np.float32)
    dev = lambda x: torch.tensor(x, device=DEV)
    kw = dict(root_orient=dev(root), pose_body=dev(body), betas=dev(betas), trans=dev(tr))
    full = [root, body]
    if model_type == "smplh":
        kw["pose_hand"] = dev(hand)
        full.append(hand)
    o = m(**kw)
    v_ref, j_ref, _ = fk_ref.model_forward(a, np.concatenate(full, axis=1), shape=betas, transl=tr)
    assert m.num_joints == (23 if model_type == "smpl" else 51)
    assert o.Jtr.shape == (B, (24 if model_type == "smpl" else 52) + 21, 3) and o.v.shape == (B, 6890, 3)
    assert np.abs(t2n(o.v) - v_ref).max() < 1e-5 and np.abs(t2n(o.Jtr) - j_ref).max() < 1e-5
    assert o.full_pose.shape == (B, (24 if model_type == "smpl" else 52) * 3)
    assert hasattr(o, "pose_hand") == (model_type == "smplh") and not hasattr(o, "pose_jaw")
    # gradients flow through the same kernels
    p = dev(body).requires_grad_(True)
    m(pose_body=p, betas=dev(betas)).Jtr.sum().backward()
    assert torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0


def test_fk_large_batch_equals_the_same_poses_in_smaller_calls(bm):
    """300003 poses in one call (4688 single-wave blocks, ragged tail) vs the same poses in three calls: bit-identical joints,
    translation and root orientation included."""
    n = 300_003
    g = torch.Generator(device=DEV).manual_seed(3)
    pose = torch.randn(n, 63, device=DEV, generator=g) * 0.4
    tr = torch.randn(n, 3, device=DEV, generator=g)
    root = torch.randn(n, 3, device=DEV, generator=g) * 0.5
    for kw in (dict(), dict(trans=tr), dict(root_orient=root, trans=tr)):
        big = bm.fk_joints(pose, **kw)
        parts = []
        for lo in range(0, n, 100_000):
            sub = {k: v[lo:lo + 100_000].contiguous() for k, v in kw.items()}
            parts.append(bm.fk_joints(pose[lo:lo + 100_000].contiguous(), **sub))
        assert torch.equal(big, torch.cat(parts))


# ------------------------------------------------------------------------------------------------
# rotation conversions against scipy.spatial.transform.Rotation (independent implementation; torchgeometry is absent)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 255, 257, 50000])
def test_rotation_conversions_vs_scipy(n):
    from scipy.spatial.transform import Rotation
    from dposer_amd.utils.transforms import (axis_angle_to_mat3x3, axis_angle_to_rot6d, rot6d_to_axis_angle, rot6d_to_mat3x3,
                                             rotmat_to_axis_angle)
    rs = np.random.RandomState(n)
    aa = rs.standard_normal((n, 3)) * 1.2
    big = np.linalg.norm(aa, axis=1) > 3.0                    # keep the rotation angle below pi: the vector is then unique
    aa[big] *= 3.0 / np.linalg.norm(aa[big], axis=1, keepdims=True)
    aa[0] = 0.0                                               # identity
    aa = aa.astype(np.float32)
    Rref = Rotation.from_rotvec(aa.astype(np.float64)).as_matrix()
    R = t2n(axis_angle_to_mat3x3(torch.tensor(aa, device=DEV)))
    assert np.abs(R - Rref).max() < 2e-6
    r6 = axis_angle_to_rot6d(torch.tensor(aa, device=DEV))
    assert np.abs(t2n(r6).reshape(n, 3, 2) - Rref[:, :, :2]).max() < 2e-6          # first two columns, row-major 3x2
    assert np.abs(t2n(rot6d_to_mat3x3(r6)) - Rref).max() < 5e-6
    back = t2n(rotmat_to_axis_angle(torch.tensor(Rref.astype(np.float32), device=DEV)))
    ref_back = Rotation.from_matrix(Rref).as_rotvec()
    assert np.abs(back - ref_back).max() < 2e-5
    assert np.abs(t2n(rot6d_to_axis_angle(r6)) - aa).max() < 2e-5                  # round trip axis-angle -> 6D -> axis-angle
    # a scaled / skewed 6-D input is orthonormalised first (Gram-Schmidt), like transforms.py:227-235
    noisy = t2n(r6) * 1.7 + rs.standard_normal((n, 6)).astype(np.float32) * 0.05
    M = t2n(rot6d_to_mat3x3(torch.tensor(noisy, device=DEV))).astype(np.float64)
    assert np.abs(M @ M.transpose(0, 2, 1) - np.eye(3)).max() < 1e-5 and np.abs(np.linalg.det(M) - 1).max() < 1e-5
    got = t2n(rot6d_to_axis_angle(torch.tensor(noisy, device=DEV)))
    assert np.abs(got - Rotation.from_matrix(M).as_rotvec()).max() < 2e-5


def test_smplx_mean_poses_are_converted_on_the_device(asset, tmp_path, monkeypatch):
    """lib/body_model/smpl.py:59-62: mean_poses = rot6d_to_axis_angle(smpl_mean_params['pose']).  The conversion is a HIP kernel; a
    module built from a user-supplied smpl_mean_params.npz must end up with the converted values (round 2 swallowed the
    device-only error and registered zeros), checked here against scipy."""
    from scipy.spatial.transform import Rotation
    from dposer_amd.body_model import constants
    from dposer_amd.body_model.smpl import SMPLX
    rs = np.random.RandomState(3)
    rot = Rotation.from_rotvec(rs.standard_normal((24, 3)) * 0.6)
    R = rot.as_matrix()
    rot6d = R[:, :, :2].reshape(24, 6).astype(np.float32)          # 6D = the first two COLUMNS, row-major 3 x 2 (transforms.py:227-235)
    shape = rs.standard_normal(10).astype(np.float32)
    path = tmp_path / "smpl_mean_params.npz"
    np.savez(path, pose=rot6d.reshape(-1), shape=shape, cam=np.zeros(3, np.float32))
    monkeypatch.setattr(constants, "SMPL_MEAN_PATH", str(path))
    sm = SMPLX(asset)
    assert sm.mean_params_loaded and sm.mean_poses.shape == (72,)
    got = sm.mean_poses.cpu().numpy().reshape(24, 3)
    assert np.abs(got - rot.as_rotvec()).max() < 1e-5
    assert np.array_equal(sm.mean_shape.cpu().numpy(), shape)
    sm = sm.to(DEV)
    assert sm.mean_poses.device.type == "cuda" and np.abs(sm.mean_poses.cpu().numpy().reshape(24, 3) - rot.as_rotvec()).max() < 1e-5
    monkeypatch.setattr(constants, "SMPL_MEAN_PATH", str(tmp_path / "missing.npz"))
    sm0 = SMPLX(asset)
    assert not sm0.mean_params_loaded and float(sm0.mean_poses.abs().max()) == 0.0


def test_smplx_wrapper_accepts_rotation_matrices(asset):
    """pose2rot=False (smplx): rotation-matrix inputs must give the joints / vertices of the equivalent axis-angle call."""
    from scipy.spatial.transform import Rotation
    from dposer_amd.body_model.smpl import SMPLX
    sm = SMPLX(asset).to(DEV)
    rs = np.random.RandomState(4)
    B = 5
    body = (rs.standard_normal((B, 21, 3)) * 0.4).astype(np.float32)
    root = (rs.standard_normal((B, 1, 3)) * 0.4).astype(np.float32)
    Rb = Rotation.from_rotvec(body.reshape(-1, 3)).as_matrix().reshape(B, 21, 3, 3).astype(np.float32)
    Rr = Rotation.from_rotvec(root.reshape(-1, 3)).as_matrix().reshape(B, 1, 3, 3).astype(np.float32)
    with torch.no_grad():
        a = sm(body_pose=torch.tensor(body.reshape(B, 63), device=DEV), global_orient=torch.tensor(root.reshape(B, 3), device=DEV))
        b = sm(body_pose=torch.tensor(Rb, device=DEV), global_orient=torch.tensor(Rr, device=DEV), pose2rot=False)
    assert (a.joints - b.joints).abs().max() < 2e-6 and (a.vertices - b.vertices).abs().max() < 2e-6
    with pytest.raises(ValueError):
        sm(body_pose=torch.tensor(body.reshape(B, 63), device=DEV), pose2rot=False)


def test_empty_batch_returns_empty_outputs():
    """B = 0 (smplx returns empty vertices / joints; the kernels take B >= 1): full forward, joints-only forward and the differentiable path."""
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_smplx_asset
    bm = BodyModel(make_synthetic_smplx_asset(seed=0)).to(DEV)
    pose = torch.zeros(0, 63, device=DEV)
    out = bm(pose_body=pose)
    assert out.v.shape == (0, 10475, 3) and out.Jtr.shape[0] == 0 and out.Jtr.shape[2] == 3
    out = bm(pose_body=pose.clone().requires_grad_(True))
    assert out.v.shape == (0, 10475, 3)
    core = bm.bm
    j = core(body_pose=pose, joints_only=True, n_joints=22)
    assert j.joints.shape == (0, 22, 3)


@pytest.mark.parametrize("B,segments", [(300, "body"), (2304, "body"), (300, "body+jaw"), (64, "root only")])
def test_blend_gemms_over_the_posed_joints_only_keep_every_bit(bm, asset, B, segments, tuning_env):
    """The pose-blend GEMM reduces over the pose-feature columns of the joints that ARE posed (a NULL segment is the identity rotation:
    its columns of R - 1 are exact zeros) and the blend-gradient GEMMs produce the columns of the joints whose gradient is wanted --
    192 of 512 / 256 of 512 columns when only the body is posed, as in the fitting loops.  Adding exact zeros changes no bit:
    vertices and joints equal those of the full-width GEMMs (DPOSER_LBS_K_PREFIX=0) bit for bit, at the 128-wide (B = 300, 64) and
    the 256-wide tilings (B = 2304), the pose gradients to fp32 rounding (their split over the vertex dimension may differ); a posed
    jaw (joint 22) widens the prefix to 198 -> 224 columns."""
    rs = np.random.RandomState(B)
    pose = (rs.standard_normal((B, 63)) * 0.4).astype(np.float32)
    root = (rs.standard_normal((B, 3)) * 0.4).astype(np.float32)
    jaw = (rs.standard_normal((B, 3)) * 0.2).astype(np.float32)
    wv = torch.tensor(rs.standard_normal((B, 10475, 3)).astype(np.float32) / 100.0, device=DEV)
    wj = torch.tensor(rs.standard_normal((B, 127, 3)).astype(np.float32), device=DEV)

    def run():
        kw, leaves = {}, []
        r = torch.tensor(root, device=DEV, requires_grad=True)
        kw["root_orient"] = r
        leaves.append(r)
        if segments != "root only":
            p = torch.tensor(pose, device=DEV, requires_grad=True)
            kw["pose_body"] = p
            leaves.append(p)
        if segments == "body+jaw":
            j = torch.tensor(jaw, device=DEV, requires_grad=True)
            kw["pose_jaw"] = j
            leaves.append(j)
        out = bm(**kw)
        ((out.v * wv).sum() + (out.Jtr * wj).sum()).backward()
        return [t2n(out.v), t2n(out.Jtr)] + [t2n(x.grad) for x in leaves]

    got = run()
    tuning_env(DPOSER_LBS_K_PREFIX="0")
    full = run()
    for a, b in zip(got[:2], full[:2]):                  # vertices and joints: the same sums, minus exact zeros
        assert np.isfinite(a).all() and np.array_equal(a, b)
    for a, b in zip(got[2:], full[2:]):                  # gradients: the narrower slabs may be split over the vertices differently
        assert np.isfinite(a).all()                      # (another partition of the same fp32 sum: equal to rounding, often to the bit)
        assert np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64)) < 2e-6
    if segments == "body":                       # and the result is the oracle's
        v_ref, j_ref, _, _ = fk_ref.smplx_forward(asset, pose[:16].astype(np.float64), global_orient=root[:16].astype(np.float64), dtype=np.float64)
        assert np.abs(got[0][:16] - v_ref).max() < 1e-5


@pytest.mark.parametrize("B,batched_shape", [(7, False), (300, True), (1030, False), (4100, False)])
def test_skinning_over_runs_of_poses_returns_the_bits_of_the_per_pose_kernel(bm, B, batched_shape, tuning_env):
    """k_skin_run (a block keeps its vertices' skinning rows in registers and walks a run of poses, next pose's offsets and transforms
    in flight) against k_skin_x4 (one pose per block, DPOSER_SKIN_WAVE=2): the same per-vertex expressions in the same order, so
    vertices and joints are bit-identical -- with one shared and with per-pose rest shapes, with a translation, at batch sizes that give
    runs of 1 (B = 7, 300), 2 (1030) and 4 poses (4100, incl. a ragged last run)."""
    rs = np.random.RandomState(B)
    pose = torch.tensor((rs.standard_normal((B, 63)) * 0.4).astype(np.float32), device=DEV)
    trans = torch.tensor(rs.standard_normal((B, 3)).astype(np.float32), device=DEV)
    betas = torch.tensor(rs.standard_normal((B if batched_shape else 1, 10)).astype(np.float32), device=DEV)
    if not batched_shape:
        betas = betas.expand(B, 10).contiguous()

    def run():
        with torch.no_grad():
            out = bm(pose_body=pose, trans=trans, betas=betas)
        return t2n(out.v), t2n(out.Jtr)

    v3, j3 = run()
    tuning_env(DPOSER_SKIN_WAVE="2")
    v2, j2 = run()
    assert np.isfinite(v3).all() and np.array_equal(v3, v2) and np.array_equal(j3, j2)


def test_pack_calls_stay_inside_the_buffers_they_are_given(bm):
    """Every packing entry point writes a caller-owned buffer whose size the library itself reports.  Guard words behind that size must
    survive the call (round 6: a pack-job list that had grown by two jobs in the WRONG function wrote 32 MB behind the forward posedirs
    buffer -- the allocator's neighbours absorbed it until an unrelated test aborted)."""
    from dposer_amd import _C
    lib, core = _C.lib(), bm.bm
    h = core._handle()
    guard = 1 << 16
    for nbytes, call in ((lib.dposer_lbs_posedirs_packed_bytes(h), lib.dposer_lbs_pack_posedirs),
                         (lib.dposer_lbs_posedirs_bwd_packed_bytes(h), lib.dposer_lbs_pack_posedirs_bwd)):
        buf = torch.full((nbytes + guard,), 0xAB, dtype=torch.uint8, device=DEV)
        _C.check(call(h, _C.ptr(core.posedirs), _C.ptr(buf), _C.stream_ptr()), "pack")
        torch.cuda.synchronize()
        assert bool((buf[nbytes:] == 0xAB).all()), call
        assert bool((buf[:nbytes] != 0xAB).any())
    # the score network's and the TimeMLPs' packed weights
    from gpu_common import make_model
    for prec in ("bf16", "bf16x3", "fp32"):
        cfg, m, _ = make_model(3, precision=prec)
        eng = m._engine()
        nbytes = eng.lib.dposer_scorefc_packed_bytes(eng.h, 1)
        buf = torch.full((nbytes + guard,), 0xAB, dtype=torch.uint8, device=DEV)
        _C.check(eng.lib.dposer_scorefc_pack(eng.h, _C.ptr(m.flat_params()), _C.ptr(buf), 1, _C.stream_ptr()), "dposer_scorefc_pack")
        torch.cuda.synchronize()
        assert bool((buf[nbytes:] == 0xAB).all()), prec
