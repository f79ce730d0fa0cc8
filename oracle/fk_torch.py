"""Differentiable (torch, float64) restatement of smplx lbs / SMPLX.forward -- the arbiter for the
gradients of the HIP LBS backward.  TEST INFRASTRUCTURE (see oracle/__init__.py); same algorithm and
citations as oracle/fk_ref.py (smplx==0.1.28 lbs.py, body_models.py; PARITY UNPINNED)."""
import torch


def batch_rodrigues(rv):
    angle = torch.norm(rv + 1e-8, dim=1, keepdim=True)
    d = rv / angle
    c, s = torch.cos(angle)[:, :, None], torch.sin(angle)[:, :, None]
    rx, ry, rz = d[:, 0], d[:, 1], d[:, 2]
    z = torch.zeros_like(rx)
    K = torch.stack([z, -rz, ry, rz, z, -rx, -ry, rx, z], dim=1).view(-1, 3, 3)
    I = torch.eye(3, dtype=rv.dtype)[None]
    return I + s * K + (1 - c) * torch.bmm(K, K)


def smplx_forward(asset, body_pose, betas=None, global_orient=None, transl=None, expression=None, jaw_pose=None, leye_pose=None,
                  reye_pose=None, left_hand_pose=None, right_hand_pose=None, dtype=torch.float64):
    """``dtype``: float64 (the arbiter) or float32 -- the precision smplx itself runs in inside the reference's loops (used by
    tests/sensitivity/cfg5_sensitivity.py to size the reference's own rounding noise); inputs must already be of that dtype."""
    dt = dtype
    t = lambda a: torch.as_tensor(a, dtype=dt)
    B = body_pose.shape[0]
    z = lambda n: torch.zeros(B, n, dtype=dt)
    opt = lambda a, n: a if a is not None else z(n)
    # full_pose order of SMPLX.forward: global(1) body(21) jaw(1) leye(1) reye(1) lhand(15) rhand(15)
    full = torch.cat([opt(global_orient, 3), body_pose, opt(jaw_pose, 3), opt(leye_pose, 3), opt(reye_pose, 3),
                      opt(left_hand_pose, 45), opt(right_hand_pose, 45)], dim=1)
    nb, ne = asset["num_betas"], asset["num_expressions"]
    shape = torch.cat([opt(betas, nb), opt(expression, ne)], dim=1)
    v_shaped = t(asset["v_template"])[None] + torch.einsum("bl,mkl->bmk", shape, t(asset["shapedirs"]))
    J = torch.einsum("bik,ji->bjk", v_shaped, t(asset["J_regressor"]))
    nj = J.shape[1]
    R = batch_rodrigues(full.reshape(-1, 3)).view(B, nj, 3, 3)
    pf = (R[:, 1:] - torch.eye(3, dtype=dt)).reshape(B, -1)
    v_posed = v_shaped + (pf @ t(asset["posedirs"])).view(B, -1, 3)
    parents = [int(p) for p in asset["parents"]]
    rel = J.clone()
    rel[:, 1:] = J[:, 1:] - J[:, parents[1:]]
    T = torch.zeros(B, nj, 4, 4, dtype=dt)
    T[:, :, :3, :3] = R
    T[:, :, :3, 3] = rel
    T[:, :, 3, 3] = 1
    G = [T[:, 0]]
    for i in range(1, nj):
        G.append(G[parents[i]] @ T[:, i])
    G = torch.stack(G, dim=1)
    posed = G[:, :, :3, 3]
    jh = torch.cat([J, torch.zeros(B, nj, 1, dtype=dt)], dim=2)[..., None]
    A = G - torch.nn.functional.pad(G @ jh, [3, 0])
    W = t(asset["weights"])
    Tv = (W @ A.view(B, nj, 16)).view(B, -1, 4, 4)
    vh = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1, dtype=dt)], dim=2)
    verts = (Tv @ vh[..., None])[:, :, :3, 0]
    extra = verts[:, torch.as_tensor(asset["extra_joint_vertex_ids"]).long()]
    faces = torch.as_tensor(asset["faces"]).long()[torch.as_tensor(asset["lmk_faces_idx"]).long()]
    lmk = torch.einsum("blfi,lf->bli", verts[:, faces], t(asset["lmk_bary_coords"]))
    joints = torch.cat([posed, extra, lmk], dim=1)
    if transl is not None:
        joints = joints + transl[:, None]
        verts = verts + transl[:, None]
    return verts, joints
