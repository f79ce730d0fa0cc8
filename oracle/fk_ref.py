"""CPU restatement (numpy) of the body-model half of the hot path.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.

* ``rot6d_to_mat3x3`` follows the reference's own lib/utils/transforms.py:227-235 (pinned by a
  golden vector).
* Everything else restates the published algorithm of the un-vendored third-party dependency
  ``smplx==0.1.28`` (requirements.txt:4) -- ``smplx/lbs.py`` (``batch_rodrigues``, ``blend_shapes``,
  ``vertices2joints``, ``batch_rigid_transform``, ``lbs``, ``vertices2landmarks``),
  ``smplx/vertex_joint_selector.py`` and ``SMPLX.forward`` in ``smplx/body_models.py`` -- anchored
  on the reference's call sites lib/body_model/body_model.py:30-37,68-112 and
  lib/body_model/smpl.py:50-77.  **PARITY UNPINNED**: smplx is not installed, cannot be installed
  (no network) and the reference ships neither an SMPL-X asset nor a test for this path.
"""
from __future__ import annotations

import numpy as np

# SMPL-X kinematic tree (smplx model file 'kintree_table'[0]); first 22 entries agree with the
# reference's own get_smpl_skeleton (lib/body_model/utils.py:180-205).
SMPLX_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15,
     20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
     21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53], dtype=np.int64)


def rot6d_to_mat3x3(rot6d: np.ndarray) -> np.ndarray:
    """lib/utils/transforms.py:227-235.  The 6 numbers are a row-major 3x2 matrix = the first two
    columns of R.  F.normalize: x / max(||x||, 1e-12).  (The reference's torch.cross without
    ``dim`` picks the first size-3 axis, i.e. the wrong one when n == 3 -- Appendix C quirk; this
    restatement is the n != 3 behaviour.)"""
    m = rot6d.reshape(-1, 3, 2)
    a1, a2 = m[:, :, 0], m[:, :, 1]

    def nrm(v):
        return v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-12)

    b1 = nrm(a1)
    b2 = nrm(a2 - np.sum(b1 * a2, axis=1, keepdims=True) * b1)
    b3 = np.cross(b1, b2)
    return np.stack([b1, b2, b3], axis=-1)


def batch_rodrigues(rot_vecs: np.ndarray) -> np.ndarray:
    """smplx/lbs.py batch_rodrigues: angle = ||r + 1e-8||, k = r/angle,
    R = I + sin*K + (1-cos)*K@K."""
    dt = rot_vecs.dtype
    angle = np.linalg.norm(rot_vecs + dt.type(1e-8), axis=1, keepdims=True)
    d = rot_vecs / angle
    c = np.cos(angle)[:, :, None]
    s = np.sin(angle)[:, :, None]
    rx, ry, rz = d[:, 0], d[:, 1], d[:, 2]
    z = np.zeros_like(rx)
    K = np.stack([z, -rz, ry, rz, z, -rx, -ry, rx, z], axis=1).reshape(-1, 3, 3)
    I = np.eye(3, dtype=dt)[None]
    return I + s * K + (1 - c) * (K @ K)


def batch_rigid_transform(rot_mats, joints, parents):
    """smplx/lbs.py batch_rigid_transform -> (posed_joints [B,J,3], rel_transforms A [B,J,4,4])."""
    B, J = joints.shape[:2]
    dt = joints.dtype
    rel = joints.copy()
    rel[:, 1:] -= joints[:, parents[1:]]
    T = np.zeros((B, J, 4, 4), dtype=dt)
    T[:, :, :3, :3] = rot_mats
    T[:, :, :3, 3] = rel
    T[:, :, 3, 3] = 1
    G = [T[:, 0]]
    for i in range(1, J):
        G.append(G[parents[i]] @ T[:, i])
    G = np.stack(G, axis=1)
    posed = G[:, :, :3, 3].copy()
    jh = np.concatenate([joints, np.zeros((B, J, 1), dtype=dt)], axis=2)[..., None]
    corr = G @ jh                                     # [B,J,4,1]
    A = G.copy()
    A[:, :, :, 3:4] -= corr
    return posed, A


def lbs(betas, pose, asset, dtype=np.float32):
    """smplx/lbs.py lbs(pose2rot=True).  ``asset`` keys: v_template [V,3], shapedirs [V,3,L],
    posedirs [(J-1)*9, V*3], J_regressor [J,V], parents [J], weights [V,J]."""
    dt = dtype
    B = pose.shape[0]
    vt = asset["v_template"].astype(dt)
    sd = asset["shapedirs"].astype(dt)
    v_shaped = vt[None] + np.einsum("bl,mkl->bmk", betas.astype(dt), sd)
    J = np.einsum("bik,ji->bjk", v_shaped, asset["J_regressor"].astype(dt))
    nj = J.shape[1]
    R = batch_rodrigues(pose.astype(dt).reshape(-1, 3)).reshape(B, nj, 3, 3)
    pose_feature = (R[:, 1:] - np.eye(3, dtype=dt)).reshape(B, -1)
    v_posed = v_shaped + (pose_feature @ asset["posedirs"].astype(dt)).reshape(B, -1, 3)
    posed_joints, A = batch_rigid_transform(R, J, asset["parents"])
    W = asset["weights"].astype(dt)
    T = (W @ A.reshape(B, nj, 16)).reshape(B, -1, 4, 4)
    vh = np.concatenate([v_posed, np.ones((B, v_posed.shape[1], 1), dtype=dt)], axis=2)
    verts = (T @ vh[..., None])[:, :, :3, 0]
    return verts, posed_joints, dict(R=R, J_rest=J, A=A, v_posed=v_posed, v_shaped=v_shaped)


def smplx_forward(asset, body_pose, betas=None, global_orient=None, transl=None, expression=None,
                  jaw_pose=None, leye_pose=None, reye_pose=None, left_hand_pose=None,
                  right_hand_pose=None, dtype=np.float32):
    """SMPLX.forward (smplx/body_models.py), use_pca=False, flat_hand_mean=True (reference
    lib/body_model/body_model.py:30-37): full_pose order global(1) body(21) jaw(1) leye(1) reye(1)
    lhand(15) rhand(15); shape = [betas | expression]; joints = [55 LBS joints | 21
    vertex-selected extras | 51 static landmarks] = 127; + transl."""
    B = body_pose.shape[0]
    z = lambda n: np.zeros((B, n), dtype=dtype)
    parts = [global_orient if global_orient is not None else z(3), body_pose,
             jaw_pose if jaw_pose is not None else z(3),
             leye_pose if leye_pose is not None else z(3),
             reye_pose if reye_pose is not None else z(3),
             left_hand_pose if left_hand_pose is not None else z(45),
             right_hand_pose if right_hand_pose is not None else z(45)]
    full_pose = np.concatenate([p.astype(dtype) for p in parts], axis=1)
    nb = asset["num_betas"]
    ne = asset["num_expressions"]
    shape = np.concatenate([betas if betas is not None else z(nb),
                            expression if expression is not None else z(ne)], axis=1)
    verts, joints, aux = lbs(shape, full_pose, asset, dtype=dtype)
    extra = verts[:, asset["extra_joint_vertex_ids"]]
    faces = asset["faces"][asset["lmk_faces_idx"]]                    # [51,3]
    lmk_v = verts[:, faces]                                           # [B,51,3,3]
    landmarks = np.einsum("blfi,lf->bli", lmk_v, asset["lmk_bary_coords"].astype(dtype))
    joints = np.concatenate([joints, extra, landmarks], axis=1)
    if transl is not None:
        joints = joints + transl[:, None].astype(dtype)
        verts = verts + transl[:, None].astype(dtype)
    return verts, joints, full_pose, aux


def model_forward(asset, full_pose, shape=None, transl=None, dtype=np.float64):
    """smplx.{SMPL,SMPLH,SMPLX}.forward for an already assembled ``full_pose`` [B, J*3] (global orient first, then the body /
    hand / face segments in the order of smplx body_models.py): lbs() + vertex-selected extra joints (+ landmarks when the
    asset has them) + transl.  ``shape`` = [betas | expression] or None (zeros)."""
    B = full_pose.shape[0]
    L = asset["shapedirs"].shape[2]
    sh = np.zeros((B, L), dtype=dtype) if shape is None else shape.astype(dtype)
    verts, joints, aux = lbs(sh, full_pose.astype(dtype), asset, dtype=dtype)
    parts = [joints, verts[:, asset["extra_joint_vertex_ids"]]]
    if len(asset["lmk_faces_idx"]):
        faces = asset["faces"][asset["lmk_faces_idx"]]
        parts.append(np.einsum("blfi,lf->bli", verts[:, faces], asset["lmk_bary_coords"].astype(dtype)))
    joints = np.concatenate(parts, axis=1)
    if transl is not None:
        joints = joints + transl[:, None].astype(dtype)
        verts = verts + transl[:, None].astype(dtype)
    return verts, joints, aux
