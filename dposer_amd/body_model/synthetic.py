"""A synthetic SMPL-X-shaped asset (V = 10475, J = 55, real kinematic tree) for benchmarks and tests.

The licensed SMPL-X model file cannot be shipped (reference README.md:38-39 asks users to download
it); everything that needs a body model takes the same dictionary this function returns, which is
also what ``load_smplx_npz`` builds from a real ``SMPLX_*.npz``.
"""
import numpy as np

SMPLX_PARENTS = np.array(
    [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15,
     20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
     21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53], dtype=np.int32)

# smplx/vertex_ids.py ['smplx']: nose, reye, leye, rear, lear, LBigToe, LSmallToe, LHeel, RBigToe, RSmallToe, RHeel,
# l/r thumb, index, middle, ring, pinky tips -- the order VertexJointSelector concatenates them
SMPLX_EXTRA_VERTEX_IDS = np.array([9120, 9929, 9448, 616, 6, 5770, 5780, 8846, 8463, 8474, 8635,
                                   5361, 4933, 5058, 5169, 5286, 8079, 7669, 7794, 7905, 8022], dtype=np.int32)


SMPL_PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21], dtype=np.int32)
SMPLH_PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19,
                          20, 22, 23, 20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35,
                          21, 37, 38, 21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50], dtype=np.int32)
# smplx/vertex_ids.py ['smplh'] (shared by SMPL and SMPL-H, V = 6890) in VertexJointSelector order [upstream-knowledge]
SMPLH_EXTRA_VERTEX_IDS = np.array([332, 6260, 2800, 4071, 583, 3216, 3226, 3387, 6617, 6624, 6787,
                                   2746, 2319, 2445, 2556, 2673, 6191, 5782, 5905, 6016, 6133], dtype=np.int32)


def make_synthetic_asset(model_type="smplx", seed=0, num_betas=10, num_expressions=10, nnz_per_vertex=4):
    """SMPL (V 6890, J 24), SMPL-H (V 6890, J 52) or SMPL-X (V 10475, J 55) shaped asset with the real kinematic tree."""
    if model_type == "smplx":
        return make_synthetic_smplx_asset(seed=seed, num_betas=num_betas, num_expressions=num_expressions, nnz_per_vertex=nnz_per_vertex)
    parents = {"smpl": SMPL_PARENTS, "smplh": SMPLH_PARENTS}[model_type]
    a = make_synthetic_smplx_asset(seed=seed, num_vertices=6890, num_betas=num_betas, num_expressions=0, nnz_per_vertex=nnz_per_vertex,
                                   parents=parents)
    a.update(extra_joint_vertex_ids=SMPLH_EXTRA_VERTEX_IDS.copy(), lmk_faces_idx=np.zeros((0,), np.int32),
             lmk_bary_coords=np.zeros((0, 3), np.float32), model_type=model_type)
    return a


def make_synthetic_smplx_asset(seed=0, num_vertices=10475, num_betas=10, num_expressions=10, nnz_per_vertex=4, parents=None,
                               coherent_skinning=False):
    """``coherent_skinning``: the default asset skins every vertex to FOUR RANDOM joints of 55 -- the worst case for any kernel that
    gathers per-joint transforms (no two neighbouring vertices share a joint; the per-lane LDS reads of the skinning kernels collide in
    half of their cycles).  A real SMPL-X template is spatially coherent: a vertex follows the bone it sits on and that
    bone's neighbours in the tree, and consecutive vertex ids mostly lie on the same body part.  With this flag the vertices are cut
    into J contiguous id ranges, range j follows joint j, its parent and up to two of its tree neighbours (same arithmetic, same
    sizes -- only the index pattern changes).  Used by the benchmark's ``*_coherent_skinning`` legs; tests keep the default."""
    rs = np.random.RandomState(seed)
    parents = SMPLX_PARENTS if parents is None else parents
    V, J = num_vertices, len(parents)
    v_template = (rs.standard_normal((V, 3)) * np.array([0.25, 0.55, 0.12])).astype(np.float32)
    shapedirs = (rs.standard_normal((V, 3, num_betas + num_expressions)) * 0.01).astype(np.float32)
    posedirs = (rs.standard_normal(((J - 1) * 9, V * 3)) * 0.005).astype(np.float32)
    # joint regressor: each joint = convex combination of 32 random vertices
    J_regressor = np.zeros((J, V), dtype=np.float32)
    for j in range(J):
        idx = rs.choice(V, 32, replace=False)
        w = rs.uniform(0.1, 1.0, 32)
        J_regressor[j, idx] = (w / w.sum()).astype(np.float32)
    # skinning weights: nnz_per_vertex non-zeros per vertex, rows sum to one (real SMPL-X has <= 4)
    weights = np.zeros((V, J), dtype=np.float32)
    children = {j: [c for c in range(J) if parents[c] == j] for j in range(J)}
    for v in range(V):
        idx = rs.choice(J, nnz_per_vertex, replace=False)
        w = rs.uniform(0.05, 1.0, nnz_per_vertex)
        if coherent_skinning:
            j = min(J - 1, v * J // V)
            near = [j] + ([int(parents[j])] if parents[j] >= 0 else []) + children[j]
            if parents[j] >= 0:
                near += [c for c in children[int(parents[j])] if c != j] + ([int(parents[int(parents[j])])] if parents[int(parents[j])] >= 0 else [])
            near = list(dict.fromkeys(near))
            k = 0
            while len(near) < nnz_per_vertex:                  # (tiny trees: pad with any other joint)
                if k not in near:
                    near.append(k)
                k += 1
            idx = np.array(near[:nnz_per_vertex])
        weights[v, idx] = (w / w.sum()).astype(np.float32)
    faces = rs.randint(0, V, size=(20908, 3)).astype(np.int32)
    lmk_faces_idx = rs.randint(0, faces.shape[0], size=51).astype(np.int32)
    bary = rs.uniform(0.05, 1.0, size=(51, 3))
    lmk_bary_coords = (bary / bary.sum(axis=1, keepdims=True)).astype(np.float32)
    return dict(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs, J_regressor=J_regressor,
                parents=np.asarray(parents).astype(np.int64), weights=weights, faces=faces, lmk_faces_idx=lmk_faces_idx,
                lmk_bary_coords=lmk_bary_coords, extra_joint_vertex_ids=SMPLX_EXTRA_VERTEX_IDS.copy() % V,
                num_betas=num_betas, num_expressions=num_expressions, model_type="smplx")


def load_smplx_npz(path, num_betas=10, num_expressions=10):
    """Asset dictionary from an official SMPL-X model file (``.npz`` / ``.pkl``, v1.0 or v1.1): see ``assets.load_model_file``."""
    from .assets import load_model_file
    return load_model_file(path, "smplx", num_betas, num_expressions)


def load_model_npz(path, model_type="smplx", num_betas=10, num_expressions=10):
    """Asset dictionary from an official SMPL / SMPL-H / SMPL-X model file or directory (name kept from round 2; ``.pkl`` is read
    too): ``assets.load_model_file`` restates what smplx 0.1.28 does with the file (lib/body_model/body_model.py:39-62)."""
    from .assets import load_model_file
    return load_model_file(path, model_type, num_betas, num_expressions)
