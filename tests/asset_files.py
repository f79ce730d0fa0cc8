"""Test infrastructure: write an asset dictionary out in the key layout of the official SMPL-family model files, so that the
file door of ``BodyModel(bm_path=...)`` (reference lib/body_model/body_model.py:14-66 -> smplx 0.1.28 ``body_models.py``) can be
driven without a licensed file.  The arrays are the synthetic asset's; the KEYS, SHAPES, DTYPES and the slots the loader must
pick its directions from are the official files' (junk fills every slot the loader must not read)."""
import pickle

import numpy as np


def official_arrays(asset, layout, seed=123):
    """key -> array as the official file of ``layout`` stores them.

    layouts: 'smplx_v1.1' (shapedirs [V,3,400]: betas at [0:300], expression at [300:400]), 'smplx_v1.0' ([V,3,20]: 10 + 10),
    'smplh_amass' (AMASS SMPL+H ``model.npz``: 16 betas), 'smpl' (10 betas)."""
    rs = np.random.RandomState(seed)
    V = asset["v_template"].shape[0]
    J = asset["weights"].shape[1]
    nb, ne = int(asset["num_betas"]), int(asset["num_expressions"])
    sd = np.asarray(asset["shapedirs"], np.float64)
    if layout == "smplx_v1.1":
        full = rs.standard_normal((V, 3, 400)) * 0.01            # junk everywhere the loader must not look
        full[:, :, :nb] = sd[:, :, :nb]
        full[:, :, 300:300 + ne] = sd[:, :, nb:nb + ne]
    elif layout == "smplx_v1.0":
        assert nb <= 10 and ne <= 10
        full = rs.standard_normal((V, 3, 20)) * 0.01
        full[:, :, :nb] = sd[:, :, :nb]
        full[:, :, 10:10 + ne] = sd[:, :, nb:nb + ne]
    elif layout == "smplh_amass":
        assert nb <= 16
        full = rs.standard_normal((V, 3, 16)) * 0.01
        full[:, :, :nb] = sd[:, :, :nb]
    elif layout == "smpl":
        assert nb <= 10
        full = rs.standard_normal((V, 3, 10)) * 0.01
        full[:, :, :nb] = sd[:, :, :nb]
    else:
        raise ValueError(layout)
    P = asset["posedirs"].shape[0]
    posedirs = np.asarray(asset["posedirs"], np.float64).T.reshape(V, 3, P)
    kt = np.zeros((2, J), dtype=np.uint32)
    par = np.asarray(asset["parents"]).astype(np.int64).copy()
    par[0] = 2 ** 32 - 1                                          # the root's parent in the official uint32 tables
    kt[0] = par.astype(np.uint32)
    kt[1] = np.arange(J, dtype=np.uint32)
    d = dict(v_template=np.asarray(asset["v_template"], np.float64), shapedirs=full, posedirs=posedirs,
             J_regressor=np.asarray(asset["J_regressor"], np.float64), weights=np.asarray(asset["weights"], np.float64),
             kintree_table=kt, f=np.asarray(asset["faces"]).astype(np.uint32))
    if layout.startswith("smplx"):
        d.update(lmk_faces_idx=np.asarray(asset["lmk_faces_idx"]).astype(np.int64),
                 lmk_bary_coords=np.asarray(asset["lmk_bary_coords"], np.float64),
                 dynamic_lmk_faces_idx=rs.randint(0, 100, size=(79, 17)).astype(np.int64),
                 dynamic_lmk_bary_coords=rs.uniform(size=(79, 17, 3)),
                 hands_componentsl=rs.standard_normal((45, 45)), hands_componentsr=rs.standard_normal((45, 45)),
                 hands_meanl=rs.standard_normal(45), hands_meanr=rs.standard_normal(45),       # flat_hand_mean=True: must be ignored
                 joint2num=np.array({"Pelvis": 0, "L_Hip": 1}, dtype=object), vt=rs.uniform(size=(16, 2)), ft=rs.randint(0, 16, size=(8, 3)))
    if layout == "smplh_amass":
        d.update(hands_componentsl=rs.standard_normal((45, 45)), hands_componentsr=rs.standard_normal((45, 45)),
                 hands_meanl=rs.standard_normal(45), hands_meanr=rs.standard_normal(45), bs_style=np.array("lbs"), bs_type=np.array("lrotmin"))
    return d


def write_npz(asset, path, layout, seed=123):
    np.savez(path, **official_arrays(asset, layout, seed))
    return path


def write_pkl(asset, path, layout, seed=123, sparse_regressor=True, protocol=2):
    """A ``.pkl`` of plain arrays, the joint regressor as a scipy-sparse matrix (what smplx's ``tools/clean_ch.py`` leaves behind)."""
    d = official_arrays(asset, layout, seed)
    if sparse_regressor:
        import scipy.sparse
        d["J_regressor"] = scipy.sparse.csc_matrix(d["J_regressor"])
    d = {k: (v if not (isinstance(v, np.ndarray) and v.dtype == object) else v.item()) for k, v in d.items()}
    with open(path, "wb") as f:
        pickle.dump(d, f, protocol=protocol)
    return path


ASSET_KEYS_EXACT = ("v_template", "shapedirs", "posedirs", "J_regressor", "weights", "faces", "lmk_faces_idx", "lmk_bary_coords", "extra_joint_vertex_ids")


def assert_same_asset(loaded, asset):
    """The loader must hand back exactly the arrays the file was written from (float32 of the float64 the file stores)."""
    for k in ASSET_KEYS_EXACT:
        a, b = np.asarray(loaded[k]), np.asarray(asset[k])
        assert a.shape == b.shape, (k, a.shape, b.shape)
        assert np.array_equal(a, b.astype(a.dtype)), k
    assert np.array_equal(np.asarray(loaded["parents"]), np.asarray(asset["parents"]))
    assert loaded["parents"][0] == -1
    assert (loaded["num_betas"], loaded["num_expressions"], loaded["model_type"]) == (asset["num_betas"], asset["num_expressions"], asset["model_type"])
