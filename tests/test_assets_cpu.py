"""The model-file door on the CPU: ``dposer_amd.body_model.assets`` against files written in the official key layouts
(tests/asset_files.py) -- what smplx 0.1.28 does with a file when the reference constructs ``BodyModel(bm_path=...)``
(lib/body_model/body_model.py:14-66).  No compute: the GPU half is tests/test_gpu_assets.py."""
import pickle
import sys
import types

import numpy as np
import pytest

from asset_files import assert_same_asset, official_arrays, write_npz, write_pkl
from dposer_amd.body_model import assets
from dposer_amd.body_model.synthetic import make_synthetic_asset


@pytest.fixture(scope="module")
def smplx_asset():
    return make_synthetic_asset("smplx", seed=3)


@pytest.mark.parametrize("layout", ["smplx_v1.1", "smplx_v1.0"])
@pytest.mark.parametrize("ext", ["npz", "pkl"])
def test_smplx_files_load_to_the_arrays_they_were_written_from(tmp_path, smplx_asset, layout, ext):
    path = str(tmp_path / f"SMPLX_NEUTRAL.{ext}")
    (write_npz if ext == "npz" else write_pkl)(smplx_asset, path, layout)
    assert_same_asset(assets.load_model_file(path, "smplx", 10, 10), smplx_asset)
    # a directory resolves to smplx's default file name
    assert_same_asset(assets.load_model_file(str(tmp_path), "smplx", 10, 10), smplx_asset)


def test_smplx_shape_space_clamps_like_smplx(tmp_path, smplx_asset):
    p10 = write_npz(smplx_asset, str(tmp_path / "v10.npz"), "smplx_v1.0")
    a = assets.load_model_file(p10, "smplx", num_betas=16, num_expressions=50)          # a 10 + 10 file: both clamp to 10
    assert (a["num_betas"], a["num_expressions"], a["shapedirs"].shape[2]) == (10, 10, 20)
    a = assets.load_model_file(p10, "smplx", num_betas=4, num_expressions=3)            # fewer: betas [0:4], expression [10:13]
    full = official_arrays(smplx_asset, "smplx_v1.0")["shapedirs"]
    assert np.array_equal(a["shapedirs"], np.concatenate([full[:, :, :4], full[:, :, 10:13]], axis=2).astype(np.float32))
    p11 = write_npz(smplx_asset, str(tmp_path / "v11.npz"), "smplx_v1.1")
    a = assets.load_model_file(p11, "smplx", num_betas=16, num_expressions=50)          # a 300 + 100 file serves them
    full = official_arrays(smplx_asset, "smplx_v1.1")["shapedirs"]
    assert a["shapedirs"].shape[2] == 66
    assert np.array_equal(a["shapedirs"], np.concatenate([full[:, :, :16], full[:, :, 300:350]], axis=2).astype(np.float32))
    a = assets.load_model_file(p11, "smplx", num_betas=400, num_expressions=400)
    assert (a["num_betas"], a["num_expressions"]) == (300, 100)


def test_smplh_amass_npz_pads_the_shape_space_like_the_reference(tmp_path):
    asset = make_synthetic_asset("smplh", seed=4, num_betas=16)
    path = write_npz(asset, str(tmp_path / "model.npz"), "smplh_amass")
    assert_same_asset(assets.load_model_file(path, "smplh", num_betas=16), asset)
    a = assets.load_model_file(path, "smplh", num_betas=20)             # body_model.py:53-56: zero-padded to 300 before smplx clamps
    assert a["num_betas"] == 20 and np.array_equal(a["shapedirs"][:, :, :16], asset["shapedirs"]) and not a["shapedirs"][:, :, 16:].any()


def test_smpl_pkl_with_sparse_regressor_and_uint32_root(tmp_path):
    asset = make_synthetic_asset("smpl", seed=5)
    path = write_pkl(asset, str(tmp_path / "SMPL_NEUTRAL.pkl"), "smpl", sparse_regressor=True)
    raw = assets.read_model_file(path)
    assert hasattr(raw["J_regressor"], "todense") and raw["kintree_table"].dtype == np.uint32 and raw["kintree_table"][0, 0] == 2 ** 32 - 1
    assert_same_asset(assets.load_model_file(path, "smpl", num_betas=10), asset)
    assert assets.load_model_file(path, "smpl", num_betas=16)["num_betas"] == 10        # a 10-beta file: smplx clamps, no padding for SMPL
    assert_same_asset(assets.load_model_file(str(tmp_path), "smpl"), asset)


def test_chumpy_pickles_are_read_without_chumpy(tmp_path):
    """Legacy ``.pkl`` files wrap some arrays in chumpy objects.  A stand-in ``chumpy`` package exists only while the file is
    WRITTEN; the loader then runs without it."""
    asset = make_synthetic_asset("smpl", seed=6)
    d = official_arrays(asset, "smpl")
    mod, sub = types.ModuleType("chumpy"), types.ModuleType("chumpy.ch")

    class Ch:
        def __init__(self, x):
            self.x = x
            self._dirty_vars = set()
    Ch.__module__, Ch.__qualname__ = "chumpy.ch", "Ch"
    sub.Ch = Ch
    mod.ch = sub
    sys.modules["chumpy"], sys.modules["chumpy.ch"] = mod, sub
    try:
        for k in ("v_template", "shapedirs", "posedirs", "weights"):
            d[k] = Ch(d[k])
        path = str(tmp_path / "basicModel.pkl")
        with open(path, "wb") as f:
            pickle.dump(d, f, protocol=2)
    finally:
        del sys.modules["chumpy"], sys.modules["chumpy.ch"]
    with pytest.raises(ModuleNotFoundError):
        pickle.load(open(path, "rb"), encoding="latin1")
    assert_same_asset(assets.load_model_file(path, "smpl"), asset)


def test_wrong_files_are_refused_with_a_reason(tmp_path, smplx_asset):
    smplh = make_synthetic_asset("smplh", seed=7, num_betas=16)
    p = write_npz(smplh, str(tmp_path / "h.npz"), "smplh_amass")
    with pytest.raises(assets.ModelFileError, match="not a smplx file|lmk_faces_idx|shapedirs has 16"):
        assets.load_model_file(p, "smplx")
    d = official_arrays(smplx_asset, "smplx_v1.0")
    del d["weights"]
    np.savez(str(tmp_path / "broken.npz"), **d)
    with pytest.raises(assets.ModelFileError, match="weights"):
        assets.load_model_file(str(tmp_path / "broken.npz"), "smplx")
    with pytest.raises(FileNotFoundError):
        assets.load_model_file(str(tmp_path / "nope.npz"), "smplx")
    with pytest.raises(assets.ModelFileError, match="extension"):
        open(str(tmp_path / "m.json"), "w").write("{}")
        assets.load_model_file(str(tmp_path / "m.json"), "smplx")


def test_influence_count_reported(smplx_asset):
    assert assets.max_skinning_influences(smplx_asset) == 4
    assert assets.max_skinning_influences(make_synthetic_asset("smpl", seed=1, nnz_per_vertex=7)) == 7


def test_pin_script_plumbing_with_a_stand_in_for_smplx(tmp_path, monkeypatch):
    """tests/golden/pin_fk_parity.py end to end with ``run_smplx`` replaced by the oracle on the loaded file (smplx itself cannot be
    imported here): files are written, keyword sets map onto the full pose in smplx's order, the golden has the keys its consumers
    read.  What this cannot show -- that smplx agrees -- is exactly what the script is for."""
    import pin_fk_parity as P
    from oracle import fk_ref
    from pin_cases import build_case, case_inputs

    def fake_run_smplx(path, model_type, nb, ne, d):
        loaded = assets.load_model_file(path, model_type, nb, ne)
        full, shape = P.full_pose_and_shape(model_type, d)
        v, j, _ = fk_ref.model_forward(loaded, full.astype(np.float64), shape=shape.astype(np.float64), transl=d["transl"].astype(np.float64))
        return v, j, full
    monkeypatch.setattr(P, "run_smplx", fake_run_smplx)
    monkeypatch.setitem(sys.modules, "smplx", types.SimpleNamespace(__version__="stand-in"))
    monkeypatch.setattr(P, "CASES", P.CASES[1:2] + P.CASES[3:4])          # (the 10 + 10 SMPL-X file and the SMPL .pkl: the small ones)
    out = str(tmp_path / "g23.npz")
    monkeypatch.setattr(sys, "argv", ["pin_fk_parity.py", "--out", out])
    with pytest.raises(SystemExit) as e:
        P.main()
    assert e.value.code == 0
    g = np.load(out, allow_pickle=False)
    assert [str(c) for c in g["cases"]] == ["smplx_v1.0", "smpl"]
    for i, layout in enumerate([str(c) for c in g["cases"]]):
        bm_kwargs = build_case(g, i, layout, tmp_path)[1]
        d = case_inputs(g, layout)
        assert set(bm_kwargs) >= {"root_orient", "pose_body", "betas", "trans"} and g[f"{layout}/vertices"].shape[0] == P.B
        assert ("pose_hand" in bm_kwargs) == ("left_hand_pose" in d)
