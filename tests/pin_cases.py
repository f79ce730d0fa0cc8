"""Test infrastructure shared by the consumers of golden g23 (tests/golden/pin_fk_parity.py): rebuild the synthetic model file of a case
from its seed and map the golden's smplx keyword inputs onto ``BodyModel.forward``'s."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def case_inputs(g, layout):
    pre = f"{layout}/in/"
    return {k[len(pre):]: g[k] for k in g.files if k.startswith(pre)}


def to_body_model_kwargs(d):
    """smplx keywords -> lib/body_model/body_model.py:68-88 keywords."""
    kw = dict(root_orient=d["global_orient"], pose_body=d["body_pose"], betas=d["betas"], trans=d["transl"])
    if "jaw_pose" in d:
        kw.update(pose_jaw=d["jaw_pose"], pose_eye=np.concatenate([d["leye_pose"], d["reye_pose"]], axis=1), expression=d["expression"])
    if "left_hand_pose" in d:
        kw["pose_hand"] = np.concatenate([d["left_hand_pose"], d["right_hand_pose"]], axis=1)
    return kw


def build_case(g, i, layout, tmp_path):
    """(BodyModel on the case's file, forward kwargs as numpy, the loaded asset dictionary)."""
    import pin_fk_parity as P
    from asset_files import write_npz, write_pkl
    from dposer_amd.body_model import assets
    from dposer_amd.body_model.body_model import BodyModel
    from dposer_amd.body_model.synthetic import make_synthetic_asset
    mt, lay, ext, nb, ne = P.CASES[i]
    assert lay == layout
    asset = make_synthetic_asset(mt, seed=int(g["asset_seed"]) + i, num_betas=nb, num_expressions=ne)
    path = os.path.join(str(tmp_path), f"{layout}.{ext}")
    (write_npz if ext == "npz" else write_pkl)(asset, path, layout)
    bm = BodyModel(bm_path=path, num_betas=nb, num_expressions=ne, model_type=mt)
    return bm, to_body_model_kwargs(case_inputs(g, layout)), assets.load_model_file(path, mt, nb, ne)
