#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (lives beside gen_golden.py; imports oracle/).  Pin the FK / LBS oracle (oracle/fk_ref.py) to the reference's real dependency, ``smplx==0.1.28``.

The reference reaches the body model only through smplx (lib/body_model/body_model.py:4-5,39-62, smpl.py:3,52).  smplx is not
installable in the build container (no network), so ``oracle/fk_ref.py`` is a restatement whose parity is UNPINNED.  This script
turns it green wherever smplx CAN be imported -- and it needs NO licensed model file:

    pip install smplx==0.1.28          # (a networked machine; torch + numpy are its only hard dependencies)
    python tests/golden/pin_fk_parity.py      # writes tests/golden/g23_smplx_pin.npz

1. write the synthetic SMPL-X / SMPL-H / SMPL assets (``dposer_amd.body_model.synthetic``) to files in the official key layout
   (tests/asset_files.py: v1.1 ``[V,3,400]`` shapedirs, uint32 kintree_table, ...);
2. construct the REFERENCE's own objects on those files -- ``smplx.SMPLX / SMPLH / SMPL`` with exactly the keyword set of
   body_model.py:30-37 (``use_pca=False, flat_hand_mean=True``, SMPL-H through the ``data_struct`` route of :44-57) -- and run them
   on seeded inputs (all pose segments, betas, expression, transl);
3. run ``oracle/fk_ref.py`` through this repository's loader on the same files, print the differences, and store inputs +
   smplx outputs as the golden ``g23`` (arrays only; the generating asset is re-made from its seed by the test).

``tests/test_oracle_golden.py::test_fk_oracle_against_smplx_golden`` (CPU) and ``tests/test_gpu_assets.py::test_lbs_against_smplx_golden``
(GPU) pick the golden up when it exists; until then they skip with "parity unpinned".

``--model PATH --model-type smplx`` additionally checks a licensed file (never stored: the golden of that run keeps the file's
sha256, the inputs and smplx's outputs, and the tests look for the file in ``$DPOSER_SMPLX_MODEL``)."""
import argparse
import hashlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = (("smplx", "smplx_v1.1", "npz", 10, 10), ("smplx", "smplx_v1.0", "npz", 10, 10), ("smplh", "smplh_amass", "npz", 16, 0),
         ("smpl", "smpl", "pkl", 10, 0))
ASSET_SEED = 41
B = 6


def make_inputs(model_type, nb, ne, seed):
    rs = np.random.RandomState(seed)
    nbody = 23 if model_type == "smpl" else 21
    d = dict(global_orient=rs.standard_normal((B, 3)) * 0.5, body_pose=rs.standard_normal((B, nbody * 3)) * 0.4,
             betas=rs.standard_normal((B, nb)) * 0.7, transl=rs.standard_normal((B, 3)))
    if model_type == "smplx":
        d.update(jaw_pose=rs.standard_normal((B, 3)) * 0.2, leye_pose=rs.standard_normal((B, 3)) * 0.2, reye_pose=rs.standard_normal((B, 3)) * 0.2,
                 expression=rs.standard_normal((B, ne)) * 0.7)
    if model_type in ("smplh", "smplx"):
        d.update(left_hand_pose=rs.standard_normal((B, 45)) * 0.3, right_hand_pose=rs.standard_normal((B, 45)) * 0.3)
    return {k: v.astype(np.float32) for k, v in d.items()}


def full_pose_and_shape(model_type, d):
    order = {"smpl": ("global_orient", "body_pose"), "smplh": ("global_orient", "body_pose", "left_hand_pose", "right_hand_pose"),
             "smplx": ("global_orient", "body_pose", "jaw_pose", "leye_pose", "reye_pose", "left_hand_pose", "right_hand_pose")}[model_type]
    full = np.concatenate([d[k] for k in order], axis=1)
    shape = d["betas"] if model_type != "smplx" else np.concatenate([d["betas"], d["expression"]], axis=1)
    return full, shape


def run_smplx(path, model_type, nb, ne, d):
    """The reference's construction (lib/body_model/body_model.py:30-62), then one forward."""
    import torch
    from smplx import SMPL, SMPLH, SMPLX
    from smplx.utils import Struct
    kwargs = dict(model_type=model_type, num_betas=nb, batch_size=B, num_expression_coeffs=ne, use_pca=False, flat_hand_mean=True)
    if model_type == "smpl":
        bm = SMPL(path, **kwargs)
    elif model_type == "smplh":
        smpl_dict = np.load(path, encoding="latin1", allow_pickle=True)
        ds = Struct(**smpl_dict)
        ds.hands_componentsl = np.zeros((0))
        ds.hands_componentsr = np.zeros((0))
        ds.hands_meanl = np.zeros((15 * 3))
        ds.hands_meanr = np.zeros((15 * 3))
        V, D, S = ds.shapedirs.shape
        ds.shapedirs = np.concatenate([ds.shapedirs, np.zeros((V, D, SMPL.SHAPE_SPACE_DIM - S))], axis=-1)
        kwargs["data_struct"] = ds
        bm = SMPLH(path, **kwargs)
    else:
        bm = SMPLX(path, **kwargs)
    with torch.no_grad():
        o = bm(**{k: torch.tensor(v) for k, v in d.items()}, return_full_pose=True)
    return o.vertices.numpy(), o.joints.numpy(), o.full_pose.numpy()


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "g23_smplx_pin.npz"))
    ap.add_argument("--model", default=None, help="optional licensed model file (never stored)")
    ap.add_argument("--model-type", default="smplx", choices=["smpl", "smplh", "smplx"])
    args = ap.parse_args()
    try:
        import smplx
    except ImportError:
        sys.exit("smplx is not importable here: run this script where `pip install smplx==0.1.28` is possible (see the docstring)")
    from asset_files import write_npz, write_pkl
    from dposer_amd.body_model import assets
    from dposer_amd.body_model.synthetic import make_synthetic_asset
    from oracle import fk_ref
    out = {"smplx_version": np.array(getattr(smplx, "__version__", "unknown")), "asset_seed": np.array(ASSET_SEED), "cases": np.array([c[1] for c in CASES])}
    worst = 0.0
    with tempfile.TemporaryDirectory() as tmp:
        for i, (mt, layout, ext, nb, ne) in enumerate(CASES):
            asset = make_synthetic_asset(mt, seed=ASSET_SEED + i, num_betas=nb, num_expressions=ne)
            path = os.path.join(tmp, f"{layout}.{ext}")
            (write_npz if ext == "npz" else write_pkl)(asset, path, layout)
            d = make_inputs(mt, nb, ne, seed=100 + i)
            v, j, fp = run_smplx(path, mt, nb, ne, d)
            loaded = assets.load_model_file(path, mt, nb, ne)
            full, shape = full_pose_and_shape(mt, d)
            assert np.array_equal(fp, full), "full_pose order differs from smplx"
            v_o, j_o, _ = fk_ref.model_forward(loaded, full.astype(np.float64), shape=shape.astype(np.float64), transl=d["transl"].astype(np.float64))
            ev, ej = float(np.abs(v_o - v).max()), float(np.abs(j_o - j).max())
            worst = max(worst, ev, ej)
            print(f"{layout:12s} smplx vs oracle/fk_ref.py (fp64): vertices {ev:.2e}, joints {ej:.2e}  [{j.shape[1]} joints]")
            for k, a in d.items():
                out[f"{layout}/in/{k}"] = a
            out[f"{layout}/vertices"], out[f"{layout}/joints"] = v.astype(np.float32), j.astype(np.float32)
        if args.model:
            mt = args.model_type
            nb, ne = (10, 10 if mt == "smplx" else 0)
            d = make_inputs(mt, nb, ne, seed=999)
            v, j, _ = run_smplx(args.model, mt, nb, ne, d)
            loaded = assets.load_model_file(args.model, mt, nb, ne)
            full, shape = full_pose_and_shape(mt, d)
            v_o, j_o, _ = fk_ref.model_forward(loaded, full.astype(np.float64), shape=shape.astype(np.float64), transl=d["transl"].astype(np.float64))
            ev, ej = float(np.abs(v_o - v).max()), float(np.abs(j_o - j).max())
            worst = max(worst, ev, ej)
            print(f"licensed {mt} file: smplx vs oracle: vertices {ev:.2e}, joints {ej:.2e}; influences per vertex {assets.max_skinning_influences(loaded)}")
            out["licensed/sha256"] = np.array(hashlib.sha256(open(args.model, "rb").read()).hexdigest())
            out["licensed/model_type"] = np.array(mt)
            for k, a in d.items():
                out[f"licensed/in/{k}"] = a
            out["licensed/vertices"], out["licensed/joints"] = v.astype(np.float32), j.astype(np.float32)
    np.savez_compressed(args.out, **out)
    print(f"wrote {args.out}; worst difference {worst:.2e} ({'PINNED' if worst < 1e-5 else 'MISMATCH: the restatement is wrong somewhere'})")
    sys.exit(0 if worst < 1e-5 else 1)


if __name__ == "__main__":
    main()
