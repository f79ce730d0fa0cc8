"""Body-part index sets and skeleton tables -- counterpart of the reference's
lib/body_model/utils.py (BODY_JOINT_NAMES :11-34, BodyPartIndices :39-47, BodySegIndices :50-61,
smpl_to_openpose :68-177, get_smpl_skeleton :180-205).  Data comes from tables.json."""
import numpy as np

from .constants import _T

BODY_JOINT_NAMES = _T["BODY_JOINT_NAMES"]
name_to_index = {name: index - 1 for index, name in enumerate(BODY_JOINT_NAMES)}   # pelvis excluded


class _IndexSets:
    def __init__(self, table):
        for part, idx in table.items():
            setattr(self, part, list(idx))


BodyPartIndices = _IndexSets(_T["BodyPartIndices"])    # joint indices into the 21 body poses
BodySegIndices = _IndexSets(_T["BodySegIndices"])      # SMPL-X vertex indices per body part


def smpl_to_openpose(model_type="smplx", use_hands=True, use_face=True, use_face_contour=False, openpose_format="coco25"):
    """Permutation SMPL-family joints -> OpenPose (utils.py:68-177) for every argument combination the reference accepts:
    model_type in {smpl, smplh, smplx} x use_hands x use_face x use_face_contour x {coco25 (any case), coco19}.  The reference
    concatenates a body map, two 21-entry hand maps (SMPL-H / SMPL-X) and an arange of 51 (+ 17 contour) face indices (SMPL-X);
    tables.json holds those blocks as tools/gen_tables.py read them off the reference's outputs (bit-exact: golden g24)."""
    fmt = openpose_format
    if fmt.lower() == "coco25":               # (utils.py:91 lower-cases for coco25 only; :135 compares 'coco19' as given)
        fmt = "coco25"
    elif fmt != "coco19":
        raise ValueError("Unknown joint format: {}".format(openpose_format))
    key = f"{fmt}/{model_type}"
    if key not in _T["smpl_to_openpose_blocks"]:
        raise ValueError("Unknown model type: {}".format(model_type))
    b = _T["smpl_to_openpose_blocks"][key]
    mapping = [np.array(b["body"], dtype=np.int32)]
    if model_type == "smpl":                  # (a single literal in the reference: the flags are not consulted)
        return mapping[0]
    if use_hands:
        mapping += [np.array(b["lhand"], dtype=np.int32), np.array(b["rhand"], dtype=np.int32)]
    if use_face and b["face_start"] >= 0:
        mapping.append(np.arange(b["face_start"], b["face_start"] + 51 + 17 * int(bool(use_face_contour)), dtype=np.int32))
    return np.concatenate(mapping)


def get_smpl_skeleton():
    return np.array(_T["smpl_skeleton"])


def skeleton_parents(n=22):
    """parents[] of the first n SMPL joints derived from the bone list."""
    parents = -np.ones(n, dtype=np.int64)
    for a, b in _T["smpl_skeleton"]:
        if b < n:
            parents[b] = a
    return parents
