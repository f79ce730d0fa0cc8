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
    """Permutation SMPL-family joints -> OpenPose (utils.py:68-177); the default argument combination
    of the reference is tabulated, other combinations are not built."""
    if openpose_format.lower() != "coco25" or not (use_hands and use_face) or use_face_contour:
        raise NotImplementedError("only the reference's default smpl_to_openpose(model_type) tables are shipped")
    if model_type not in _T["smpl_to_openpose"]:
        raise ValueError("Unknown model type: {}".format(model_type))
    return np.array(_T["smpl_to_openpose"][model_type], dtype=np.int32)


def get_smpl_skeleton():
    return np.array(_T["smpl_skeleton"])


def skeleton_parents(n=22):
    """parents[] of the first n SMPL joints derived from the bone list."""
    parents = -np.ones(n, dtype=np.int64)
    for a, b in _T["smpl_skeleton"]:
        if b < n:
            parents[b] = a
    return parents
