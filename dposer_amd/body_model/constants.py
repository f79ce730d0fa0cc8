"""Joint naming / index tables -- counterpart of the reference's lib/body_model/constants.py:12-131.
All integer tables are loaded from ``tables.json`` (extracted from the reference by
tools/gen_tables.py, pinned bit-exactly by tests/test_tables.py)."""
import json
import os
from os.path import join

curr_dir = os.path.dirname(os.path.abspath(__file__))
SMPL_MEAN_PATH = join(curr_dir, "smpl_mean_params.npz")      # user-supplied asset (not shipped)
BEND_POSE_PATH = join(curr_dir, "../data/bend_pose.npz")

with open(join(curr_dir, "tables.json")) as _f:
    _T = json.load(_f)

CROP_IMG_HEIGHT = 256
CROP_IMG_WIDTH = 192
CROP_ASPECT_RATIO = CROP_IMG_HEIGHT / float(CROP_IMG_WIDTH)
IMG_NORM_MEAN = _T["IMG_NORM_MEAN"]
IMG_NORM_STD = _T["IMG_NORM_STD"]
FOCAL_LENGTH = _T["FOCAL_LENGTH"]
IMG_RES = _T["IMG_RES"]

JOINT_NAMES = _T["JOINT_NAMES"]                      # 25 OpenPose + 24 ground-truth joints
JOINT_IDS = {name: i for i, name in enumerate(JOINT_NAMES)}
JOINT_MAP = _T["JOINT_MAP"]                          # name -> SMPL joint index
H36M_TO_J17 = _T["H36M_TO_J17"]
H36M_TO_J14 = _T["H36M_TO_J14"]
J24_TO_J17 = _T["J24_TO_J17"]
J24_TO_J14 = _T["J24_TO_J14"]
SMPL_JOINTS_FLIP_PERM = _T["SMPL_JOINTS_FLIP_PERM"]
SMPL_POSE_FLIP_PERM = _T["SMPL_POSE_FLIP_PERM"]
J24_FLIP_PERM = _T["J24_FLIP_PERM"]
J49_FLIP_PERM = _T["J49_FLIP_PERM"]
SMPLX_OPENPOSE_25 = _T["SMPLX_OPENPOSE_25"]          # lib/body_model/smpl.py:55-57
