#!/usr/bin/env python3
"""Extract the reference's integer index tables (pure data) into dposer_amd/body_model/tables.json.

Run in the build container only (imports /root/reference read-only):  python tools/gen_tables.py
Sources: lib/body_model/constants.py:34-131 (JOINT_NAMES, JOINT_MAP, flip permutations, selectors),
lib/body_model/utils.py:11-61 (BODY_JOINT_NAMES, BodyPartIndices, BodySegIndices from
smplx_vert_segmentation.json), :68-177 (smpl_to_openpose), :180-205 (get_smpl_skeleton),
lib/body_model/smpl.py:55-57 (SMPL-X OpenPose joint list).  tests/test_host_cpu.py (test_index_tables_bit_exact and neighbours, golden g9) pins the result
bit-exactly against tests/golden/g9_tables.npz.
"""
import json
import os
import sys
from unittest import mock

sys.path.insert(0, "/root/reference")
for _n in ["torchgeometry", "smplx", "smplx.utils", "smplx.body_models", "cv2"]:
    sys.modules[_n] = mock.MagicMock()

import lib.body_model.constants as C  # noqa: E402
import lib.body_model.utils as U  # noqa: E402

PARTS = ["left_leg", "right_leg", "left_arm", "right_arm", "trunk", "hands", "legs", "arms"]
out = {
    "JOINT_NAMES": list(C.JOINT_NAMES),
    "JOINT_MAP": {k: int(v) for k, v in C.JOINT_MAP.items()},
    "FOCAL_LENGTH": C.FOCAL_LENGTH if hasattr(C, "FOCAL_LENGTH") else None,
    "IMG_RES": C.IMG_RES if hasattr(C, "IMG_RES") else None,
    "IMG_NORM_MEAN": list(C.IMG_NORM_MEAN) if hasattr(C, "IMG_NORM_MEAN") else None,
    "IMG_NORM_STD": list(C.IMG_NORM_STD) if hasattr(C, "IMG_NORM_STD") else None,
    "BODY_JOINT_NAMES": list(U.BODY_JOINT_NAMES),
    "BodyPartIndices": {p: [int(i) for i in getattr(U.BodyPartIndices, p)] for p in PARTS},
    "BodySegIndices": {p: [int(i) for i in getattr(U.BodySegIndices, p)] for p in PARTS},
    "SMPLX_OPENPOSE_25": [55, 12, 17, 19, 21, 16, 18, 20, 0, 2, 5, 8, 1, 4, 7, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65],
    "smpl_to_openpose": {mt: [int(i) for i in U.smpl_to_openpose(mt)] for mt in ("smpl", "smplh", "smplx")},
    "smpl_skeleton": [[int(a), int(b)] for a, b in U.get_smpl_skeleton()],
}
# smpl_to_openpose for every argument combination (utils.py:68-177), as the blocks the function concatenates: per (format, model
# type) the body map, the two hand maps and the first face index -- read off the function's own outputs (body = no hands / no face;
# hands = what use_hands appends; face = an arange from the first appended index, 51 + 17 * use_face_contour long).
blocks = {}
for fmt in ("coco25", "coco19"):
    for mt in ("smpl", "smplh", "smplx"):
        body = U.smpl_to_openpose(mt, use_hands=False, use_face=False, openpose_format=fmt)
        hands = U.smpl_to_openpose(mt, use_hands=True, use_face=False, openpose_format=fmt)[len(body):]
        face = U.smpl_to_openpose(mt, use_hands=False, use_face=True, openpose_format=fmt)[len(body):]
        face_c = U.smpl_to_openpose(mt, use_hands=False, use_face=True, use_face_contour=True, openpose_format=fmt)[len(body):]
        assert len(hands) in (0, 42) and len(face) in (0, 51) and len(face_c) in (0, 68)
        if len(face):
            assert list(face) == list(range(int(face[0]), int(face[0]) + 51)) and list(face_c) == list(range(int(face[0]), int(face[0]) + 68))
        blocks[f"{fmt}/{mt}"] = {"body": [int(i) for i in body], "lhand": [int(i) for i in hands[:21]], "rhand": [int(i) for i in hands[21:]],
                                 "face_start": int(face[0]) if len(face) else -1}
out["smpl_to_openpose_blocks"] = blocks
for perm in ("H36M_TO_J17", "H36M_TO_J14", "J24_TO_J17", "J24_TO_J14", "SMPL_JOINTS_FLIP_PERM", "SMPL_POSE_FLIP_PERM",
             "J24_FLIP_PERM", "J49_FLIP_PERM"):
    out[perm] = [int(i) for i in getattr(C, perm)]
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dposer_amd", "body_model", "tables.json")
with open(path, "w") as f:
    json.dump(out, f, separators=(",", ":"))
print(path, os.path.getsize(path))
