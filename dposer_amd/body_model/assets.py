"""Model files -> the asset dictionary the HIP body model is built from.

The reference never parses a model file itself: ``BodyModel(bm_path=...)`` (lib/body_model/body_model.py:14-66) and the ``SMPLX``
wrapper (lib/body_model/smpl.py:49-52) hand the path to the un-vendored ``smplx==0.1.28`` (``smplx.body_models.{SMPL,SMPLH,SMPLX}``).
This module restates what that loader does with the file -- nothing of the arithmetic, which lives in ``csrc/fk.hip``:

* a path may be a file or a directory (``<dir>/SMPL_NEUTRAL.pkl``, ``SMPLH_NEUTRAL.pkl``, ``SMPLX_NEUTRAL.npz``: smplx's defaults,
  gender 'neutral');
* ``.npz`` (``np.load(allow_pickle=True)``) and ``.pkl`` (``pickle.load(encoding='latin1')``) hold the same keys: ``v_template [V,3]``,
  ``shapedirs [V,3,S]``, ``posedirs [V,3,9(J-1)]``, ``J_regressor [J,V]`` (dense or scipy-sparse), ``weights [V,J]``,
  ``kintree_table [2,J]`` (row 0 = parents, the root's entry is ``2**32 - 1`` in the uint32 files), ``f [F,3]``; SMPL-X adds
  ``lmk_faces_idx [51]``, ``lmk_bary_coords [51,3]``;
* shape space (``SMPL.__init__``): a file with fewer than 300 directions is a "10-beta" file and ``num_betas`` is clamped to 10,
  else to 300; ``shapedirs[:, :, :num_betas]``;
* SMPL-X expression space (``SMPLX.__init__``): a file with fewer than 300 + 100 directions (SMPL-X v1.0: 10 + 10) keeps its
  expression directions at ``[10:20]`` and ``num_expression_coeffs`` is clamped to 10; a v1.1 file (400) at ``[300:300 + n]``;
* SMPL-H through the reference (body_model.py:44-57): the shape space is zero-padded to 300 directions FIRST, so any ``num_betas``
  up to 300 is served (AMASS files carry 16);
* ``use_pca=False, flat_hand_mean=True`` (body_model.py:35-36): hand PCA components and means are not used, the pose mean is zero.

Legacy ``.pkl`` files (SMPL ``basicModel_*.pkl``, SMPL-H ``SMPLH_*.pkl``) store some arrays as ``chumpy`` objects.  chumpy is not
a dependency here: the unpickler stands in a shim for every ``chumpy.*`` class and reads the array it wraps (the ``x`` entry of its
state); a file whose chumpy objects have another shape raises and names smplx's ``tools/clean_ch.py``, which writes plain arrays.

The asset dictionary carries the clamped ``num_betas`` / ``num_expressions``: ``BodyModel`` builds with those (smplx prints a warning
and does the same).
"""
import os
import pickle

import numpy as np

from .synthetic import SMPLH_EXTRA_VERTEX_IDS, SMPLX_EXTRA_VERTEX_IDS

SHAPE_SPACE_DIM = 300            # smplx.SMPL.SHAPE_SPACE_DIM
EXPRESSION_SPACE_DIM = 100       # smplx.SMPLX.EXPRESSION_SPACE_DIM
NUM_JOINTS = {"smpl": 24, "smplh": 52, "smplx": 55}
_DEFAULT_FILE = {"smpl": "SMPL_NEUTRAL.pkl", "smplh": "SMPLH_NEUTRAL.pkl", "smplx": "SMPLX_NEUTRAL.npz"}


class ModelFileError(ValueError):
    """The file is not a model file this loader understands (the message says which key / shape is off)."""


class _ChumpyShim:
    """Stand-in for any ``chumpy`` class met while unpickling: keeps the pickled state, ``to_array()`` digs the wrapped ndarray out."""

    def __init__(self, *args, **kwargs):
        self._state = {}

    def __setstate__(self, state):
        self._state = state if isinstance(state, dict) else {"state": state}

    def to_array(self):
        st = self._state
        if "x" in st:
            x = st["x"]
            return x.to_array() if isinstance(x, _ChumpyShim) else np.asarray(x)
        raise ModelFileError("the file pickles a chumpy expression (not a plain chumpy array): run smplx's tools/clean_ch.py on it, "
                             "which writes the same keys as numpy arrays")


class _ModelUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == "chumpy" or module.startswith("chumpy."):
            return _ChumpyShim
        return super().find_class(module, name)


def to_np(a, dtype=None):
    """smplx.utils.to_np: dense ndarray from ndarray / scipy-sparse / chumpy."""
    if isinstance(a, _ChumpyShim):
        a = a.to_array()
    elif hasattr(a, "todense"):                 # scipy.sparse (J_regressor of the .pkl files)
        a = np.asarray(a.todense())
    elif type(a).__module__.split(".")[0] == "chumpy":      # chumpy installed after all
        a = np.array(a)
    a = np.asarray(a)
    return a if dtype is None else a.astype(dtype)


def resolve_model_path(path, model_type):
    """File as given, or smplx's default file name inside a directory."""
    if os.path.isdir(path):
        cand = os.path.join(path, _DEFAULT_FILE[model_type])
        if not os.path.exists(cand):                        # (smplx lets ``ext`` choose; accept the other extension too)
            other = os.path.splitext(cand)[0] + (".pkl" if cand.endswith(".npz") else ".npz")
            if os.path.exists(other):
                cand = other
        path = cand
    if not os.path.exists(path):
        raise FileNotFoundError(f"body model file {path!r} does not exist")
    return path


def read_model_file(path):
    """{key: value} of a ``.npz`` or ``.pkl`` model file (values still raw: sparse / chumpy-shim / ndarray)."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npz":
        with np.load(path, allow_pickle=True, encoding="latin1") as d:
            return {k: d[k] for k in d.files}
    if ext == ".pkl":
        with open(path, "rb") as f:
            d = _ModelUnpickler(f, encoding="latin1").load()
        if not isinstance(d, dict):
            raise ModelFileError(f"{path!r}: the pickle holds a {type(d).__name__}, expected a dict of arrays")
        return d
    raise ModelFileError(f"unknown model file extension {ext!r} (smplx reads .npz and .pkl)")


def _need(d, key, path):
    if key not in d:
        raise ModelFileError(f"{path!r} has no {key!r} (keys: {sorted(d)[:12]}...)")
    return d[key]


def asset_from_arrays(d, model_type="smplx", num_betas=10, num_expressions=10, path="<arrays>"):
    """The asset dictionary from the raw key -> array mapping of a model file (see the module docstring for every rule)."""
    if model_type not in NUM_JOINTS:
        raise ValueError(f"model_type must be one of {sorted(NUM_JOINTS)}, got {model_type!r}")
    v_template = to_np(_need(d, "v_template", path), np.float32)
    if v_template.ndim != 2 or v_template.shape[1] != 3:
        raise ModelFileError(f"{path!r}: v_template has shape {v_template.shape}, expected [V, 3]")
    V = v_template.shape[0]
    sd = to_np(_need(d, "shapedirs", path), np.float32)
    if sd.ndim < 3:
        sd = sd[:, :, None]
    if sd.shape[:2] != (V, 3):
        raise ModelFileError(f"{path!r}: shapedirs has shape {sd.shape}, expected [{V}, 3, S]")
    S = sd.shape[2]
    if model_type == "smplh" and S < SHAPE_SPACE_DIM:
        # body_model.py:53-56: "super hacky way to let smplh use 16-size beta" -- zero-pad to 300 before smplx looks at it
        sd = np.concatenate([sd, np.zeros((V, 3, SHAPE_SPACE_DIM - S), np.float32)], axis=2)
        S = SHAPE_SPACE_DIM
    nb = min(int(num_betas), 10 if S < SHAPE_SPACE_DIM else SHAPE_SPACE_DIM)
    parts = [sd[:, :, :nb]]
    ne = 0
    if model_type == "smplx":
        if S < SHAPE_SPACE_DIM + EXPRESSION_SPACE_DIM:
            ne = min(int(num_expressions), 10)
            e0 = 10
        else:
            ne = min(int(num_expressions), EXPRESSION_SPACE_DIM)
            e0 = SHAPE_SPACE_DIM
        if e0 + ne > S:
            raise ModelFileError(f"{path!r}: shapedirs has {S} directions, no room for {ne} expression directions at [{e0}:{e0 + ne}]")
        parts.append(sd[:, :, e0:e0 + ne])
    shapedirs = np.ascontiguousarray(np.concatenate(parts, axis=2))
    pd = to_np(_need(d, "posedirs", path), np.float32)
    posedirs = np.ascontiguousarray(np.reshape(pd, [-1, pd.shape[-1]]).T)           # [P, V*3] (smplx SMPL.__init__)
    J_regressor = to_np(_need(d, "J_regressor", path), np.float32)
    weights = to_np(_need(d, "weights", path), np.float32)
    kt = to_np(_need(d, "kintree_table", path))
    parents = kt[0].astype(np.int64)
    parents[0] = -1                                                                 # (the uint32 files store 2**32 - 1 for the root)
    J = NUM_JOINTS[model_type]
    if J_regressor.shape != (J, V) or weights.shape != (V, J) or parents.shape != (J,):
        raise ModelFileError(f"{path!r} is not a {model_type} file: J_regressor {J_regressor.shape}, weights {weights.shape}, "
                             f"kintree_table {kt.shape}; expected {J} joints over {V} vertices")
    if posedirs.shape != ((J - 1) * 9, V * 3):
        raise ModelFileError(f"{path!r}: posedirs has shape {pd.shape}, expected [{V}, 3, {(J - 1) * 9}]")
    if (parents[1:] < 0).any() or (parents[1:] >= np.arange(1, J)).any():
        raise ModelFileError(f"{path!r}: kintree_table is not a tree in topological order (parent index must precede the child)")
    faces = to_np(_need(d, "f", path)).astype(np.int64).astype(np.int32)
    if model_type == "smplx":
        lmk_faces_idx = to_np(_need(d, "lmk_faces_idx", path)).astype(np.int64).astype(np.int32).reshape(-1)
        lmk_bary = to_np(_need(d, "lmk_bary_coords", path), np.float32).reshape(-1, 3)
        extra = SMPLX_EXTRA_VERTEX_IDS.copy()
    else:
        lmk_faces_idx, lmk_bary = np.zeros((0,), np.int32), np.zeros((0, 3), np.float32)
        extra = SMPLH_EXTRA_VERTEX_IDS.copy()
    if extra.max() >= V:
        raise ModelFileError(f"{path!r}: {V} vertices, but the {model_type} extra-joint vertex ids reach {int(extra.max())} "
                             "(smplx/vertex_ids.py) -- not a template of this model family")
    return dict(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs, J_regressor=J_regressor, parents=parents, weights=weights,
                faces=faces, lmk_faces_idx=lmk_faces_idx, lmk_bary_coords=lmk_bary, extra_joint_vertex_ids=extra,
                num_betas=nb, num_expressions=ne, model_type=model_type)


def load_model_file(path, model_type="smplx", num_betas=10, num_expressions=10):
    """``BodyModel(bm_path=path)``'s loader: file or directory, ``.npz`` or ``.pkl``, SMPL / SMPL-H / SMPL-X."""
    path = resolve_model_path(os.fspath(path), model_type)
    return asset_from_arrays(read_model_file(path), model_type, num_betas, num_expressions, path=path)


def max_skinning_influences(asset):
    """Non-zero skinning weights of the busiest vertex (the ELL width the kernels run with; 4 selects the fast paths)."""
    return int((np.asarray(asset["weights"]) != 0).sum(axis=1).max())
