"""Generation metrics (reference lib/utils/metric.py).

``average_pairwise_distance`` is the APD of run/demo.py's ``--metrics`` path.  The reference fills a [B, B] matrix with
a Python double loop (B^2/2 tiny kernels); here it is one batched distance computation on whatever device the joints
live on, chunked so the [chunk, B, J] intermediate stays bounded."""
import numpy as np
import torch


def average_pairwise_distance(joints3d, chunk=1024):
    """APD of ``joints3d [B, J, 3]`` (metric.py:8-37): mean over ordered pairs i != j of the mean per-joint Euclidean
    distance between poses i and j."""
    B = joints3d.shape[0]
    x = joints3d.float()
    total = torch.zeros((), dtype=torch.float64, device=x.device)
    for lo in range(0, B, chunk):
        d = torch.linalg.norm(x[lo:lo + chunk, None] - x[None], dim=-1).mean(dim=-1)      # [c, B]; the diagonal is exactly 0
        total += d.double().sum()
    return (total / (B * (B - 1))).float()


def self_intersections_percentage(vertices, faces):
    """Percentage of self-intersecting faces per mesh (metric.py:41-92).  The reference delegates to PyMeshLab and returns
    NaNs when it is not importable; PyMeshLab is not part of this image, so this always takes that branch."""
    try:
        import pymeshlab as pyml
    except ImportError:
        return np.ones(len(vertices)) * np.nan
    if isinstance(vertices, torch.Tensor):
        vertices = vertices.detach().cpu().numpy()
    if isinstance(faces, torch.Tensor):
        faces = faces.detach().cpu().numpy()
    out = np.zeros(len(vertices))
    for i, v in enumerate(vertices):
        ms = pyml.MeshSet()
        ms.add_mesh(pyml.Mesh(v, faces))
        n_all = ms.get_topological_measures()["faces_number"]
        ms.compute_selection_by_self_intersections_per_face()
        ms.meshing_remove_selected_faces()
        out[i] = (n_all - ms.get_topological_measures()["faces_number"]) / n_all * 100
    return out
