"""Host side of the two-hand mesh self-collision score.

`compute_non_collision_score` scores a whole batch in one kernel instead of the reference's per-frame trimesh + BVH loop
(/root/reference/src/Ev2Hands/evaluate_ev2hands_r.py:128-160; same formula in utils/__init__.py:106-124):
    score = 100 - round(n_collisions / n_triangles * 100, 2)
n_collisions is the number of intersecting, non-adjacent triangle pairs of the concatenated left+right mesh.  The
reference takes it from the un-vendored torch-mesh-isect BVH (max_collisions=8 candidates per triangle); this is the
uncapped count (parity unpinned).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


def mesh_collisions(verts_left: torch.Tensor, verts_right: torch.Tensor, faces_left, faces_right, max_pairs: int = 0,
                    scale: float = 1000.0):
    """verts_* [B,nv,3] float32 metres on the GPU, faces_* [nf,3] (ndarray or tensor).  Returns (counts [B] int32 tensor,
    pairs [B,max_pairs,2] int32 tensor or None; rows past counts[b] are unspecified)."""
    B, nv, _ = verts_left.shape
    dev = verts_left.device
    vl = verts_left.to(torch.float32).contiguous()
    vr = verts_right.to(dev, torch.float32).contiguous()
    fl = torch.as_tensor(np.asarray(faces_left.cpu() if torch.is_tensor(faces_left) else faces_left).astype(np.int32)).to(dev).contiguous()
    fr = torch.as_tensor(np.asarray(faces_right.cpu() if torch.is_tensor(faces_right) else faces_right).astype(np.int32)).to(dev).contiguous()
    nf = fl.shape[0]
    counts = torch.empty(B, device=dev, dtype=torch.int32)
    pairs = torch.empty(B, max_pairs, 2, device=dev, dtype=torch.int32) if max_pairs > 0 else None
    _lib.check(_lib.lib().ev2h_mesh_collisions(vl.data_ptr(), vr.data_ptr(), fl.data_ptr(), fr.data_ptr(), B, nv, nf, float(scale),
                                               max_pairs, _lib.ptr(pairs), counts.data_ptr(), _lib.stream_handle()),
               "ev2h_mesh_collisions")
    return counts, pairs


def compute_non_collision_score(verts_left_pred, faces_left, verts_right_pred, faces_right):
    """Same call and first return value as the reference's compute_non_collision_score (a list of B floats); the second
    (the trimesh objects the reference builds for visualisation) is not produced -- None."""
    counts, _ = mesh_collisions(verts_left_pred, verts_right_pred, faces_left, faces_right)
    n_tri = 2 * np.asarray(faces_left.cpu() if torch.is_tensor(faces_left) else faces_left).shape[0]
    return [100 - round(int(c) / n_tri * 100, 2) for c in counts.cpu().numpy()], None
