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


def _faces(f, dev):
    return torch.as_tensor(np.asarray(f.cpu() if torch.is_tensor(f) else f).astype(np.int32)).to(dev).contiguous()


def mesh_collisions(verts_left: torch.Tensor, verts_right: torch.Tensor, faces_left, faces_right, max_pairs: int = 0,
                    scale: float = 1000.0, max_per_triangle: int = 0):
    """verts_* [B,nv,3] float32 metres on the GPU, faces_* [nf,3] (ndarray or tensor).  Returns (counts [B] int32 tensor,
    pairs [B,max_pairs,2] int32 tensor or None; rows past counts[b] are unspecified).  max_per_triangle: the BVH's
    `max_collisions` cap (0 = none), see ev2h_mesh_collisions."""
    B, nv, _ = verts_left.shape
    dev = verts_left.device
    vl = verts_left.to(torch.float32).contiguous()
    vr = verts_right.to(dev, torch.float32).contiguous()
    fl, fr = _faces(faces_left, dev), _faces(faces_right, dev)
    nf = fl.shape[0]
    counts = torch.empty(B, device=dev, dtype=torch.int32)
    pairs = torch.empty(B, max_pairs, 2, device=dev, dtype=torch.int32) if max_pairs > 0 else None
    _lib.check(_lib.lib().ev2h_mesh_collisions(vl.data_ptr(), vr.data_ptr(), fl.data_ptr(), fr.data_ptr(), B, nv, nf, float(scale),
                                               max_pairs, _lib.ptr(pairs), counts.data_ptr(), int(max_per_triangle), _lib.stream_handle()),
               "ev2h_mesh_collisions")
    return counts, pairs


def compute_non_collision_score(verts_left_pred, faces_left, verts_right_pred, faces_right, max_collisions: int = 8):
    """Same call and first return value as the reference's compute_non_collision_score (a list of B floats); the second
    (the trimesh objects the reference builds for visualisation) is not produced -- None.  max_collisions = the reference BVH's
    per-triangle cap (evaluate_ev2hands_r.py:131); 0 counts every pair."""
    counts, _ = mesh_collisions(verts_left_pred, verts_right_pred, faces_left, faces_right, max_per_triangle=max_collisions)
    n_tri = 2 * np.asarray(faces_left.cpu() if torch.is_tensor(faces_left) else faces_left).shape[0]
    return [100 - round(int(c) / n_tri * 100, 2) for c in counts.cpu().numpy()], None


class CollisionLoss:
    """Value of the reference's intersection-aware loss term (/root/reference/src/Ev2Hands/losses.py:60-102) for a batch of
    predictions: colliding triangle pairs of the concatenated two-hand mesh in METRES (max_collisions = 16 per triangle), their
    conic distance-field penetration penalty (sigma = 0.5), mean over the windows with a non-zero penalty, times
    collision_weight = 100.  Forward value only (the path is inference); parity unpinned (un-vendored torch-mesh-isect)."""

    def __init__(self, device=None, max_collisions: int = 16, sigma: float = 0.5, collision_weight: float = 1e2, max_pairs: int = 8192):
        self.max_collisions, self.sigma, self.collision_weight, self.max_pairs = max_collisions, sigma, collision_weight, max_pairs

    def per_window(self, outs) -> torch.Tensor:
        vl, vr = outs["left"]["vertices"], outs["right"]["vertices"]
        fl, fr = outs["left"]["faces"], outs["right"]["faces"]
        fl, fr = (f[0] if np.asarray(f).ndim == 3 else f for f in (fl, fr))         # eval outputs tile the faces per window
        counts, pairs = mesh_collisions(vl, vr, fl, fr, max_pairs=self.max_pairs, scale=1.0, max_per_triangle=self.max_collisions)
        B, nv, _ = vl.shape
        dev = vl.device
        loss = torch.zeros(B, device=dev, dtype=torch.float64)
        flt, frt = _faces(fl, dev), _faces(fr, dev)
        vlc, vrc = vl.to(torch.float32).contiguous(), vr.to(dev, torch.float32).contiguous()
        _lib.check(_lib.lib().ev2h_collision_penalty(vlc.data_ptr(), vrc.data_ptr(),
                                                     flt.data_ptr(), frt.data_ptr(), B, nv, flt.shape[0], 1.0, float(self.sigma),
                                                     pairs.data_ptr(), counts.data_ptr(), self.max_pairs, loss.data_ptr(),
                                                     _lib.stream_handle()), "ev2h_collision_penalty")
        return loss

    def __call__(self, outs):
        loss = self.per_window(outs)
        nz = loss[loss != 0]
        return (nz.mean() * self.collision_weight).to(torch.float32) if nz.numel() else 0
