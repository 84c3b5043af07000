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
    """faces [nf,3] (ndarray or tensor) as a contiguous int32 tensor on `dev`.  A tensor that already is one is returned as it is
    (no copy, no synchronisation): callers on a hot path convert ONCE (device_faces) and pass the result."""
    if torch.is_tensor(f) and f.dtype == torch.int32 and f.device == torch.device(dev) and f.is_contiguous():
        return f
    return torch.as_tensor(np.asarray(f.cpu() if torch.is_tensor(f) else f).astype(np.int32)).to(dev).contiguous()


def device_faces(faces, dev) -> torch.Tensor:
    """One-time conversion of a hand model's `.faces` (ndarray [nf,3], or the [B,nf,3] tiling the eval forward returns) for the
    functions below; a pageable host->device copy synchronises the stream, so it does not belong into a per-step call."""
    a = np.asarray(faces.cpu() if torch.is_tensor(faces) else faces)
    return _faces(a[0] if a.ndim == 3 else a, dev)


def mesh_collisions(verts_left: torch.Tensor, verts_right: torch.Tensor, faces_left, faces_right, max_pairs: int = 0,
                    scale: float = 1000.0, max_per_triangle: int = 0, scratch: torch.Tensor | None = None):
    """verts_* [B,nv,3] float32 metres on the GPU, faces_* [nf,3] (ndarray or tensor).  Returns (counts [B] int32 tensor,
    pairs [B,max_pairs,2] int32 tensor or None; rows past counts[b] are unspecified).  max_per_triangle: the BVH's
    `max_collisions` cap (0 = none), see ev2h_mesh_collisions.  scratch: optional caller-owned uint8 buffer of at least
    `ev2h_mesh_collisions_scratch_bytes(B, nf)` bytes, used on the CURRENT stream only (the phases of one search communicate
    through it); without one a buffer cached per (device, stream) is used."""
    B, nv, _ = verts_left.shape
    dev = verts_left.device
    vl = verts_left.to(torch.float32).contiguous()
    vr = verts_right.to(dev, torch.float32).contiguous()
    fl, fr = _faces(faces_left, dev), _faces(faces_right, dev)
    nf = fl.shape[0]
    counts = torch.empty(B, device=dev, dtype=torch.int32)
    pairs = torch.empty(B, max_pairs, 2, device=dev, dtype=torch.int32) if max_pairs > 0 else None
    L = _lib.lib()
    # a scratch buffer lets the library split a window's row blocks over two workgroups when the batch alone cannot fill the chip
    if scratch is None and B <= 128:
        scratch = _scratch(L.ev2h_mesh_collisions_scratch_bytes(B, nf), dev)
    elif scratch is not None:
        if scratch.device != vl.device or scratch.dtype != torch.uint8 or not scratch.is_contiguous():
            raise ValueError("mesh_collisions: scratch must be a contiguous uint8 tensor on the vertices' device")
    _lib.check(L.ev2h_mesh_collisions_ws(vl.data_ptr(), vr.data_ptr(), fl.data_ptr(), fr.data_ptr(), B, nv, nf, float(scale),
                                         max_pairs, _lib.ptr(pairs), counts.data_ptr(), int(max_per_triangle), _lib.ptr(scratch),
                                         scratch.numel() if scratch is not None else 0, _lib.stream_handle()),
               "ev2h_mesh_collisions_ws")
    return counts, pairs


_SCRATCH = {}


def _scratch(nbytes: int, dev):
    """Scratch for ev2h_mesh_collisions_ws, one buffer per (device, STREAM), grown on demand.  The three phases of a search
    communicate through it, so two searches may share it only when they are ordered -- i.e. on one stream; searches on different
    streams (a user's own, or the library's main / side schedule) get different buffers.  A buffer that is outgrown goes back to
    the caching allocator on the stream that used it, which is the allocator's own ordering rule.  During a stream capture nothing
    may be allocated: a search that finds no (large enough) buffer then runs without one (one workgroup per window)."""
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream()
        key = (str(dev), int(st.cuda_stream))
        t = _SCRATCH.get(key)
        if t is None or t.numel() < nbytes:
            if torch.cuda.is_current_stream_capturing():
                return None
            t = _SCRATCH[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return t


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
    collision_weight = 100.  Forward value only (the path is inference); parity unpinned (un-vendored torch-mesh-isect).

    Deviation from upstream for B > 1 (documented): losses.py:88-93 builds the triangles with
    `verts_tensor.view([-1, 3])[face_tensor]` WITHOUT offsetting the face indices per batch item, so every item indexes item 0's
    vertices and the upstream value is 100 x the penalty of window 0 alone.  Here every window uses its own vertices (what the
    code evidently means); `reference_batch_quirk=True` reproduces the upstream value instead.

    max_pairs: capacity of the per-window pair list; default 2 * nf * max_collisions, the most a capped search can return, so the
    penalty is never computed over a silently truncated list.  With an explicit smaller capacity `truncated(counts)` tells."""

    def __init__(self, device=None, max_collisions: int = 16, sigma: float = 0.5, collision_weight: float = 1e2, max_pairs: int | None = None,
                 reference_batch_quirk: bool = False):
        self.max_collisions, self.sigma, self.collision_weight, self.max_pairs = max_collisions, sigma, collision_weight, max_pairs
        self.reference_batch_quirk = reference_batch_quirk
        self._faces_key, self._faces_dev = None, None
        self.last_counts = None

    def _device_faces(self, fl, fr, dev):
        """the two face tables on the device, converted once per (object, device)"""
        key = (id(fl), id(fr), str(dev))
        if self._faces_key != key:
            self._faces_dev = (device_faces(fl, dev), device_faces(fr, dev), fl, fr)       # (the originals are kept: ids stay unique)
            self._faces_key = key
        return self._faces_dev[0], self._faces_dev[1]

    def capacity(self, nf: int) -> int:
        if self.max_pairs is not None:
            return int(self.max_pairs)
        if self.max_collisions <= 0:
            raise ValueError("an uncapped pair search (max_collisions = 0) needs an explicit max_pairs")
        return 2 * nf * self.max_collisions

    def per_window(self, outs, faces=None) -> torch.Tensor:
        """[B] float64 penalties.  faces: optional (faces_left, faces_right) overriding outs[side]['faces'] (e.g. device_faces()
        tensors prepared once)."""
        vl, vr = outs["left"]["vertices"], outs["right"]["vertices"]
        dev = vl.device
        fl, fr = faces if faces is not None else (outs["left"]["faces"], outs["right"]["faces"])
        flt, frt = self._device_faces(fl, fr, dev)
        if self.reference_batch_quirk:
            vl, vr = vl[:1].expand_as(vl), vr[:1].expand_as(vr)                          # losses.py:88-93: every item reads item 0
        cap = self.capacity(flt.shape[0])
        counts, pairs = mesh_collisions(vl, vr, flt, frt, max_pairs=cap, scale=1.0, max_per_triangle=self.max_collisions)
        self.last_counts = counts
        B, nv, _ = vl.shape
        loss = torch.zeros(B, device=dev, dtype=torch.float64)
        vlc, vrc = vl.to(torch.float32).contiguous(), vr.to(dev, torch.float32).contiguous()
        _lib.check(_lib.lib().ev2h_collision_penalty(vlc.data_ptr(), vrc.data_ptr(),
                                                     flt.data_ptr(), frt.data_ptr(), B, nv, flt.shape[0], 1.0, float(self.sigma),
                                                     pairs.data_ptr(), counts.data_ptr(), cap, loss.data_ptr(),
                                                     _lib.stream_handle()), "ev2h_collision_penalty")
        return loss

    def truncated(self) -> bool:
        """True if the last per_window() found more pairs in some window than its pair list holds (host synchronisation)."""
        if self.last_counts is None:
            return False
        nf = self._faces_dev[0].shape[0]
        return bool(int(self.last_counts.max()) > self.capacity(nf))

    def __call__(self, outs, faces=None):
        loss = self.per_window(outs, faces)
        nz = loss[loss != 0]
        return (nz.mean() * self.collision_weight).to(torch.float32) if nz.numel() else 0
