"""Host side of the on-device joint metrics.

`evaluate_joints_real_batch` scores a whole batch of frames in one kernel instead of the reference's per-frame
`.cpu()` loop (/root/reference/src/Ev2Hands/evaluate_ev2hands_r.py:91-125) and returns, per frame, the same dictionary
as the reference's evaluate_joints_real (:58-89); `get_auc` rounding (:35-39) is applied here.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib


def evaluate_joints_real_batch(j3d_left: torch.Tensor, j3d_right: torch.Tensor, j3d_gts: torch.Tensor, num_steps: int,
                               dist_max_mm: float = 100.0):
    """j3d_left / j3d_right [B,21,3] float32 metres on the GPU (outputs['left'|'right']['j3d']); j3d_gts [B,G,2,21,3] metres
    (any float dtype; compared in float64 like the reference).  Returns a list of B dicts."""
    B = j3d_left.shape[0]
    G = j3d_gts.shape[1]
    dev = j3d_left.device
    l = j3d_left.to(torch.float32).contiguous()
    r = j3d_right.to(torch.float32).contiguous()
    g = j3d_gts.to(dev, torch.float64).contiguous()
    n = num_steps + 1
    pck = torch.empty(B, 3, n, device=dev, dtype=torch.float32)
    auc = torch.empty(B, 3, device=dev, dtype=torch.float64)
    mp = torch.empty(B, device=dev, dtype=torch.float64)
    rd = torch.empty(B, device=dev, dtype=torch.float64)
    best = torch.empty(B, device=dev, dtype=torch.int32)
    L = _lib.lib()
    _lib.check(L.ev2h_joint_metrics(l.data_ptr(), r.data_ptr(), g.data_ptr(), B, G, num_steps, float(dist_max_mm), pck.data_ptr(),
                                    auc.data_ptr(), mp.data_ptr(), rd.data_ptr(), best.data_ptr(), _lib.stream_handle()),
               "ev2h_joint_metrics")
    pck_h, auc_h, mp_h, rd_h, best_h = pck.cpu().numpy().astype(np.float64), auc.cpu().numpy(), mp.cpu().numpy(), rd.cpu().numpy(), best.cpu().numpy()
    out = []
    for b in range(B):
        out.append({"root_distance": [float(rd_h[b])], "joint_loss": float(mp_h[b]),
                    "absolute_pck3d": pck_h[b, 0], "relative_pck3d": pck_h[b, 1], "right_root_relative_pck3d": pck_h[b, 2],
                    "absolute_auc": round(auc_h[b, 0], 3), "relative_auc": round(auc_h[b, 1], 3),
                    "right_root_relative_auc": round(auc_h[b, 2], 3), "gt_index": int(best_h[b])})
    return out
