"""ORACLE -- test infrastructure, not product code.

Restatement of the reference's per-frame joint metrics, the consumer right after the hot path:
  /root/reference/src/Ev2Hands/evaluate.py:185-234             absolute / relative / right-root-relative PCK curves
  /root/reference/src/Ev2Hands/evaluate_ev2hands_r.py:35-89    get_auc (trapezoid / n, rounded to 3), mepj_frame (root-relative
                                                               MPJPE), evaluate_joints_real (best-of-G ground-truth candidates)
PINNED: oracle/make_golden_metrics.py compiles those function definitions straight out of the reference files (the modules
themselves import trimesh / mesh_intersection / dv and cannot be imported) and asserts equality before writing the fixture.
"""
from __future__ import annotations

import numpy as np
import torch


def _pck(dists: torch.Tensor, num_steps: int, dist_max_mm: float) -> np.ndarray:
    pck = np.zeros(num_steps + 1)
    for s in range(num_steps + 1):
        pck[s] = (dists < (dist_max_mm / num_steps) * s).float().mean()
    return pck


def _dists(pred: torch.Tensor, gt: torch.Tensor) -> torch.Tensor:
    return torch.norm(torch.cat([pred[0], pred[1]], 0) - torch.cat([gt[0], gt[1]], 0), p=2, dim=1)


def absolute_pck(pred, gt, num_steps=100, dist_max_mm=100):          # evaluate.py:185-198
    return _pck(_dists(pred, gt), num_steps, dist_max_mm)


def relative_pck(pred, gt, num_steps=100, dist_max_mm=100):          # evaluate.py:201-217, each hand minus its own root
    return _pck(_dists(pred - pred[:, :1, :], gt - gt[:, :1, :]), num_steps, dist_max_mm)


def right_root_relative_pck(pred, gt, num_steps=100, dist_max_mm=100):   # evaluate.py:220-234, both hands minus the RIGHT root
    return _pck(_dists(pred - pred[1:, :1, :], gt - gt[1:, :1, :]), num_steps, dist_max_mm)


def auc(pck: np.ndarray, digits: int = 3):                           # evaluate_ev2hands_r.py:35-39 (sklearn.metrics.auc = trapezoid)
    n = pck.shape[0]
    return round(np.sum((pck[1:] + pck[:-1]) * 0.5) / n, digits)


def mpjpe(pred, gt):                                                  # evaluate_ev2hands_r.py:43-54
    return _dists(pred - pred[:, :1, :], gt - gt[:, :1, :]).mean()


def evaluate_joints(pred_mm: torch.Tensor, gts_mm: torch.Tensor, num_steps: int) -> dict:
    """evaluate_ev2hands_r.py:58-89.  pred_mm [2,21,3], gts_mm [G,2,21,3]; the candidate with the best (rounded)
    right-root-relative AUC is scored (first one on ties, np.argmax)."""
    aucs = [auc(right_root_relative_pck(pred_mm, g, num_steps)) for g in gts_mm]
    k = int(np.argmax(aucs))
    g = gts_mm[k]
    return {
        "best": k,
        "root_distance": [torch.norm(g[0] - g[1], p=2, dim=-1).min(-1)[0].cpu().numpy().tolist()],
        "joint_loss": mpjpe(pred_mm, g).item(),
        "absolute_pck3d": absolute_pck(pred_mm, g, num_steps),
        "relative_pck3d": relative_pck(pred_mm, g, num_steps),
        "right_root_relative_pck3d": right_root_relative_pck(pred_mm, g, num_steps),
    }
