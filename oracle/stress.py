"""ORACLE-SIDE TEST INFRASTRUCTURE -- stress inputs for the discrete outputs (segmentation argmax, selections).

Imported by tests/ and by oracle/make_golden.py only.
"""
from __future__ import annotations

from collections import OrderedDict

import torch
import torch.nn.functional as F

from oracle import tehnet_oracle


def near_tie_state_dict(sd: dict, xyz: torch.Tensor, inits, hands, eps: float = 1e-3) -> "OrderedDict":
    """Checkpoint whose segmentation head (classifier.4, /root/reference/src/Ev2Hands/model/TEHNet.py:135-141) produces
    near-ties on the given clouds: the four rows become one common row plus eps times the original rows, and the biases are
    chosen so that the eps-sized class gaps have zero mean over the points.  The four logits of a point then differ by O(eps) of
    their magnitude with gaps of either sign, so the top-2 margins are spread densely down to zero and a percent-level fraction
    of the points sits within a few ulps of the arithmetic -- the stress case for `class_logits.argmax(1)` parity.
    Everything but classifier.4 is unchanged (so the selections and l0 features are those of `sd`)."""
    trace = {}
    with torch.no_grad():
        tehnet_oracle.tehnet_forward(sd, xyz.clone(), hands, fps_init=inits, trace=trace)
        h = F.relu(F.conv1d(trace["l0_points"], sd["classifier.0.weight"], sd["classifier.0.bias"]))
        h = tehnet_oracle._bn(h, sd, "classifier.2")
    center = h.double().mean(dim=(0, 2))                                   # [256]
    w, b = sd["classifier.4.weight"].double(), sd["classifier.4.bias"].double()
    out = OrderedDict((k, v.clone()) for k, v in sd.items())
    out["classifier.4.weight"] = (w[:1] + eps * w).float()
    out["classifier.4.bias"] = (b[:1] - eps * (w[:, :, 0] @ center)).float()
    return out


def margin_report(logits: torch.Tensor):
    """(top-2 margin per point [B,N] float64, logit scale = max |logit|)."""
    lg = logits.double()
    top = lg.topk(2, dim=1).values
    return top[:, 0] - top[:, 1], float(lg.abs().max())
