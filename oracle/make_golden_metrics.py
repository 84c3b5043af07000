"""Generate tests/golden/metrics_0.npz by running the reference's OWN metric functions (survey container only).

evaluate.py and evaluate_ev2hands_r.py import trimesh, mesh_intersection, dv, ... at module level and cannot be imported here;
their metric functions only need torch / numpy / sklearn, so the FunctionDef nodes are compiled straight from the reference
files (no source text is copied into this repository) and executed on seeded synthetic joints.
"""
from __future__ import annotations

import ast
import os
import sys

import numpy as np
import torch
from sklearn import metrics as skmetrics

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ev2hands_amd import synth  # noqa: E402
from oracle import metrics_oracle as MO  # noqa: E402

REF = "/root/reference/src/Ev2Hands"


def load_functions():
    ns = {"torch": torch, "np": np, "skmetrics": skmetrics, "metrics": skmetrics}
    want = {"evaluate.py": ["absolute_pck3d_frame", "relative_pck3d_frame", "right_root_relative_pck3d_frame"],
            "evaluate_ev2hands_r.py": ["get_auc", "mepj_frame", "evaluate_joints_real"]}
    for fn, names in want.items():
        tree = ast.parse(open(os.path.join(REF, fn)).read())
        body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
        assert len(body) == len(names), (fn, [n.name for n in body])
        exec(compile(ast.Module(body=body, type_ignores=[]), os.path.join(REF, fn), "exec"), ns)
    return ns


def synth_case(B, G, seed):
    """pred [B,2,21,3] float32 metres; gts [B,G,2,21,3] float64 metres (the dataset hands double precision joints)."""
    gt = synth.hash_normal("gt", (B, G, 2, 21, 3), seed) * 0.05
    gt[:, :, 1, :, 0] += 0.15
    err = synth.hash_normal("err", (B, 2, 21, 3), seed) * np.array([0.004, 0.01, 0.03])[(np.arange(B) % 3)][:, None, None, None]
    pred = gt[np.arange(B), (np.arange(B) * 7) % G] + err
    return torch.from_numpy(pred.astype(np.float32)), torch.from_numpy(gt)


def main():
    ns = load_functions()
    B, G = 12, 3
    out = {}
    for num_steps in (100, 20):
        pred, gts = synth_case(B, G, num_steps)
        rows = []
        for b in range(B):
            ref = ns["evaluate_joints_real"](pred[b] * 1000, gts[b] * 1000, num_steps)
            mine = MO.evaluate_joints(pred[b] * 1000, gts[b] * 1000, num_steps)
            for k in ("absolute_pck3d", "relative_pck3d", "right_root_relative_pck3d"):
                assert np.array_equal(ref[k], mine[k]), k
                assert ns["get_auc"](ref[k]) == MO.auc(mine[k]), k
            assert ref["joint_loss"] == mine["joint_loss"] and ref["root_distance"] == mine["root_distance"]
            rows.append(ref)
        tag = f"s{num_steps}"
        out[tag + ".pred"] = pred.numpy()
        out[tag + ".gts"] = gts.numpy()
        out[tag + ".abs"] = np.stack([r["absolute_pck3d"] for r in rows])
        out[tag + ".rel"] = np.stack([r["relative_pck3d"] for r in rows])
        out[tag + ".rrr"] = np.stack([r["right_root_relative_pck3d"] for r in rows])
        out[tag + ".mpjpe"] = np.array([r["joint_loss"] for r in rows])
        out[tag + ".rootd"] = np.array([r["root_distance"][0] for r in rows])
        out[tag + ".auc"] = np.array([[ns["get_auc"](r[k]) for k in ("absolute_pck3d", "relative_pck3d", "right_root_relative_pck3d")] for r in rows])
        out[tag + ".best"] = np.array([MO.evaluate_joints(pred[b] * 1000, gts[b] * 1000, num_steps)["best"] for b in range(B)])
        print(tag, "best candidates", out[tag + ".best"], "mpjpe mm", out[tag + ".mpjpe"].round(2)[:4], "auc", out[tag + ".auc"][0])
    path = os.path.join(ROOT, "tests", "golden", "metrics_0.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
