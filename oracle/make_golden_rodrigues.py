"""Generate tests/golden/rodrigues_0.npz by running the reference's OWN axis-angle -> rotation-matrix functions
(survey container only).

/root/reference/src/Ev2Hands/losses.py:14-51 (`quat_to_rotmat`, `batch_rodrigues`) is the in-tree statement of the rotation
formula the MANO layer uses (manopth's rodrigues_layer has the same one).  losses.py imports mesh_intersection, trimesh, ... at
module level and cannot be imported here; the two functions only need torch, so their FunctionDef nodes are compiled straight from
the reference file (no source text is copied into this repository) and executed on seeded axis-angle vectors.  Before the fixture
is written the oracle's restatement (oracle/mano_oracle.py: rodrigues) is asserted bit-identical to the reference's output.
"""
from __future__ import annotations

import ast
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ev2hands_amd import synth  # noqa: E402
from oracle import mano_oracle  # noqa: E402

REF = "/root/reference/src/Ev2Hands/losses.py"


def load_functions():
    ns = {"torch": torch, "np": np}
    names = ["quat_to_rotmat", "batch_rodrigues"]
    tree = ast.parse(open(REF).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(body) == len(names), [n.name for n in body]
    exec(compile(ast.Module(body=body, type_ignores=[]), REF, "exec"), ns)
    return ns


def thetas() -> torch.Tensor:
    """Axis-angle vectors: hand-pose sized (|theta| <~ 2), exact zeros, tiny angles around the 1e-8 offset, angles near pi
    and 2 pi, and a few large ones."""
    rows = [synth.hash_normal("rod.a", (96, 3), 0) * 0.7,
            np.zeros((2, 3)),
            synth.hash_normal("rod.tiny", (12, 3), 1) * np.logspace(-9, -4, 12)[:, None],
            np.eye(3) * np.pi, -np.eye(3) * np.pi, np.eye(3) * 2 * np.pi,
            synth.hash_normal("rod.big", (8, 3), 2) * 6.0,
            np.array([[1e-8, -1e-8, 1e-8], [-1e-8, -1e-8, -1e-8], [1.0, 0.0, 0.0], [0.0, -2.5, 0.0]])]
    return torch.from_numpy(np.concatenate(rows, 0).astype(np.float32))


def main():
    ns = load_functions()
    th = thetas()
    with torch.no_grad():
        ref = ns["batch_rodrigues"](th.clone())
        mine = mano_oracle.rodrigues(th.clone())
        quat = torch.from_numpy(synth.hash_normal("rod.q", (32, 4), 3).astype(np.float32))
        ref_q = ns["quat_to_rotmat"](quat.clone())
    finite = torch.isfinite(ref).all(-1).all(-1)
    assert torch.equal(ref[finite], mine[finite]), float((ref[finite] - mine[finite]).abs().max())
    assert torch.equal(torch.isfinite(mine).all(-1).all(-1), finite)
    path = os.path.join(ROOT, "tests", "golden", "rodrigues_0.npz")
    np.savez_compressed(path, theta=th.numpy(), R=ref.numpy(), quat=quat.numpy(), R_quat=ref_q.numpy())
    print("wrote", path, os.path.getsize(path), "bytes;", int(finite.sum()), "of", len(th), "rows finite;",
          "orthogonality error", float((ref[finite] @ ref[finite].transpose(1, 2) - torch.eye(3)).abs().max()))


if __name__ == "__main__":
    main()
