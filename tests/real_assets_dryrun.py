"""Dry run of the real-asset validation procedure (tools/validate_real_assets.py) without the licensed files: a synthetic
checkpoint and MANO-shaped assets are written in the REAL file formats and the CPU oracle stands in for the reference when the
fixture is recorded.  Test infrastructure (imports oracle/).   python tests/real_assets_dryrun.py ASSET_DIR OUT.npz"""
import os
import pickle
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import validate_real_assets as V  # noqa: E402
from ev2hands_amd import synth  # noqa: E402


def run_oracle(assets: dict, sd: dict):
    from oracle import mano_oracle, tehnet_oracle
    hands = mano_oracle.make_hands(assets["left"], assets["right"])

    def run(xyz):
        B, _, N = xyz.shape
        drawn = [torch.randint(0, hi, (B,), dtype=torch.long) for hi in (N, synth.SA1_NPOINT, N, N)]
        with torch.no_grad():
            out = tehnet_oracle.tehnet_forward(sd, xyz.clone(), hands, fps_init=drawn)
        return out, drawn
    return run


def write_synthetic_assets(out_dir: str, C: int = 5, seed: int = 7):
    """A synthetic checkpoint and MANO-shaped pkl files in the REAL file formats (dry run of the procedure)."""
    import scipy.sparse as sp
    os.makedirs(os.path.join(out_dir, "mano"), exist_ok=True)
    sd = synth.synth_state_dict(C, seed)
    ckpt = os.path.join(out_dir, "best_model_state_dict.pth")
    torch.save({"state_dict": {"module." + k: v for k, v in sd.items()}}, ckpt)
    assets = {}
    for side in ("left", "right"):
        a = synth.synth_mano_assets(side, seed)
        assets[side] = a
        d = {"v_template": a["v_template"], "shapedirs": a["shapedirs"], "posedirs": a["posedirs"],
             "J_regressor": sp.csc_matrix(a["J_regressor"]), "weights": a["weights"], "hands_components": a["hands_components"],
             "hands_mean": a["hands_mean"], "f": a["faces"].astype(np.uint32),
             "kintree_table": np.array([[4294967295] + a["parents"][1:], list(range(16))], dtype=np.int64)}
        with open(os.path.join(out_dir, "mano", f"MANO_{side.upper()}.pkl"), "wb") as f:
            pickle.dump(d, f, protocol=2)
    return ckpt, out_dir, assets



def make(asset_dir: str, out: str) -> int:
    ckpt, mano_dir, assets = write_synthetic_assets(asset_dir)
    sd = V.load_checkpoint(ckpt)
    return V.make_fixture(run_oracle(assets, sd), V.channels_of(sd), ckpt, mano_dir, out, "oracle (dry run, synthetic assets)")


if __name__ == "__main__":
    sys.exit(make(sys.argv[1], sys.argv[2]))
