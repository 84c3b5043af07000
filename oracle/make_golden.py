"""Generate tests/golden/*.npz from the REAL reference (survey container only).

Imports /root/reference/src/Ev2Hands/model/{pointnet2_utils,TEHNet}.py as a synthetic package
(the package __init__ drags in trimesh/pyrender/manopth, which are absent -- SURVEY.md
Appendix A), loads a synthetic checkpoint with the reference's own `load_state_dict(strict=True)`,
runs its forward on seeded synthetic clouds and records selections + outputs.  The MANO call is
served by oracle/mano_oracle.py (manopth and the MANO assets are not available; that part is
UNPINNED and stored under keys starting with `unpinned.`).

It also asserts that oracle/tehnet_oracle.py reproduces the reference bit-for-bit here, so a
fixture is only ever written from a state where restatement == reference.

Run:  python oracle/make_golden.py            (needs /root/reference; never runs on the GPU box)
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ev2hands_amd import synth  # noqa: E402
from oracle import mano_oracle, stress, tehnet_oracle  # noqa: E402

REF_MODEL_DIR = "/root/reference/src/Ev2Hands/model"
CASES = [
    # name,            kind, C, N,    B, seed
    ("U_c4_n2048", "U", 4, 2048, 2, 0),
    ("E_c5_n2048", "E", 5, 2048, 2, 1),
    ("U_c5_n256", "U", 5, 256, 2, 2),
    ("E_c4_n256", "E", 4, 256, 2, 3),
    ("E_c4_n8192", "E", 4, 8192, 1, 4),      # BASELINE.json config 5 window size
    ("E_c5_n256_mhlnes", "E", 5, 256, 2, 6),  # MHLNES=1 (TEHNet.py:148,176-177): channel 2 overwritten in place (name suffix)
    ("E_c4_n2048_ties", "E", 4, 2048, 2, 31),  # near-tie segmentation head (oracle/stress.py: near_tie_state_dict, eps 0.3)
    ("E_c4_n256_pose12", "E", 4, 256, 2, 8),   # TEHNet(n_pose_params=12) (TEHNet.py:114-125; name suffix _pose<K>): head width 28
    ("U_c5_n256_pose45", "U", 5, 256, 2, 9),   # all 45 MANO pose components: head width 61
]
TIE_EPS = 0.3


def load_reference():
    pkg = types.ModuleType("refmodel")
    pkg.__path__ = [REF_MODEL_DIR]
    sys.modules["refmodel"] = pkg

    def load(name):
        spec = importlib.util.spec_from_file_location(f"refmodel.{name}", f"{REF_MODEL_DIR}/{name}.py")
        m = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = m
        spec.loader.exec_module(m)
        return m

    return load("pointnet2_utils"), load("TEHNet")


def sample(t: torch.Tensor, stride: int = 97) -> np.ndarray:
    return t.detach().reshape(-1)[::stride].numpy().copy()


def stats(t: torch.Tensor) -> np.ndarray:
    t = t.detach().double()
    return np.array([t.mean().item(), t.abs().max().item(), t.pow(2).mean().sqrt().item()])


def run_case(pn, te, name, kind, C, N, B, seed, sd_override=None, extra=None):
    """`sd_override`: a checkpoint other than synth_state_dict(C, seed) (oracle/make_golden_trained.py: weights that came out of
    the reference's own training loop); `extra`: additional arrays to store with the fixture."""
    os.environ["ERPC"] = "1" if C == 5 else "0"
    mhlnes = name.endswith("_mhlnes")
    os.environ["MHLNES"] = "1" if mhlnes else "0"
    n_pose = int(name.rsplit("_pose", 1)[1]) if "_pose" in name else synth.MANO_CMPS
    sd = synth.synth_state_dict(C, seed, n_pose) if sd_override is None else sd_override
    net = te.TEHNet(n_pose_params=n_pose)
    assert net.mhlnes == int(mhlnes)
    net.load_state_dict(sd, strict=True)
    net.eval()
    hands = mano_oracle.make_hands(synth.synth_mano_assets("left", seed), synth.synth_mano_assets("right", seed), ncomps=n_pose)
    xyz = synth.synth_cloud(kind, B, C, N, seed)
    inits = synth.fps_inits(B, N, seed)
    if name.endswith("_ties"):
        sd = stress.near_tie_state_dict(sd, xyz, inits, hands, TIE_EPS)
        net.load_state_dict(sd, strict=True)

    # spy on the reference's selections
    rec = {"fps": [], "ball": [], "nn": []}
    o_fps, o_ball, o_sq = pn.farthest_point_sample, pn.query_ball_point, pn.square_distance
    it = iter(inits)
    o_randint = torch.randint

    def spy_randint(*a, **k):
        v = next(it)
        assert a[1] in (N, 512) and tuple(a[2]) == (B,)
        return v.clone()

    def spy_fps(x, n):
        r = o_fps(x, n)
        rec["fps"].append(r.clone())
        return r

    def spy_ball(r_, k_, x, nx):
        r = o_ball(r_, k_, x, nx)
        rec["ball"].append(r.clone())
        return r

    pn.farthest_point_sample, pn.query_ball_point = spy_fps, spy_ball
    feats = {}
    hooks = []
    for mod_name in ("sa1", "sa2", "sa3", "fp3", "fp2", "fp1", "classifier", "left_query_conv", "right_query_conv"):
        def mk(nm):
            def hook(_m, _i, o):
                feats[nm] = o
            return hook
        hooks.append(getattr(net, mod_name).register_forward_hook(mk(mod_name)))
    for side in ("left", "right"):
        def mkp(sd_):
            def hook(_m, i, o):
                feats[sd_ + ".params"] = o
            return hook
        hooks.append(getattr(net, side + "_mano_regressor").mano_regressor.register_forward_hook(mkp(side)))
        def mkf(sd_):
            def hook(_m, i):
                feats[sd_ + ".hand_features"] = i[1]
            return hook
        hooks.append(getattr(net, side + "_mano_regressor").register_forward_pre_hook(mkf(side)))

    torch.randint = spy_randint
    xyz_ref_in = xyz.clone()                 # MHLNES=1 mutates its input (TEHNet.py:176-177): keep what the reference left behind
    try:
        with torch.no_grad():
            ref = net(xyz_ref_in, hands)
    finally:
        torch.randint = o_randint
        pn.farthest_point_sample, pn.query_ball_point = o_fps, o_ball
        for h in hooks:
            h.remove()

    # our restatement must be identical here
    trace = {}
    xyz_mine_in = xyz.clone()
    with torch.no_grad():
        mine = tehnet_oracle.tehnet_forward(sd, xyz_mine_in, hands, fps_init=inits, trace=trace, mhlnes=mhlnes, n_pose=n_pose)
    assert torch.equal(xyz_mine_in, xyz_ref_in), "oracle leaves a different input behind than the reference"
    assert torch.equal(xyz_ref_in, xyz) != mhlnes, "MHLNES must (only) mutate the input when set"
    assert torch.equal(mine["class_logits"], ref["class_logits"]), "oracle != reference (logits)"
    for side in ("left", "right"):
        for k in ("vertices", "j3d", "global_orient", "hand_pose", "betas", "transl"):
            assert torch.equal(mine[side][k], ref[side][k]), f"oracle != reference ({side}.{k})"
        assert np.array_equal(mine[side]["faces"], ref[side]["faces"])
    fps_names = ["sa1.fps", "sa2.fps", "left_mano_regressor.sa1.fps", "right_mano_regressor.sa1.fps"]
    for nm, r in zip(fps_names, rec["fps"]):
        assert torch.equal(trace[nm], r), nm
    ball_names = ([f"sa1.group{i}" for i in range(3)] + [f"sa2.group{i}" for i in range(2)]
                  + [f"left_mano_regressor.sa1.group{i}" for i in range(2)]
                  + [f"right_mano_regressor.sa1.group{i}" for i in range(2)])
    for nm, r in zip(ball_names, rec["ball"]):
        assert torch.equal(trace[nm], r), nm
    assert torch.equal(trace["l0_points"], feats["fp1"])
    assert torch.equal(trace["left.query"], feats["left_query_conv"])
    assert torch.equal(trace["left.params"], feats["left.params"])

    out = {
        "meta": np.array([B, C, N, seed], dtype=np.int64),
        "kind": np.array(kind),
        "xyz": xyz.numpy(),
        "mhlnes": np.array(int(mhlnes)),
        "n_pose": np.array(n_pose),
        "xyz_after": xyz_ref_in.numpy() if mhlnes else np.zeros(0, dtype=np.float32),
        "fps_init": torch.stack(inits).numpy().astype(np.int32),
        "class_logits": ref["class_logits"].numpy(),
        "argmax": ref["class_logits"].argmax(1).numpy().astype(np.uint8),
    }
    for nm, r in zip(fps_names, rec["fps"]):
        out[nm] = r.numpy().astype(np.int16)
    for nm, r in zip(ball_names, rec["ball"]):
        out[nm] = r.numpy().astype(np.int16)
    for nm in ("fp2", "fp1"):
        out[nm + ".nn_idx"] = trace[nm + ".nn_idx"].numpy().astype(np.int16)
        out[nm + ".nn_w"] = trace[nm + ".nn_w"].numpy()
    for nm in ("sa1", "sa2", "sa3"):
        out[nm + ".feat.stats"] = stats(feats[nm][1])
        out[nm + ".feat.sample"] = sample(feats[nm][1])
    out["sa1.new_xyz"] = feats["sa1"][0].numpy()
    out["sa2.new_xyz"] = feats["sa2"][0].numpy()
    for nm in ("fp3", "fp2", "fp1", "left_query_conv", "right_query_conv"):
        out[nm + ".stats"] = stats(feats[nm])
        out[nm + ".sample"] = sample(feats[nm])
    for side in ("left", "right"):
        out[side + ".hand_features"] = feats[side + ".hand_features"].numpy()
        out[side + ".params"] = feats[side + ".params"].numpy()
        out[f"unpinned.{side}.vertices"] = ref[side]["vertices"].numpy()
        out[f"unpinned.{side}.j3d"] = ref[side]["j3d"].numpy()
    if name.endswith("_ties"):
        m, scale = stress.margin_report(ref["class_logits"])
        out["tie_eps"] = np.array(TIE_EPS)
        print(f"{name}: logit scale {scale:.3g}; top-2 margin < 1e-5 scale at {float((m < 1e-5 * scale).float().mean()) * 100:.2f} % of the points, "
              f"< 1e-6 scale at {float((m < 1e-6 * scale).float().mean()) * 100:.2f} %, exact ties {int((m == 0).sum())}")
    if extra:
        out.update(extra)
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


def main():
    torch.set_num_threads(8)
    pn, te = load_reference()
    only = sys.argv[1:]                      # optional: names of the cases to (re)generate
    for case in CASES:
        if not only or case[0] in only:
            run_case(pn, te, *case)
    os.environ["MHLNES"] = "0"


if __name__ == "__main__":
    main()
