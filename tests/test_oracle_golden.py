"""CPU: the oracle (oracle/tehnet_oracle.py) against the fixtures captured from the imported
reference by oracle/make_golden.py.  Selections must be identical; floats are bit-exact in the
container that made the fixtures and are allowed 2e-6 relative elsewhere (other host ISA / MKL path)."""
import glob
import os

import numpy as np
import pytest
import torch

from ev2hands_amd import synth
from oracle import mano_oracle, tehnet_oracle

CASES = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")) if not os.path.basename(p).startswith(("events_", "metrics_", "rodrigues_", "trained_weights", "trained2_weights")))


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_oracle_matches_reference_fixture(path):
    g = np.load(path)
    B, C, N, seed = [int(v) for v in g["meta"]]
    kind = str(g["kind"])
    n_pose = int(g["n_pose"]) if "n_pose" in g.files else synth.MANO_CMPS        # TEHNet(n_pose_params): head width 3 + n_pose + 13
    xyz = synth.synth_cloud(kind, B, C, N, seed)
    assert np.array_equal(xyz.numpy(), g["xyz"]), "synthetic input generator drifted"
    inits = synth.fps_inits(B, N, seed)
    assert np.array_equal(torch.stack(inits).numpy(), g["fps_init"])
    if "ckpt" in g.files:          # weights that came out of the reference's training loop (oracle/make_golden_trained.py)
        import trained_ckpt
        assert str(g["ckpt"]) == "trained"
        sd = trained_ckpt.trained_state_dict(C, trained_ckpt.run_of_fixture(path))
    else:
        sd = synth.synth_state_dict(C, seed, n_pose)
    hands = mano_oracle.make_hands(synth.synth_mano_assets("left", seed), synth.synth_mano_assets("right", seed), ncomps=n_pose)
    if "tie_eps" in g.files:       # near-tie segmentation head (oracle/stress.py)
        from oracle import stress
        sd = stress.near_tie_state_dict(sd, xyz, inits, hands, float(g["tie_eps"]))
    trace = {}
    mhlnes = bool(int(g["mhlnes"])) if "mhlnes" in g.files else False
    xin = xyz.clone()
    with torch.no_grad():
        out = tehnet_oracle.tehnet_forward(sd, xin, hands, fps_init=inits, trace=trace, mhlnes=mhlnes, n_pose=n_pose)
    if mhlnes:       # TEHNet.py:176-177 overwrites channel 2 of the caller's tensor; the fixture holds what the reference left behind
        assert np.array_equal(xin.numpy(), g["xyz_after"]) and not np.array_equal(xin.numpy(), g["xyz"])
    else:
        assert torch.equal(xin, xyz)
    # discrete selections: exact
    for k in g.files:
        if k.endswith(".fps") or ".group" in k or k.endswith(".nn_idx"):
            assert np.array_equal(trace[k].numpy(), g[k].astype(np.int64)), k
    if "tie_eps" in g.files:       # points inside the float tolerance band may flip on another host's MKL path
        lg = torch.from_numpy(g["class_logits"]).double()
        top = lg.topk(2, dim=1).values
        safe = (top[:, 0] - top[:, 1]) > 4e-6 * float(lg.abs().max())
        assert np.array_equal(out["class_logits"].argmax(1).numpy()[safe.numpy()], g["argmax"][safe.numpy()])
    else:
        assert np.array_equal(out["class_logits"].argmax(1).numpy(), g["argmax"])
    # floats
    assert rel(out["class_logits"].numpy(), g["class_logits"]) < 2e-6
    for side in ("left", "right"):
        assert rel(trace[side + ".params"].numpy(), g[side + ".params"]) < 2e-6
        assert rel(trace[side + ".hand_features"].numpy(), g[side + ".hand_features"]) < 2e-6
        assert rel(out[side]["vertices"].numpy(), g[f"unpinned.{side}.vertices"]) < 2e-6
        assert rel(out[side]["j3d"].numpy(), g[f"unpinned.{side}.j3d"]) < 2e-6
        assert out[side]["faces"].shape == (B, 1538, 3)
    assert rel(trace["fp1.nn_w"].numpy(), g["fp1.nn_w"]) < 1e-5
    assert rel(trace["l0_points"].reshape(-1)[::97].numpy(), g["fp1.sample"]) < 2e-6


def test_forward_draws_fps_inits_like_reference():
    """With fps_init=None the oracle consumes torch's global RNG in the reference's order."""
    B, C, N = 1, 4, 256
    sd = synth.synth_state_dict(C, 5)
    hands = mano_oracle.make_hands(synth.synth_mano_assets("left", 5), synth.synth_mano_assets("right", 5))
    xyz = synth.synth_cloud("U", B, C, N, 5)
    torch.manual_seed(77)
    inits = [torch.randint(0, hi, (B,), dtype=torch.long) for hi in (N, 512, N, N)]
    with torch.no_grad():
        a = tehnet_oracle.tehnet_forward(sd, xyz.clone(), hands, fps_init=inits)
        torch.manual_seed(77)
        b = tehnet_oracle.tehnet_forward(sd, xyz.clone(), hands)
    assert torch.equal(a["class_logits"], b["class_logits"])
    assert torch.equal(a["right"]["vertices"], b["right"]["vertices"])


def test_input_not_mutated_without_mhlnes():
    B, C, N = 1, 5, 256
    sd = synth.synth_state_dict(C, 1)
    hands = mano_oracle.make_hands(synth.synth_mano_assets("left", 1), synth.synth_mano_assets("right", 1))
    xyz = synth.synth_cloud("E", B, C, N, 1)
    keep = xyz.clone()
    with torch.no_grad():
        tehnet_oracle.tehnet_forward(sd, xyz, hands, fps_init=synth.fps_inits(B, N, 1))
    assert torch.equal(xyz, keep)
