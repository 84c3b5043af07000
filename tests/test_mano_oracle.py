"""CPU: known-answer tests that anchor the (unpinned) MANO restatement -- SURVEY.md 8c (i)-(v)."""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from ev2hands_amd import synth
from oracle import mano_oracle


def assets(side="right", seed=0, zero_mean=False):
    a = synth.synth_mano_assets(side, seed)
    if zero_mean:
        a = dict(a)
        a["hands_mean"] = np.zeros(45)
    return a


def test_rodrigues_vs_scipy():
    th = torch.from_numpy((synth.hash_uniform("rv", (200, 3), 0) * 2 - 1) * 2.5).float()
    R = mano_oracle.rodrigues(th).numpy()
    Rs = Rotation.from_rotvec(th.numpy().astype(np.float64)).as_matrix()
    assert np.abs(R - Rs).max() < 1e-6
    R0 = mano_oracle.rodrigues(torch.zeros(1, 3)).numpy()
    assert np.abs(R0 - np.eye(3)).max() < 1e-6


def test_rodrigues_is_pinned_to_the_reference_formula():
    """tests/golden/rodrigues_0.npz holds outputs of the reference's own batch_rodrigues / quat_to_rotmat
    (/root/reference/src/Ev2Hands/losses.py:14-51, compiled from the reference file by oracle/make_golden_rodrigues.py): the
    oracle's rotation formula must reproduce them bit for bit, including the row where theta + 1e-8 vanishes (NaN)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rodrigues_0.npz"))
    th, R = torch.from_numpy(g["theta"]), torch.from_numpy(g["R"])
    mine = mano_oracle.rodrigues(th)
    fin = torch.isfinite(R).all(-1).all(-1)
    assert int((~fin).sum()) == 1 and torch.equal(torch.isfinite(mine).all(-1).all(-1), fin)
    assert torch.equal(mine[fin], R[fin])


@pytest.mark.parametrize("side", ["left", "right"])
def test_rest_pose_is_template_plus_shape(side):
    a = assets(side, 3, zero_mean=True)
    m = mano_oracle.ManoOracle(a, dtype=torch.float64)
    B = 3
    betas = torch.from_numpy(synth.hash_normal("b", (B, 10), 1))
    tr = torch.from_numpy(synth.hash_normal("t", (B, 3), 1) * 0.1)
    out = m(torch.zeros(B, 3, dtype=torch.float64), torch.zeros(B, 6, dtype=torch.float64), betas, tr)
    v_shaped = a["v_template"][None] + np.einsum("vck,bk->bvc", a["shapedirs"], betas.numpy())
    assert np.abs(out.vertices.numpy() - (v_shaped + tr.numpy()[:, None])).max() < 1e-7
    J = np.einsum("jv,bvc->bjc", a["J_regressor"], v_shaped) + tr.numpy()[:, None]
    # joints 0..15 of the chain land at reorder positions; tips come from vertices
    order = synth.MANO_JOINT_REORDER
    got = out.joints.numpy()
    for pos, src in enumerate(order):
        if src < 16:
            assert np.abs(got[:, pos] - J[:, src]).max() < 1e-7
        else:
            tip = synth.MANO_TIPS[side][src - 16]
            assert np.abs(got[:, pos] - out.vertices.numpy()[:, tip]).max() < 1e-12


def test_global_rotation_is_rigid_about_root():
    a = assets("right", 4)
    m = mano_oracle.ManoOracle(a, dtype=torch.float64)
    B = 2
    pose = torch.from_numpy(synth.hash_normal("p", (B, 6), 2))
    betas = torch.from_numpy(synth.hash_normal("b", (B, 10), 2))
    z3 = torch.zeros(B, 3, dtype=torch.float64)
    base = m(z3, pose, betas, z3)
    rv = torch.from_numpy(synth.hash_normal("r", (B, 3), 2))
    rot = m(rv, pose, betas, z3)
    R = Rotation.from_rotvec(rv.numpy()).as_matrix()
    j0 = base.joints.numpy()[:, 0:1]
    exp_v = np.einsum("bij,bvj->bvi", R, base.vertices.numpy() - j0) + j0
    assert np.abs(rot.vertices.numpy() - exp_v).max() < 1e-6    # 1e-8 inside the norm perturbs the angle
    exp_j = np.einsum("bij,bvj->bvi", R, base.joints.numpy() - j0) + j0
    assert np.abs(rot.joints.numpy() - exp_j).max() < 1e-6


def _mano_numpy64(a, side, go, hp, betas, tr):
    """Independent fp64 re-derivation: per-joint recursion over the parent table instead of levels."""
    B = go.shape[0]
    full = np.concatenate([go, a["hands_mean"][None] + hp @ a["hands_components"][:6]], 1).reshape(B, 16, 3)
    R = Rotation.from_rotvec((full + 1e-8).reshape(-1, 3)).as_matrix().reshape(B, 16, 3, 3)
    v_shaped = a["v_template"][None] + np.einsum("vck,bk->bvc", a["shapedirs"], betas)
    J = np.einsum("jv,bvc->bjc", a["J_regressor"], v_shaped)
    pm = (R[:, 1:] - np.eye(3)).reshape(B, 135)
    v_posed = v_shaped + np.einsum("vck,bk->bvc", a["posedirs"], pm)
    G = np.zeros((B, 16, 4, 4))
    for k in range(16):
        L = np.zeros((B, 4, 4))
        L[:, :3, :3] = R[:, k]
        L[:, 3, 3] = 1
        p = a["parents"][k]
        if p < 0:
            L[:, :3, 3] = J[:, k]
            G[:, k] = L
        else:
            L[:, :3, 3] = J[:, k] - J[:, p]
            G[:, k] = G[:, p] @ L
    A = G.copy()
    A[:, :, :3, 3] -= np.einsum("bkij,bkj->bki", G[:, :, :3, :3], J)
    T = np.einsum("vk,bkij->bvij", a["weights"], A)
    vh = np.concatenate([v_posed, np.ones((B, 778, 1))], 2)
    verts = np.einsum("bvij,bvj->bvi", T, vh)[:, :, :3]
    jt = np.concatenate([G[:, :, :3, 3], verts[:, synth.MANO_TIPS[side]]], 1)[:, synth.MANO_JOINT_REORDER]
    return verts + tr[:, None], jt + tr[:, None]


@pytest.mark.parametrize("side", ["left", "right"])
def test_fp32_layer_vs_independent_fp64(side):
    a = assets(side, 5)
    B = 4
    prm = synth.hash_normal("prm", (B, 22), 3) * 0.5
    go, hp, be, tr = prm[:, :3], prm[:, 3:9], prm[:, 9:19], prm[:, 19:] * 0.2
    v64, j64 = _mano_numpy64(a, side, go, hp, be, tr)
    m32 = mano_oracle.ManoOracle(a, dtype=torch.float32)
    t = lambda x: torch.from_numpy(x).float()
    out = m32(t(go), t(hp), t(be), t(tr))
    assert np.abs(out.vertices.numpy() - v64).max() < 1e-5       # metres
    assert np.abs(out.joints.numpy() - j64).max() < 1e-5
    assert out.vertices.shape == (B, 778, 3) and out.joints.shape == (B, 21, 3)


def test_tables_literal():
    assert synth.MANO_TIPS["right"] == [745, 317, 444, 556, 673]
    assert synth.MANO_TIPS["left"] == [745, 317, 445, 556, 673]
    assert synth.MANO_JOINT_REORDER == [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]
    assert mano_oracle.CHAIN_REORDER == [0, 1, 6, 11, 2, 7, 12, 3, 8, 13, 4, 9, 14, 5, 10, 15]
    # the oracle keeps its own literal copies (it does not import the product's): the two must agree
    assert mano_oracle.MANO_TIPS == synth.MANO_TIPS and mano_oracle.MANO_JOINT_REORDER == synth.MANO_JOINT_REORDER


def test_left_shapedirs_fix():
    r = synth.synth_mano_assets("right", 0)
    l = dict(synth.synth_mano_assets("left", 0))
    l["shapedirs"] = r["shapedirs"].copy()          # the MANO release bug: identical first components
    hands = mano_oracle.make_hands(l, r)
    assert torch.allclose(hands["left"].shapedirs[:, 0, :], -hands["right"].shapedirs[:, 0, :])
    assert torch.allclose(hands["left"].shapedirs[:, 1:, :], hands["right"].shapedirs[:, 1:, :])
