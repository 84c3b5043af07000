"""ORACLE -- test infrastructure, not product code.  PARITY UNPINNED.

CPU restatement of the MANO layer the reference calls through its `SmplxAdapter`
(/root/reference/src/Ev2Hands/model/utils.py:13-42).  The arithmetic lives in the third-party
package `manopth` (pinned only as `manopth==0.0.1`, /root/reference/ev2hands.yml:116; upstream
hassony2/manopth, `manopth/manolayer.py` ManoLayer.forward with use_pca=True, ncomps=6,
flat_hand_mean=False, root_rot_mode='axisang'), which is not vendored in the reference tree and
not installed here, and the licensed MANO_{LEFT,RIGHT}.pkl assets are absent.  This file restates
manopth's published algorithm; nothing in the reference pins its outputs, so tests anchor it
with known-answer checks instead (tests/test_mano_oracle.py): rest pose, rigid global rotation,
scipy Rotation.from_rotvec, an fp64 re-derivation, literal tip / reorder tables.

The axis-angle -> matrix formula is the one the reference itself states in
/root/reference/src/Ev2Hands/losses.py:14-51 (quaternion route, `theta + 1e-8` inside the norm).
"""
from __future__ import annotations

import numpy as np
import torch

# The oracle's OWN literal tables (not imported from the product package, so that the checker stays independent of what it
# checks; tests/test_mano_oracle.py::test_tables_literal compares the two):
#  * fingertip vertices appended to the 16 regressed joints -- manopth/manolayer.py ManoLayer.forward, `side == 'right'`:
#    tips = th_verts[:, [745, 317, 444, 556, 673]], left: [745, 317, 445, 556, 673];
#  * the reorder of the 21 joints to the "standard" hand-joint order, same file:
#    th_jtr = th_jtr[:, [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]];
#  * the kinematic chain: three levels below the root (lev1 = joints 1,4,7,10,13 ...) and the reorder [0, 1, 6, 11, ...] that
#    puts the level-major results back into joint order.
MANO_TIPS = {"right": [745, 317, 444, 556, 673], "left": [745, 317, 445, 556, 673]}
MANO_JOINT_REORDER = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]

LEVELS = ([1, 4, 7, 10, 13], [2, 5, 8, 11, 14], [3, 6, 9, 12, 15])
CHAIN_REORDER = [0, 1, 6, 11, 2, 7, 12, 3, 8, 13, 4, 9, 14, 5, 10, 15]


def rodrigues(theta: torch.Tensor) -> torch.Tensor:
    """losses.py:14-51 (same formula as manopth.rodrigues_layer.batch_rodrigues).  [M,3] -> [M,3,3]."""
    ang = torch.norm(theta + 1e-8, p=2, dim=1, keepdim=True)
    axis = theta / ang
    half = ang * 0.5
    q = torch.cat([torch.cos(half), torch.sin(half) * axis], dim=1)
    q = q / q.norm(p=2, dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    w2, x2, y2, z2 = w * w, x * x, y * y, z * z
    wx, wy, wz, xy, xz, yz = w * x, w * y, w * z, x * y, x * z, y * z
    R = torch.stack([w2 + x2 - y2 - z2, 2 * xy - 2 * wz, 2 * wy + 2 * xz,
                     2 * wz + 2 * xy, w2 - x2 + y2 - z2, 2 * yz - 2 * wx,
                     2 * xz - 2 * wy, 2 * wx + 2 * yz, w2 - x2 - y2 + z2], dim=1)
    return R.view(-1, 3, 3)


def _with_zeros(m34: torch.Tensor) -> torch.Tensor:
    pad = m34.new_zeros(m34.shape[:-2] + (1, 4))
    pad[..., 0, 3] = 1.0
    return torch.cat([m34, pad], dim=-2)


class ManoOracle:
    """One hand.  `assets` as produced by ev2hands_amd.synth.synth_mano_assets or the pkl loader.
    __call__(global_orient, hand_pose, betas, transl) -> object with .vertices [B,778,3] and
    .joints [B,21,3] in metres, like the reference adapter (utils.py:25-31)."""

    class Output:
        def __init__(self, vertices, joints):
            self.vertices = vertices
            self.joints = joints

    def __init__(self, assets: dict, ncomps: int = 6, dtype=torch.float32):
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype)
        self.side = assets["side"]
        self.dtype = dtype
        self.v_template = t(assets["v_template"]).unsqueeze(0)
        self.shapedirs = t(assets["shapedirs"])
        self.posedirs = t(assets["posedirs"])
        self.J_regressor = t(assets["J_regressor"])
        self.weights = t(assets["weights"])
        self.hands_mean = t(assets["hands_mean"]).unsqueeze(0)
        self.selected_comps = t(assets["hands_components"])[:ncomps]
        self.faces = np.asarray(assets["faces"]).astype(np.int64)
        self.ncomps = ncomps

    def __call__(self, global_orient, hand_pose, betas, transl):
        v, j = self.forward_mm(torch.cat([global_orient, hand_pose], 1), betas, transl)
        v = v / 1000      # utils.py:28-29
        j = j / 1000
        return ManoOracle.Output(v, j)

    def forward_mm(self, pose_coeffs, betas, trans):
        """manopth ManoLayer.forward: returns (verts, joints) in millimetres."""
        dt = self.dtype
        pose_coeffs, betas, trans = pose_coeffs.to(dt), betas.to(dt), trans.to(dt)
        B = pose_coeffs.shape[0]
        full_hand = pose_coeffs[:, 3:3 + self.ncomps].mm(self.selected_comps)
        full_pose = torch.cat([pose_coeffs[:, :3], self.hands_mean + full_hand], 1)           # [B,48]
        rots = rodrigues(full_pose.contiguous().view(-1, 3)).view(B, 16, 3, 3)
        root_rot = rots[:, 0]
        eye = torch.eye(3, dtype=dt).view(1, 1, 3, 3)
        pose_map = (rots[:, 1:] - eye).reshape(B, 135)
        v_shaped = torch.matmul(self.shapedirs, betas.transpose(1, 0)).permute(2, 0, 1) + self.v_template
        J = torch.matmul(self.J_regressor, v_shaped)                                          # [B,16,3]
        v_posed = v_shaped + torch.matmul(self.posedirs, pose_map.transpose(0, 1)).permute(2, 0, 1)

        root_j = J[:, 0, :].contiguous().view(B, 3, 1)
        root_T = _with_zeros(torch.cat([root_rot, root_j], 2))                                # [B,4,4]
        chain = [root_T.unsqueeze(1)]
        prev_T = root_T.unsqueeze(1).repeat(1, 5, 1, 1)
        prev_j = root_j.transpose(1, 2)                                                       # [B,1,3]
        for lev in LEVELS:
            r = rots[:, lev]
            jl = J[:, lev]
            rel = _with_zeros(torch.cat([r, (jl - prev_j).unsqueeze(3)], 3))                  # [B,5,4,4]
            cur = torch.matmul(prev_T, rel)
            chain.append(cur)
            prev_T, prev_j = cur, jl
        G = torch.cat(chain, 1)[:, CHAIN_REORDER]                                             # [B,16,4,4]

        Jh = torch.cat([J, J.new_zeros(B, 16, 1)], 2)
        t = torch.matmul(G, Jh.unsqueeze(3))                                                  # [B,16,4,1]
        A = (G - torch.cat([t.new_zeros(B, 16, 4, 3), t], 3)).permute(0, 2, 3, 1)             # [B,4,4,16]
        T = torch.matmul(A, self.weights.transpose(0, 1))                                     # [B,4,4,778]
        rest_h = torch.cat([v_posed.transpose(2, 1), torch.ones((B, 1, v_posed.shape[1]), dtype=dt)], 1)
        verts = (T * rest_h.unsqueeze(1)).sum(2).transpose(2, 1)[:, :, :3]
        jtr = G[:, :, :3, 3]
        tips = verts[:, MANO_TIPS[self.side]]
        jtr = torch.cat([jtr, tips], 1)[:, MANO_JOINT_REORDER]
        jtr = jtr + trans.unsqueeze(1)
        verts = verts + trans.unsqueeze(1)
        return verts * 1000, jtr * 1000


def make_hands(assets_left: dict, assets_right: dict, ncomps: int = 6, dtype=torch.float32) -> dict:
    """utils.py:33-40: both hands, with the left-hand shapedirs sign fix applied when the two
    assets carry (almost) identical first shapedirs components."""
    hands = {"left": ManoOracle(assets_left, ncomps, dtype), "right": ManoOracle(assets_right, ncomps, dtype)}
    if torch.sum(torch.abs(hands["left"].shapedirs[:, 0, :] - hands["right"].shapedirs[:, 0, :])) < 1:
        hands["left"].shapedirs[:, 0, :] *= -1
    return hands
