"""ORACLE -- test infrastructure, not product code.

CPU restatement (PyTorch-CPU ops, fp32) of the reference's per-frame inference hot path,
written as pure functions over a checkpoint `state_dict`.  Only tests/, bench.py's
`cpu_baseline` leg and __graft_entry__.smoke() may import this module; the product path
(ev2hands_amd/) never does and fails loudly when the HIP library is missing.

Parity status: PINNED for the encoder/regressor part -- tests/test_oracle_golden.py checks it
against fixtures captured from the imported reference (oracle/make_golden.py; bit-exact in the
container that made them).  The MANO layer is in oracle/mano_oracle.py and is UNPINNED.

Each function cites the reference lines it restates (paths relative to
/root/reference/src/Ev2Hands/model/).  The algebraic forms (operation order, matmul-form
distances, sort-based selections) are kept because the discrete selections downstream
amplify ulps (SURVEY.md section 7 "hard parts").
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

BN_EPS = 1e-5


# --------------------------------------------------------------------------- point ops
def pairwise_sqdist(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """pointnet2_utils.py:19-40.  [B,N,3],[B,M,3] -> [B,N,M] as -2*src.dst^T + |src|^2 + |dst|^2,
    added in that order (can be slightly negative for coincident points)."""
    B, N, _ = src.shape
    M = dst.shape[1]
    d = -2 * torch.matmul(src, dst.transpose(1, 2))
    d += (src ** 2).sum(-1).view(B, N, 1)
    d += (dst ** 2).sum(-1).view(B, 1, M)
    return d


def gather_points(points: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """pointnet2_utils.py:43-60.  points [B,N,D], idx [B,...] -> [B,...,D]."""
    B = points.shape[0]
    bsel = torch.arange(B, dtype=torch.long).view([B] + [1] * (idx.dim() - 1)).expand_as(idx)
    return points[bsel, idx, :]


def farthest_point_sample(xyz: torch.Tensor, npoint: int, init: torch.Tensor | None = None) -> torch.Tensor:
    """pointnet2_utils.py:63-84.  xyz [B,N,3] -> int64 [B,npoint].  `init` is the start index per
    cloud; when None it is drawn from the global CPU RNG exactly like the reference (:75)."""
    B, N, _ = xyz.shape
    out = torch.zeros(B, npoint, dtype=torch.long)
    mind = torch.full((B, N), 1e10)
    far = torch.randint(0, N, (B,), dtype=torch.long) if init is None else init.clone().long()
    rows = torch.arange(B, dtype=torch.long)
    for i in range(npoint):
        out[:, i] = far
        c = xyz[rows, far, :].view(B, 1, 3)
        d = ((xyz - c) ** 2).sum(-1)
        closer = d < mind
        mind[closer] = d[closer]
        far = mind.max(-1)[1]
    return out


def ball_query(radius: float, nsample: int, xyz: torch.Tensor, centers: torch.Tensor) -> torch.Tensor:
    """pointnet2_utils.py:87-107.  First `nsample` in-radius indices in ascending index order,
    padded with the first one.  Out-of-radius test is `d > radius**2` (python double squared,
    then compared in fp32)."""
    B, N, _ = xyz.shape
    S = centers.shape[1]
    idx = torch.arange(N, dtype=torch.long).view(1, 1, N).repeat(B, S, 1)
    d = pairwise_sqdist(centers, xyz)
    idx[d > radius ** 2] = N
    idx = idx.sort(dim=-1)[0][:, :, :nsample]
    first = idx[:, :, :1].expand(-1, -1, nsample)
    pad = idx == N
    idx[pad] = first[pad]
    return idx


def three_nn_weights(xyz1: torch.Tensor, xyz2: torch.Tensor):
    """pointnet2_utils.py:296-302.  Returns (idx [B,N,3] int64, weight [B,N,3])."""
    d = pairwise_sqdist(xyz1, xyz2)
    d, idx = d.sort(dim=-1)
    d, idx = d[:, :, :3], idx[:, :, :3]
    recip = 1.0 / (d + 1e-8)
    w = recip / recip.sum(dim=2, keepdim=True)
    return idx, w


# --------------------------------------------------------------------------- layers
def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, BN_EPS)


def _conv_bn_relu_2d(x, sd, pc, pb):
    return F.relu(_bn(F.conv2d(x, sd[pc + ".weight"], sd[pc + ".bias"]), sd, pb))


def _conv_bn_relu_1d(x, sd, pc, pb):
    return F.relu(_bn(F.conv1d(x, sd[pc + ".weight"], sd[pc + ".bias"]), sd, pb))


def sa_msg(sd, prefix, xyz_cm, feat_cm, npoint, radii, nsamples, init=None, trace=None):
    """pointnet2_utils.py:224-262 (multi-scale set abstraction).  xyz_cm [B,3,N], feat_cm [B,D,N]
    -> (new_xyz [B,3,S], new_feat [B,sum(D'),S]).  Channel order per group is [features, rel-xyz]."""
    xyz = xyz_cm.permute(0, 2, 1).contiguous()
    feat = feat_cm.permute(0, 2, 1).contiguous()
    B, N, _ = xyz.shape
    fps = farthest_point_sample(xyz, npoint, init)
    ctr = gather_points(xyz, fps)
    outs = []
    if trace is not None:
        trace[prefix + ".fps"] = fps
    for i, (r, K) in enumerate(zip(radii, nsamples)):
        gi = ball_query(r, K, xyz, ctr)
        if trace is not None:
            trace[f"{prefix}.group{i}"] = gi
        gx = gather_points(xyz, gi)
        gx -= ctr.view(B, npoint, 1, 3)
        g = torch.cat([gather_points(feat, gi), gx], dim=-1)
        g = g.permute(0, 3, 2, 1).contiguous()            # [B, D, K, S]
        j = 0
        while f"{prefix}.conv_blocks.{i}.{j}.weight" in sd:
            g = _conv_bn_relu_2d(g, sd, f"{prefix}.conv_blocks.{i}.{j}", f"{prefix}.bn_blocks.{i}.{j}")
            j += 1
        outs.append(g.max(2)[0])
    return ctr.permute(0, 2, 1).contiguous(), torch.cat(outs, dim=1)


def sa_group_all(sd, prefix, xyz_cm, feat_cm):
    """pointnet2_utils.py:141-158,176-202 (group_all).  Channel order is [x,y,z, features], the
    xyz are NOT centred; the new xyz is all-zero."""
    xyz = xyz_cm.permute(0, 2, 1).contiguous()
    feat = feat_cm.permute(0, 2, 1).contiguous()
    B, N, _ = xyz.shape
    g = torch.cat([xyz.view(B, 1, N, 3), feat.view(B, 1, N, -1)], dim=-1)
    g = g.permute(0, 3, 2, 1).contiguous()                # [B, 3+D, N, 1]
    k = 0
    while f"{prefix}.mlp_convs.{k}.weight" in sd:
        g = _conv_bn_relu_2d(g, sd, f"{prefix}.mlp_convs.{k}", f"{prefix}.mlp_bns.{k}")
        k += 1
    return torch.zeros(B, 3, 1), g.max(2)[0]


def feature_propagation(sd, prefix, xyz1_cm, xyz2_cm, feat1_cm, feat2_cm, trace=None):
    """pointnet2_utils.py:276-315.  3-NN inverse-distance interpolation of feat2 onto xyz1
    (broadcast when xyz2 has one point), concat [skip, interpolated], Conv1d-BN-ReLU stack."""
    xyz1 = xyz1_cm.permute(0, 2, 1).contiguous()
    xyz2 = xyz2_cm.permute(0, 2, 1).contiguous()
    f2 = feat2_cm.permute(0, 2, 1).contiguous()
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    if S == 1:
        interp = f2.repeat(1, N, 1)
    else:
        idx, w = three_nn_weights(xyz1, xyz2)
        if trace is not None:
            trace[prefix + ".nn_idx"] = idx
            trace[prefix + ".nn_w"] = w
        interp = (gather_points(f2, idx) * w.view(B, N, 3, 1)).sum(dim=2)
    if feat1_cm is not None:
        x = torch.cat([feat1_cm.permute(0, 2, 1).contiguous(), interp], dim=-1)
    else:
        x = interp
    x = x.permute(0, 2, 1).contiguous()
    k = 0
    while f"{prefix}.mlp_convs.{k}.weight" in sd:
        x = _conv_bn_relu_1d(x, sd, f"{prefix}.mlp_convs.{k}", f"{prefix}.mlp_bns.{k}")
        k += 1
    return x


def attention(key, value, query):
    """TEHNet.py:13-27.  key [B,4,N], value [B,256,N], query [B,256,N] -> [B,4,N].
    scale = (value channels)**-0.5, softmax over dim=1 (the 4 classes)."""
    q = query.permute(0, 2, 1)
    sim = torch.bmm(key, q)
    sim = (value.shape[1] ** -.5) * sim
    sim = F.softmax(sim, dim=1)
    return torch.bmm(sim, value)


def classifier(sd, x):
    """TEHNet.py:135-141: Conv1d -> ReLU -> BN -> (Dropout) -> Conv1d."""
    h = F.relu(F.conv1d(x, sd["classifier.0.weight"], sd["classifier.0.bias"]))
    h = _bn(h, sd, "classifier.2")
    return F.conv1d(h, sd["classifier.4.weight"], sd["classifier.4.bias"])


def query_conv(sd, side, x):
    """TEHNet.py:150-166: Conv1d(k3,p1) -> ReLU -> BN -> (Dropout) -> Conv1d(k3,p1) -> BN,
    both convolutions run along the point-index axis."""
    p = f"{side}_query_conv"
    h = F.relu(F.conv1d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], padding=1))
    h = _bn(h, sd, p + ".2")
    h = F.conv1d(h, sd[p + ".4.weight"], sd[p + ".4.bias"], padding=1)
    return _bn(h, sd, p + ".5")


def mano_params(sd, side, xyz_cm, feat_cm, init=None, trace=None):
    """TEHNet.py:68-90: MSG set abstraction on all N points, group-all, FC head -> [B,22]."""
    p = f"{side}_mano_regressor"
    l1_xyz, l1 = sa_msg(sd, p + ".sa1", xyz_cm, feat_cm, 128, [0.4, 0.8], [64, 128], init, trace)      # TEHNet.py:43
    _, l2 = sa_group_all(sd, p + ".sa2", l1_xyz, l1)
    h = l2.squeeze(-1)
    h = F.relu(F.linear(h, sd[p + ".mano_regressor.0.weight"], sd[p + ".mano_regressor.0.bias"]))
    h = _bn(h, sd, p + ".mano_regressor.2")
    return F.linear(h, sd[p + ".mano_regressor.4.weight"], sd[p + ".mano_regressor.4.bias"])


def tehnet_forward(sd, xyz_in, mano_hands, fps_init=None, n_pose=6, training=False, mhlnes=False, trace=None):
    """TEHNet.py:168-197.  xyz_in [B,C,N] float32 -> {'class_logits', 'left', 'right'}.
    `fps_init`: list of four [B] int64 start vectors in consumption order (enc.sa1, enc.sa2,
    left.sa1, right.sa1); None draws them from the global RNG like the reference.
    `mano_hands[side](global_orient=, hand_pose=, betas=, transl=)` -> obj(.vertices, .joints)."""
    import numpy as np
    fi = fps_init if fps_init is not None else [None] * 4
    feat0 = xyz_in
    xyz0 = xyz_in[:, :3, :]
    if mhlnes:
        xyz0[:, -1, :] = xyz_in[:, 3:, :].mean(1)        # in place, like TEHNet.py:176-177
    l1_xyz, l1 = sa_msg(sd, "sa1", xyz0, feat0, 512, [0.1, 0.2, 0.4], [32, 64, 128], fi[0], trace)      # TEHNet.py:127
    l2_xyz, l2 = sa_msg(sd, "sa2", l1_xyz, l1, 128, [0.4, 0.8], [64, 128], fi[1], trace)               # TEHNet.py:128
    l3_xyz, l3 = sa_group_all(sd, "sa3", l2_xyz, l2)
    if trace is not None:
        trace.update({"sa1_points": l1, "sa2_points": l2})
    l2 = feature_propagation(sd, "fp3", l2_xyz, l3_xyz, l2, l3, trace)
    l1 = feature_propagation(sd, "fp2", l1_xyz, l2_xyz, l1, l2, trace)
    l0 = feature_propagation(sd, "fp1", xyz0, l1_xyz, None, l1, trace)
    seg = classifier(sd, l0)
    out = {"class_logits": seg}
    if trace is not None:
        trace.update({"l1_xyz": l1_xyz, "l1_points": l1, "l2_xyz": l2_xyz, "l2_points": l2, "l3_points": l3,
                      "l0_points": l0})
    for k, side in enumerate(("left", "right")):
        q = query_conv(sd, side, l0)
        hf = attention(seg, l0, q)
        prm = mano_params(sd, side, xyz0, hf, fi[2 + k], trace)
        if trace is not None:
            trace[side + ".query"] = q
            trace[side + ".hand_features"] = hf
            trace[side + ".params"] = prm
        args = {"global_orient": prm[:, :3], "hand_pose": prm[:, 3:3 + n_pose],
                "betas": prm[:, 3 + n_pose:-3], "transl": prm[:, -3:]}
        res = mano_hands[side](**args)
        d = {"vertices": res.vertices, "j3d": res.joints}
        d.update(args)
        if not training:
            d["faces"] = np.tile(mano_hands[side].faces, (xyz_in.shape[0], 1, 1))
        out[side] = d
    return out


# --------------------------------------------------------------------------- the same function in float64
def tehnet_forward_f64(sd, xyz_in, mano_hands64, selections, n_pose=6):
    """The network FUNCTION of tehnet_forward evaluated in float64 on the SAME discrete selections: `selections` is the trace of a
    float32 tehnet_forward of the same input (FPS indices, ball-query groups, 3-NN indices and weights -- the reference computes
    those from float32 coordinates, and they define which function is evaluated), every feature contraction, BatchNorm, softmax and
    the MANO layer (`mano_hands64` = mano_oracle.make_hands(..., dtype=torch.float64)) run in float64.  Not a parity oracle: a
    yardstick -- how far the reference's OWN float32 sums are from the exact value of its function, next to which the arithmetic
    modes of the library are read (tests/trained_truth_report.py).  Relative coordinates are formed in float32 first, as the
    reference forms them (pointnet2_utils.py:245), then widened."""
    t = selections
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}

    def msg(prefix, xyz_cm, feat_cm, npoint, nb):
        xyz = xyz_cm.permute(0, 2, 1).contiguous()
        feat = feat_cm.permute(0, 2, 1).contiguous().double()
        B = xyz.shape[0]
        ctr = gather_points(xyz, t[prefix + ".fps"])
        outs = []
        for i in range(nb):
            gi = t[f"{prefix}.group{i}"]
            gx = gather_points(xyz, gi)
            gx = gx - ctr.view(B, npoint, 1, 3)
            g = torch.cat([gather_points(feat, gi), gx.double()], dim=-1).permute(0, 3, 2, 1).contiguous()
            j = 0
            while f"{prefix}.conv_blocks.{i}.{j}.weight" in sd:
                g = _conv_bn_relu_2d(g, sd, f"{prefix}.conv_blocks.{i}.{j}", f"{prefix}.bn_blocks.{i}.{j}")
                j += 1
            outs.append(g.max(2)[0])
        return ctr.permute(0, 2, 1).contiguous(), torch.cat(outs, dim=1)

    def group_all(prefix, xyz_cm, feat_cm):
        xyz = xyz_cm.permute(0, 2, 1).contiguous().double()
        feat = feat_cm.permute(0, 2, 1).contiguous()
        B, N, _ = xyz.shape
        g = torch.cat([xyz.view(B, 1, N, 3), feat.view(B, 1, N, -1)], dim=-1).permute(0, 3, 2, 1).contiguous()
        k = 0
        while f"{prefix}.mlp_convs.{k}.weight" in sd:
            g = _conv_bn_relu_2d(g, sd, f"{prefix}.mlp_convs.{k}", f"{prefix}.mlp_bns.{k}")
            k += 1
        return g.max(2)[0]

    def fp(prefix, N, S, feat1_cm, feat2_cm):
        f2 = feat2_cm.permute(0, 2, 1).contiguous()
        B = f2.shape[0]
        if S == 1:
            interp = f2.repeat(1, N, 1)
        else:
            interp = (gather_points(f2, t[prefix + ".nn_idx"]) * t[prefix + ".nn_w"].double().view(B, N, 3, 1)).sum(dim=2)
        x = torch.cat([feat1_cm.permute(0, 2, 1).contiguous(), interp], dim=-1) if feat1_cm is not None else interp
        x = x.permute(0, 2, 1).contiguous()
        k = 0
        while f"{prefix}.mlp_convs.{k}.weight" in sd:
            x = _conv_bn_relu_1d(x, sd, f"{prefix}.mlp_convs.{k}", f"{prefix}.mlp_bns.{k}")
            k += 1
        return x

    B, _, N = xyz_in.shape
    xyz0 = xyz_in[:, :3, :]
    l1_xyz, l1 = msg("sa1", xyz0, xyz_in, 512, 3)
    l2_xyz, l2 = msg("sa2", l1_xyz, l1, 128, 2)
    l3 = group_all("sa3", l2_xyz, l2)
    l2 = fp("fp3", 128, 1, l2, l3)
    l1 = fp("fp2", 512, 128, l1, l2)
    l0 = fp("fp1", N, 512, None, l1)
    seg = classifier(sd, l0)
    out = {"class_logits": seg, "l0": l0}
    for side in ("left", "right"):
        hf = attention(seg, l0, query_conv(sd, side, l0))
        p = f"{side}_mano_regressor"
        m_xyz, m1 = msg(p + ".sa1", xyz0, hf, 128, 2)
        h = group_all(p + ".sa2", m_xyz, m1).squeeze(-1)
        h = F.relu(F.linear(h, sd[p + ".mano_regressor.0.weight"], sd[p + ".mano_regressor.0.bias"]))
        h = _bn(h, sd, p + ".mano_regressor.2")
        prm = F.linear(h, sd[p + ".mano_regressor.4.weight"], sd[p + ".mano_regressor.4.bias"])
        args = {"global_orient": prm[:, :3], "hand_pose": prm[:, 3:3 + n_pose], "betas": prm[:, 3 + n_pose:-3], "transl": prm[:, -3:]}
        res = mano_hands64[side](**args)
        out[side] = {"vertices": res.vertices, "j3d": res.joints, "params": prm, "hand_features": hf, **args}
    return out
