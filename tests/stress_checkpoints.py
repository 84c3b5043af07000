"""TEST INFRASTRUCTURE: function-preserving (or oracle-evaluated) transforms of a synthetic checkpoint / event cloud that stress the
f16x2 arithmetic of the library -- trained-like per-channel BatchNorm spreads, globally rescaled hidden activations, heavy-tailed
weights, hot pixels, and (round 4) hidden activations made large by CORRELATED INPUTS / BatchNorm running statistics at unchanged
weight norms.  Used by tests/ and tools/ only; the product package does not contain them.
Reference facts (shapes, concatenation offsets): /root/reference/src/Ev2Hands/model/pointnet2_utils.py:155,248,261,307,
model/TEHNet.py:20-26,116-166,179-195."""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch

from ev2hands_amd.synth import (FP1_MLP, FP2_MLP, FP3_MLP, MANO_SA1_MLPS, MANO_SA2_MLP, SA1_MLPS, SA2_MLPS, SA3_MLP, hash_normal,
                                hash_randint, hash_uniform)


def rescale_hidden(sd: dict, alpha: float) -> "OrderedDict":
    """Checkpoint whose hidden activations are `alpha` times those of `sd` while the network function is unchanged
    (ReLU is positively homogeneous): every eval-BatchNorm's weight and bias are multiplied by alpha, and the columns of the next
    convolutions / linears that read such a scaled tensor are divided by alpha.  Raw inputs (coordinates, input channels, relative
    xyz) keep their columns; `*_query_conv.5` keeps its scale because the attention softmax (TEHNet.py:22-23) is not homogeneous in
    the query; the final layers (classifier.4, mano_regressor.4) divide their inputs only, so the outputs are the original ones.
    Test helper for the range handling of the f16x2 arithmetic (hidden activations far outside the fp16 range)."""
    a = float(alpha)
    out = OrderedDict((k, v.clone()) for k, v in sd.items())

    def raw_cols(key: str, width: int):
        """input columns of conv `key` that carry raw (unscaled) inputs"""
        if key.startswith(("sa1.conv_blocks.",)) and key.split(".")[3] == "0":
            return slice(0, width)                                        # [features, dx]: all raw
        if key.startswith("sa2.conv_blocks.") and key.split(".")[3] == "0":
            return slice(320, width)                                      # [l1_points | dx]
        if key.startswith("sa3.mlp_convs.0."):
            return slice(0, 3)                                            # [xyz | l2_points]
        for side in ("left", "right"):
            p = f"{side}_mano_regressor."
            if key.startswith(p + "sa1.conv_blocks.") and key.split(".")[4] == "0":
                return slice(4, width)                                    # [hand features (scaled) | dx]
            if key.startswith(p + "sa2.mlp_convs.0."):
                return slice(0, 3)
        return slice(0, 0)

    for k in list(out.keys()):
        if k.endswith("num_batches_tracked"):
            continue
        v = out[k]
        is_bn_affine = (".bn_blocks." in k or ".mlp_bns." in k or k.startswith("classifier.2.") or "_query_conv.2." in k
                        or ".mano_regressor.2." in k) and k.endswith((".weight", ".bias"))
        if is_bn_affine:
            out[k] = v * a
        elif k.endswith(".weight") and v.dim() >= 2 and not ("_query_conv.5." in k):
            w = v.clone()
            keep = raw_cols(k, w.shape[1])
            scaled = torch.ones(w.shape[1], dtype=torch.bool)
            scaled[keep] = False
            shape = (1, -1) + (1,) * (w.dim() - 2)
            factor = torch.where(scaled, torch.tensor(1.0 / a, dtype=torch.float64), torch.tensor(1.0, dtype=torch.float64))
            out[k] = (w.double() * factor.view(shape)).to(w.dtype)
    return out


# --------------------------------------------------------------------------- trained-like checkpoint transforms
def channel_wiring(in_channels: int = 4):
    """(producer BN prefix, channel count, [(consumer conv/linear key, first input column)]) for every hidden tensor of the
    network whose channels can be rescaled one by one without changing the network function: the producer is an eval BatchNorm
    (before or after a ReLU), the consumers are convolutions / linears that read the channels as input columns (through gathers,
    max-pooling, interpolation and concatenation, which all act per channel).  Concatenation offsets follow
    pointnet2_utils.py:155,248,261,307 and TEHNet.py:179-195.  Not listed, because a consumer is not linear in them: fp1's
    output (the attention's `value`, TEHNet.py:20-26), `*_query_conv.5` (softmax input) and the network outputs."""
    del in_channels
    wires = []

    def msg(prefix, mlps, consumers):
        off = 0
        for i, mlp in enumerate(mlps):
            for j in range(len(mlp) - 1):
                wires.append((f"{prefix}.bn_blocks.{i}.{j}", mlp[j], [(f"{prefix}.conv_blocks.{i}.{j + 1}.weight", 0)]))
            wires.append((f"{prefix}.bn_blocks.{i}.{len(mlp) - 1}", mlp[-1], [(k, o + off) for k, o in consumers]))
            off += mlp[-1]

    def stack(prefix, mlp, consumers):
        for k in range(len(mlp) - 1):
            wires.append((f"{prefix}.mlp_bns.{k}", mlp[k], [(f"{prefix}.mlp_convs.{k + 1}.weight", 0)]))
        wires.append((f"{prefix}.mlp_bns.{len(mlp) - 1}", mlp[-1], consumers))

    msg("sa1", SA1_MLPS, [("sa2.conv_blocks.0.0.weight", 0), ("sa2.conv_blocks.1.0.weight", 0), ("fp2.mlp_convs.0.weight", 0)])
    msg("sa2", SA2_MLPS, [("sa3.mlp_convs.0.weight", 3), ("fp3.mlp_convs.0.weight", 0)])
    stack("sa3", SA3_MLP, [("fp3.mlp_convs.0.weight", 512)])
    stack("fp3", FP3_MLP, [("fp2.mlp_convs.0.weight", 320)])
    stack("fp2", FP2_MLP, [("fp1.mlp_convs.0.weight", 0)])
    stack("fp1", FP1_MLP[:-1], [("fp1.mlp_convs.2.weight", 0)])
    wires.append(("classifier.2", 256, [("classifier.4.weight", 0)]))
    for side in ("left", "right"):
        wires.append((f"{side}_query_conv.2", 256, [(f"{side}_query_conv.4.weight", 0)]))
        p = f"{side}_mano_regressor"
        msg(p + ".sa1", MANO_SA1_MLPS, [(p + ".sa2.mlp_convs.0.weight", 3)])
        stack(p + ".sa2", MANO_SA2_MLP, [(p + ".mano_regressor.0.weight", 0)])
        wires.append((p + ".mano_regressor.2", 1024, [(p + ".mano_regressor.4.weight", 0)]))
    return wires


def rescale_channels(sd: dict, log2_spread: float, seed: int = 0, dead_fraction: float = 0.0, include_l0: bool = False) -> "OrderedDict":
    """Checkpoint with a wide PER-CHANNEL dynamic range inside every hidden tensor, network function unchanged: channel c of every
    hidden tensor listed by channel_wiring() is multiplied by alpha_c = 2^u, u uniform in [-log2_spread, +log2_spread] (the BN
    affine of the producer times alpha_c, column c of every consumer divided by alpha_c; float64, rounded once to fp32) -- what
    BatchNorm scales of a trained checkpoint look like, as opposed to rescale_hidden's single factor.  `dead_fraction` of the
    channels additionally get gamma = 0 (the channel is the constant relu(beta): a dead unit; this does change the function --
    the oracle evaluates the same checkpoint).  Test helper for the f16x2 arithmetic, whose accuracy depends on the spread of
    magnitudes inside one window's tensor (csrc/planes.hpp).
    include_l0: also fp1's output, with factors 2^u, u in [-2 log2_spread, 0].  That tensor is the attention's `value` as well
    (TEHNet.py:20-26), which has no weights to compensate: the context features shrink and the network function changes (the
    oracle evaluates the same checkpoint)."""
    out = OrderedDict((k, v.clone()) for k, v in sd.items())
    wires = channel_wiring()
    if include_l0:
        wires.append(("fp1.mlp_bns.2", FP1_MLP[-1], [("classifier.0.weight", 0), ("left_query_conv.0.weight", 0), ("right_query_conv.0.weight", 0)]))
    for bn, nch, consumers in wires:
        u = (hash_uniform("rescale/" + bn, (nch,), seed) * 2 - 1) * float(log2_spread)
        if bn == "fp1.mlp_bns.2":
            u = u - float(log2_spread)
        alpha = np.exp2(u)
        dead = hash_uniform("dead/" + bn, (nch,), seed) < dead_fraction
        g = out[bn + ".weight"].double().numpy() * alpha
        g[dead] = 0.0
        out[bn + ".weight"] = torch.from_numpy(g.astype(np.float32))
        out[bn + ".bias"] = torch.from_numpy((out[bn + ".bias"].double().numpy() * alpha).astype(np.float32))
        for key, off in consumers:
            w = out[key].double().numpy().copy()
            assert w.shape[1] >= off + nch, (key, w.shape, off, nch)
            shape = (1, nch) + (1,) * (w.ndim - 2)
            w[:, off:off + nch] = w[:, off:off + nch] / alpha.reshape(shape)
            out[key] = torch.from_numpy(w.astype(np.float32))
    return out


def heavy_tailed(sd: dict, sigma: float = 1.5, seed: int = 0) -> "OrderedDict":
    """Checkpoint whose convolution / linear weights are heavy-tailed: every weight is multiplied by exp(sigma z), z ~ N(0, 1)
    (log-normal; sigma = 1.5 puts the largest weight of a 256 x 256 layer ~400x above the median), and the layer is renormalised
    to its old Frobenius norm so that activations keep their order of magnitude.  A few large weights next to many small ones
    is what the per-matrix power-of-two scale of the fp16 weight planes (ev2h_plane_unscale) has to cope with."""
    out = OrderedDict((k, v.clone()) for k, v in sd.items())
    for k, v in sd.items():
        if k.endswith(".weight") and v.dim() >= 2:
            w = v.double().numpy()
            w2 = w * np.exp(sigma * hash_normal("heavy/" + k, w.shape, seed))
            w2 *= np.linalg.norm(w) / max(np.linalg.norm(w2), 1e-300)
            out[k] = torch.from_numpy(w2.astype(np.float32))
    return out


def add_outlier_points(xyz: torch.Tensor, value: float, channel: int = 3, per_window: int = 1, seed: int = 0) -> torch.Tensor:
    """Copy of an event cloud [B, C, N] in which `per_window` points of every window carry `value` in feature channel `channel`
    (an event-count channel of the C = 5 representation: a hot pixel that fired `value` times, ev2hands_r.py:118-130)."""
    out = xyz.clone()
    B, _, N = out.shape
    for b in range(B):
        idx = hash_randint(f"outlier/{b}", 0, N, (per_window,), seed)
        out[b, channel, torch.from_numpy(idx)] = float(value)
    return out


def coherent_channels(sd: dict, fraction: float = 0.25, seed: int = 0, probe=None) -> "OrderedDict":
    """Checkpoint whose hidden activations are large in a way NO WEIGHT NORM shows (what a data-free equalisation rule cannot see):
    in every convolution / linear that reads a hidden tensor, a `fraction` of the output channels get the ABSOLUTE VALUES of their
    weights on the hidden input columns (raw-input columns keep their signs).  Hidden inputs are post-ReLU, i.e. non-negative and
    positively correlated through their common mean, so such a row adds its inputs coherently -- a pre-activation of about
    |w|_1 mean(h) instead of |w|_2 std(h), a factor ~sqrt(fan-in) -- at EXACTLY the same row norm and column norms; the eval
    BatchNorm keeps its (now stale) running statistics and does not remove that mean; and because coherent rows also read the
    previous layer's coherent channels, the effect compounds through the three layers of every MLP.  The result: inside one
    tensor some channels are orders of magnitude above the others although the packer's row / column norms are level.
    This is a DIFFERENT network (the oracle evaluates the same checkpoint), not a re-parameterisation.
    probe: optional callable(state_dict) -> dict name -> float, the magnitude of the network outputs; the last layers (classifier.4,
    *_mano_regressor.mano_regressor.4) are then scaled so that the outputs are O(1) again (hand poses of thousands of radians
    would make every arithmetic mode miss the bar through MANO's trigonometry, as the 1e8-event hot pixel does)."""
    out = OrderedDict((k, v.clone()) for k, v in sd.items())
    for bn, nch, consumers in channel_wiring():
        for key, off in consumers:
            w = out[key].double().numpy().copy()
            rows = hash_uniform("coherent/" + key, (w.shape[0],), seed) < fraction
            blk = w[:, off:off + nch]
            blk[rows] = np.abs(blk[rows])
            w[:, off:off + nch] = blk
            out[key] = torch.from_numpy(w.astype(np.float32))
    if probe is not None:
        mag = probe(out)
        for key, m in mag.items():
            if m > 0:
                out[key + ".weight"] = out[key + ".weight"] / float(m)
                out[key + ".bias"] = out[key + ".bias"] / float(m)
    return out
