"""CPU: the host-side pieces of the f16x2 range handling and of the stress inputs -- plane split and power-of-two scale as the
kernels apply them (csrc/planes.hpp), the bounds the packer hands to the kernels, and the test helpers themselves
(sc.rescale_hidden must not change the network function; oracle/stress.py must really produce near-ties)."""
import numpy as np
import pytest
import torch

import ref_pack
import stress_checkpoints as sc
from ev2hands_amd import pack, synth


def f16x2_scale(amax: np.float32) -> np.float32:
    """Python twin of planes.hpp: f16x2_scale -- the power of two s with amax * s in [2^14, 2^15)."""
    bits = np.float32(amax).view(np.uint32)
    E = int((bits >> np.uint32(23)) & np.uint32(0xFF))
    sb = 127 if E == 255 else min(268 - E, 200)
    return np.uint32(sb << 23).view(np.float32)


def test_scale_puts_the_maximum_in_the_top_binade():
    rng = np.random.default_rng(0)
    for a in np.concatenate([np.exp(rng.uniform(np.log(1e-20), np.log(1e30), 2000)), [1.0, 2.0 ** 14, 65504.0, 65536.0, 3.0e38]]).astype(np.float32):
        s = f16x2_scale(a)
        assert np.log2(float(s)) == round(np.log2(float(s)))                      # a power of two
        assert float(a) * float(s) < 2.0 ** 15                                    # never reaches the fp16 maximum 65504
        assert float(a) < 2.0 ** -58 or 2.0 ** 14 <= float(a) * float(s)         # (maxima below 2^-58: the scale is clamped at 2^73)
    assert float(f16x2_scale(np.float32(0.0))) == 2.0 ** 73                       # tiny / zero maxima: clamped
    assert float(f16x2_scale(np.float32(np.inf))) == 1.0 and float(f16x2_scale(np.float32(np.nan))) == 1.0


def test_two_plane_split_is_fp32_class_inside_the_scaled_range():
    """x = h + l with both planes fp16: after scaling by the window maximum the reconstruction error is <= 2^-22 |x| for values
    down to 2^-17 of the maximum and <= 2^-25 (absolute, in scaled units) below."""
    rng = np.random.default_rng(1)
    amax = np.float32(3.7e-4)
    x = (rng.uniform(-1, 1, 200000) * np.exp(rng.uniform(np.log(1e-9), 0, 200000))).astype(np.float32) * amax
    s = f16x2_scale(amax)
    xs = x * s
    h, l = ref_pack.split_bf16_planes(xs, 2)
    rec = h.view(np.float16).astype(np.float64) + l.view(np.float16).astype(np.float64)
    err = np.abs(rec - xs.astype(np.float64))
    big = np.abs(xs) >= 2.0 ** -3
    assert np.all(err[big] <= np.abs(xs[big]).astype(np.float64) * 2.0 ** -22)
    assert np.all(err[~big] <= 2.0 ** -25)
    assert np.isfinite(h.view(np.float16)).all() and np.isfinite(l.view(np.float16)).all()
    # unscaled, the same values lose their low plane to fp16 subnormals (the round-1 weakness)
    h0, l0 = ref_pack.split_bf16_planes(x, 2)
    rec0 = h0.view(np.float16).astype(np.float64) + l0.view(np.float16).astype(np.float64)
    rel0 = np.abs(rec0 - x.astype(np.float64)) / np.maximum(np.abs(x), 1e-30)
    assert np.median(rel0[np.abs(x) > amax / 8]) > 2.0 ** -16


def test_packed_bounds_are_upper_bounds():
    """The norms pack.py stores for the fused set-abstraction kernel bound what they must (ev2h_sa_desc / ev2h_sa_module)."""
    sd = synth.synth_state_dict(5, 3)
    pw = pack.PackedWeights(sd, "cpu", 5, "f16x2")
    for mod, name in ((pw.struct.sa1, "sa1"), (pw.struct.sa2, "sa2"), (pw.struct.mano_sa1[0], "left_mano_regressor.sa1")):
        W1f, b1 = pw.tensors[name + ".W1f"].double(), pw.tensors[name + ".b1"].double()
        assert mod.w1f_norm >= float(W1f.abs().sum(1).max()) and mod.b1_max >= float(b1.abs().max())
        assert mod.w1f_unscale > 0 and np.log2(mod.w1f_unscale) == round(np.log2(mod.w1f_unscale))
        for i in range(mod.nbranch):
            br = mod.br[i]
            W1x, W2, b2 = (pw.tensors[f"{name}.{i}.{k}"].double() for k in ("W1x", "W2", "b2"))
            assert br.w1x_norm >= float(W1x[:, :3].abs().sum(1).max())
            assert br.w2_norm >= float(W2.abs().sum(1).max()) and br.b2_max >= float(b2.abs().max())
            # bound of the hidden layer for any input inside the bound of layer 1
            x = torch.rand(64, W2.shape[1], dtype=torch.float64) * 3.0
            assert float((x @ W2.T + b2).abs().max()) <= br.w2_norm * 3.0 + br.b2_max
    assert pw.struct.sa2.W1fs and not pw.struct.sa1.W1fs           # K = 320 table: plane images; K = 8 tables: fp32 kernel


@pytest.mark.parametrize("alpha", [1e-4, 6e4])
def test_rescale_hidden_keeps_the_network_function(alpha):
    from oracle import mano_oracle, tehnet_oracle
    C, N, B, seed = 5, 256, 1, 4
    sd = synth.synth_state_dict(C, seed)
    hands = mano_oracle.make_hands(synth.synth_mano_assets("left", seed), synth.synth_mano_assets("right", seed))
    xyz, inits = synth.synth_cloud("E", B, C, N, seed), synth.fps_inits(B, N, seed)
    ta, tb = {}, {}
    with torch.no_grad():
        a = tehnet_oracle.tehnet_forward(sd, xyz.clone(), hands, fps_init=inits, trace=ta)
        b = tehnet_oracle.tehnet_forward(sc.rescale_hidden(sd, alpha), xyz.clone(), hands, fps_init=inits, trace=tb)
    rel = lambda x, y: float((x - y).abs().max() / y.abs().max())           # noqa: E731
    assert rel(b["class_logits"], a["class_logits"]) < 5e-6 and rel(b["left"]["vertices"], a["left"]["vertices"]) < 5e-6
    for k in ("l0_points", "l1_points", "l2_points"):                       # hidden tensors really are alpha times larger
        assert rel(tb[k] / alpha, ta[k]) < 5e-6


def test_near_tie_head_produces_near_ties():
    from oracle import mano_oracle, stress, tehnet_oracle
    C, N, B, seed = 4, 1024, 1, 31
    sd = synth.synth_state_dict(C, seed)
    hands = mano_oracle.make_hands(synth.synth_mano_assets("left", seed), synth.synth_mano_assets("right", seed))
    xyz, inits = synth.synth_cloud("E", B, C, N, seed), synth.fps_inits(B, N, seed)
    with torch.no_grad():
        plain = tehnet_oracle.tehnet_forward(sd, xyz.clone(), hands, fps_init=inits)
        tied = tehnet_oracle.tehnet_forward(stress.near_tie_state_dict(sd, xyz, inits, hands, 0.3), xyz.clone(), hands, fps_init=inits)
    m0, s0 = stress.margin_report(plain["class_logits"])
    m1, s1 = stress.margin_report(tied["class_logits"])
    assert float((m0 < 1e-3 * s0).float().mean()) == 0.0                    # the plain synthetic head never comes close to a tie
    assert float((m1 < 2e-5 * s1).float().mean()) > 0.01                    # the stress head does, for percent-level fractions
    assert len(torch.unique(tied["class_logits"].argmax(1))) >= 3           # and the classes really alternate


# ------------------------------------------------------------------------------------------------ channel equalisation
# (properties of the algorithm, checked on the numpy restatement tests/ref_pack.py; tests/test_pack_abi.py requires the library's
# packer, csrc/pack.hip, to reproduce that restatement -- factors and packed bytes -- exactly)
def _layer_spread(F):
    """worst log2(max / min) over the hidden tensors of (row norm of the producer) and of (column norm of a consumer)"""
    worst_r = worst_c = 0.0
    for _, producers, consumers in ref_pack.hidden_tensors():
        r = np.concatenate([np.sqrt((F[l]["W"].reshape(F[l]["W"].shape[0], -1) ** 2).sum(1) + F[l]["b"] ** 2) *
                            (np.abs(F[l]["ps"]) if k == "post" else 1.0) for l, k in producers])
        n = r.shape[0]
        worst_r = max(worst_r, float(np.log2(r.max() / r[r > 0].min())))
        for l, off in consumers:
            cols = F[l]["W"][:, off:off + n]
            cn = np.sqrt((cols ** 2).sum(axis=tuple(i for i in range(cols.ndim) if i != 1)))
            worst_c = max(worst_c, float(np.log2(cn.max() / cn[cn > 0].min())))
    return worst_r, worst_c


def test_hidden_tensor_table_matches_the_checkpoint_schema():
    F = ref_pack.fold_checkpoint(synth.synth_state_dict(5, 2))
    seen_rows, seen_cols = set(), {}
    for name, producers, consumers in ref_pack.hidden_tensors():
        n = sum(F[l]["W"].shape[0] for l, _ in producers)
        for l, _ in producers:
            assert l not in seen_rows, l                       # every layer's rows belong to exactly one tensor
            seen_rows.add(l)
        for l, off in consumers:
            assert F[l]["W"].shape[1] >= off + n, (name, l)
            for c in range(off, off + n):
                assert (l, c) not in seen_cols, (name, l, c)
            seen_cols.update({(l, c): name for c in range(off, off + n)})
    # every input column of every layer is either a hidden channel or a raw input (coordinates / input features / hand features)
    raw = {"sa1.0.0": 8, "sa1.1.0": 8, "sa1.2.0": 8, "sa2.0.0": 3, "sa2.1.0": 3, "sa3.0": 3}
    for side in ("left", "right"):
        p = f"{side}_mano_regressor"
        raw.update({p + ".sa1.0.0": 7, p + ".sa1.1.0": 7, p + ".sa2.0": 3})
    for l, L in F.items():
        covered = sum(1 for c in range(L["W"].shape[1]) if (l, c) in seen_cols)
        assert covered + raw.get(l, 0) == L["W"].shape[1], (l, covered, L["W"].shape)


@pytest.mark.parametrize("log2_spread", [8, 16])
def test_equalisation_undoes_per_channel_rescaling(log2_spread):
    """A checkpoint whose hidden channels were rescaled by 2^+-s (sc.rescale_channels: same network, other representative)
    packs to an equally well-conditioned representation: the accumulated factors absorb the rescaling up to the power-of-
    two rounding, and the row / column spreads the 16-bit planes see are those of the original checkpoint, not 2^(2s)."""
    sd = synth.synth_state_dict(4, 9)
    F0 = ref_pack.fold_checkpoint(sd)
    F1 = ref_pack.fold_checkpoint(sc.rescale_channels(sd, log2_spread, 9))
    r_raw, c_raw = _layer_spread(F1)
    assert r_raw > 1.5 * log2_spread and c_raw > 1.5 * log2_spread
    e0, e1 = ref_pack.equalize_channels(F0), ref_pack.equalize_channels(F1)
    for k, e in e1.items():
        assert np.all(np.log2(e) == np.round(np.log2(e))), k                          # exact powers of two
    r_eq, c_eq = _layer_spread(F1)
    r_ref, c_ref = _layer_spread(F0)
    assert r_eq <= r_ref + 2.5 and c_eq <= c_ref + 2.5, (r_eq, r_ref, c_eq, c_ref)


def test_equalised_pack_is_a_power_of_two_rescaling():
    """Equalised vs plain packing of one checkpoint: every packed fp32 weight differs by an exact power of two (or is equal), so
    every fp32 product and partial sum of the exact mode changes by an exact power of two as well."""
    sd = synth.synth_state_dict(5, 3)
    a = pack.PackedWeights(sd, "cpu", 5, "f32", equalize=False)
    b = pack.PackedWeights(sd, "cpu", 5, "f32", equalize=True)
    assert b.equalization and not a.equalization
    for k, ta in a.tensors.items():
        tb = b.tensors[k]
        nz = ta != 0
        assert torch.equal(nz, tb != 0), k
        q = torch.log2((tb[nz] / ta[nz]).double())
        assert torch.equal(q, q.round()), k
