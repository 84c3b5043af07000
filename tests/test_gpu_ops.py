"""GPU: every C-ABI operator against the oracle (or an fp64 restatement of the same formula) on
seeded inputs.  Index outputs must be identical; float outputs within 1e-4 relative (||d||inf/||ref||inf),
most are far tighter.  Runs through ev2hands_amd.ops -> ctypes -> libev2hands_hip.so."""
import numpy as np
import pytest
import torch

from ev2hands_amd import synth

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def rel(a, b):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def cloud_xyz(kind, B, N, seed):
    return synth.synth_cloud(kind, B, 4, N, seed)[:, :3].permute(0, 2, 1).contiguous()


@pytest.mark.parametrize("kind,N,S", [("U", 2048, 512), ("E", 2048, 128), ("U", 512, 128), ("E", 300, 64), ("U", 8192, 512),
                                      # beyond the register / LDS-resident sizes: 1024-thread variants reading the winner from global memory
                                      ("U", 8193, 64), ("E", 20000, 512), ("U", 32768, 128)])
def test_fps_indices_exact(kind, N, S):
    _need_gpu()
    from ev2hands_amd import ops
    from oracle import tehnet_oracle as O
    B = 3
    xyz = cloud_xyz(kind, B, N, 11)
    init = synth.hash_randint("init", 0, N, (B,), 5)
    init = torch.from_numpy(init)
    ref = O.farthest_point_sample(xyz, S, init)
    got = ops.farthest_point_sample(xyz.cuda(), S, init).cpu()
    assert torch.equal(got, ref), f"{(got != ref).sum().item()} of {ref.numel()} FPS indices differ"


def test_fps_degenerate_duplicates():
    """fewer unique points than samples: argmax ties must resolve to the first index (torch.max)."""
    _need_gpu()
    from ev2hands_amd import ops
    from oracle import tehnet_oracle as O
    B, N, S = 2, 256, 64
    base = cloud_xyz("U", B, 16, 3)
    xyz = base.repeat(1, N // 16, 1).contiguous()
    init = torch.tensor([5, 200])
    ref = O.farthest_point_sample(xyz, S, init)
    got = ops.farthest_point_sample(xyz.cuda(), S, init).cpu()
    assert torch.equal(got, ref)


@pytest.mark.parametrize("kind,N,S,radii,ks", [
    ("U", 2048, 512, [0.1, 0.2, 0.4], [32, 64, 128]),
    ("E", 2048, 128, [0.4, 0.8], [64, 128]),
    ("E", 512, 128, [0.4, 0.8], [64, 128]),
    ("U", 300, 40, [0.2], [32]),
    ("E", 20000, 512, [0.1, 0.2, 0.4], [32, 64, 128]),          # > 8192 points: the variant that reads the cloud from global memory
    ("U", 32768, 128, [0.05, 0.8], [64, 128]),
])
def test_ball_query_exact(kind, N, S, radii, ks):
    _need_gpu()
    from ev2hands_amd import ops
    from oracle import tehnet_oracle as O
    B = 2
    xyz = cloud_xyz(kind, B, N, 21)
    fps = O.farthest_point_sample(xyz, S, torch.zeros(B, dtype=torch.long))
    ctr = O.gather_points(xyz, fps)
    got, cnt = ops.query_ball_point(radii, ks, xyz.cuda(), ctr.cuda(), return_counts=True)
    for i, (r, k) in enumerate(zip(radii, ks)):
        ref = O.ball_query(r, k, xyz, ctr)
        g = got[i].cpu()
        assert torch.equal(g, ref), f"radius {r}: {(g != ref).sum().item()} of {ref.numel()} group indices differ"
        d = O.pairwise_sqdist(ctr, xyz)
        true_cnt = (~(d > r ** 2)).sum(-1).clamp(max=k)
        assert torch.equal(cnt[:, :, i].cpu().long(), true_cnt)


@pytest.mark.parametrize("kind,N1,N2,D", [("U", 2048, 512, 128), ("E", 2048, 512, 128), ("E", 512, 128, 256)])
def test_three_nn_interp(kind, N1, N2, D):
    _need_gpu()
    from ev2hands_amd import ops
    from oracle import tehnet_oracle as O
    B = 2
    xyz1 = cloud_xyz(kind, B, N1, 31)
    fps = O.farthest_point_sample(xyz1, N2, torch.zeros(B, dtype=torch.long))
    xyz2 = O.gather_points(xyz1, fps)                       # subset => coincident points, d ~ 0 +- 5e-7
    f2 = torch.from_numpy(synth.hash_normal("f2", (B, N2, D), 1)).float()
    idx, w = O.three_nn_weights(xyz1, xyz2)
    ref = (O.gather_points(f2, idx) * w.view(B, N1, 3, 1)).sum(dim=2)
    out, gi, gw = ops.three_nn_interpolate(xyz1.cuda(), xyz2.cuda(), f2.cuda())
    assert torch.equal(gi.cpu(), idx), f"{(gi.cpu() != idx).sum().item()} of {idx.numel()} 3-NN indices differ"
    assert rel(gw, w) < 1e-5
    assert rel(out, ref) < 1e-5


@pytest.mark.parametrize("precision,tol", [("f16x2", 6e-6), ("bf16x3", 6e-6), ("bf16", 3e-2), ("f16", 4e-3)])
@pytest.mark.parametrize("kind,B,N1,N2,mag", [("E", 2, 2048, 512, 1.0), ("U", 3, 1000, 512, 1.0), ("E", 1, 130, 128, 1.0),
                                              ("E", 2, 2048, 512, 3e-5), ("E", 2, 2048, 512, 4e5)])
def test_feature_propagation_fused(kind, B, N1, N2, mag, precision, tol):
    """ev2h_fp_mlp (3-NN blend of layer-1 table rows + layers 2-3 in one kernel) against the reference's order of operations
    -- interpolate, then three Conv1d+ReLU (pointnet2_utils.py:296-316) -- in float64.  Ragged N (1000, 130: partial strips),
    and input magnitudes that need the F16X2 range handling (mag != 1: with records; the hidden layers scale with the input)."""
    _need_gpu()
    from ev2hands_amd import ops
    from oracle import tehnet_oracle as O
    xyz1 = cloud_xyz(kind, B, N1, 41)
    fps = O.farthest_point_sample(xyz1, N2, torch.zeros(B, dtype=torch.long))
    xyz2 = O.gather_points(xyz1, fps)
    f2 = torch.from_numpy(synth.hash_normal("f2", (B, N2, 128), 5)).float() * mag
    f2[B - 1] *= 0.01                                         # windows of different magnitude in one batch
    Ws = [torch.from_numpy(synth.hash_normal(f"W{i}", (o, k), 6 + i) / np.sqrt(k)).float() for i, (o, k) in enumerate(((128, 128), (128, 128), (256, 128)))]
    bs = [torch.from_numpy(synth.hash_normal(f"b{i}", (o,), 9 + i) * 0.1).float() * mag for i, o in enumerate((128, 128, 256))]
    ranges = precision in ("f16x2", "f16")         # the fp16-plane modes: range records
    x_amax = out_amax = None
    if ranges:
        x_amax = ops.range_record(B, "cuda")
        x_amax.copy_(f2.abs().amax(dim=(1, 2)).view(torch.int32).cuda())
        out_amax = ops.range_record(B, "cuda")
    cu = lambda t: t.cuda()                                   # noqa: E731
    out, gi, gw = ops.feature_propagation(cu(xyz1), cu(xyz2), cu(f2), cu(Ws[0]), cu(bs[0]), cu(Ws[1]), cu(bs[1]), cu(Ws[2]), cu(bs[2]),
                                          precision=precision, ranges=ranges, x_amax=x_amax, out_amax=out_amax)
    # the neighbour search itself is test_three_nn_interp's subject (and event clouds hold duplicate points: equal distances, whose
    # order the reference leaves to an unstable sort) -- here the weights must match and the blend is checked with the kernel's own choice
    _, w = O.three_nn_weights(xyz1, xyz2)
    assert rel(gw, w) < 1e-5
    h = (O.gather_points(f2, gi.cpu()).double() * gw.cpu().double().view(B, N1, 3, 1)).sum(dim=2)
    for W, b in zip(Ws, bs):
        h = torch.relu(h @ W.double().T + b.double())
    for b in range(B):                                        # per window: a window's accuracy must not depend on its neighbours
        assert rel(out[b], h[b].float()) < tol, (b, rel(out[b], h[b].float()))
    if ranges:
        assert torch.equal(ops.range_values(out_amax).cpu(), out.abs().amax(dim=(1, 2)).cpu())


@pytest.mark.parametrize("precision,tol", [("f16x2", 6e-6), ("bf16x3", 6e-6), ("bf16", 3e-2), ("f16", 4e-3)])
@pytest.mark.parametrize("B,N,mag", [(2, 2048, 1.0), (3, 1000, 1.0), (1, 130, 1.0), (2, 512, 1e-5), (2, 512, 3e6)])
def test_row_chain_segmentation_head(B, N, mag, precision, tol):
    """ev2h_fp_mlp on plain rows (both classifier layers in one kernel, logits point-major and channel-major) against float64;
    signed inputs, ragged N, windows of different magnitude, and -- F16X2 -- magnitudes far outside fp16 with range records."""
    _need_gpu()
    from ev2hands_amd import ops
    X = torch.from_numpy(synth.hash_normal("X", (B, N, 256), 21)).float() * mag
    X[B - 1] *= 0.03
    W2 = torch.from_numpy(synth.hash_normal("W2", (256, 256), 22) / 16).float()
    b2 = torch.from_numpy(synth.hash_normal("b2", (256,), 23) * 0.1).float() * mag
    W3 = torch.from_numpy(synth.hash_normal("W3", (4, 256), 24) / 16).float()
    b3 = torch.from_numpy(synth.hash_normal("b3", (4,), 25) * 0.1).float() * mag
    ref = torch.relu(X.double() @ W2.double().T + b2.double()) @ W3.double().T + b3.double()
    xa = oa = None
    if precision in ("f16x2", "f16"):
        xa = ops.range_record(B, "cuda")
        xa.copy_(X.abs().amax(dim=(1, 2)).view(torch.int32).cuda())
        oa = ops.range_record(B, "cuda")
    out, out_cm = ops.row_chain(X.cuda(), W2.cuda(), b2.cuda(), W3.cuda(), b3.cuda(), precision, x_amax=xa, out_amax=oa)
    for b in range(B):
        assert rel(out[b], ref[b].float()) < tol, (b, rel(out[b], ref[b].float()))
    assert torch.equal(out_cm, out.permute(0, 2, 1))
    if oa is not None:
        assert torch.equal(ops.range_values(oa).cpu(), out.abs().amax(dim=(1, 2)).cpu())


GEMM_CASES = [
    # M,   N,   K,  relu, post, taps, rowmax, group
    (256, 128, 128, True, False, 1, 0, 0),
    (300, 160, 8, False, False, 1, 0, 0),
    (384, 256, 520, True, False, 1, 0, 0),
    (1024, 4, 256, False, False, 1, 0, 0),
    (5, 22, 1024, False, False, 1, 0, 0),
    (7, 1024, 512, True, True, 1, 0, 0),
    (512, 256, 256, True, True, 3, 0, 0),
    (1024, 128, 64, True, False, 3, 0, 0),        # 256-row sequences: interior tile boundaries exercise the row halo
    (256, 512, 256, True, False, 1, 128, 0),
    (384, 256, 512, True, False, 1, 0, 128),
    (4096, 256, 576, True, False, 1, 0, 0),
]


@pytest.mark.parametrize("precision,tol,wrows", [("f32", 2e-6, 128), ("bf16x3", 3e-6, 128), ("bf16x3", 3e-6, 256), ("f16x2", 6e-6, 128), ("f16x2", 6e-6, 256), ("bf16", 2e-2, 128),
                                                 ("bf16", 2e-2, 256), ("f16", 3e-3, 128), ("f16", 3e-3, 256)])
@pytest.mark.parametrize("M,N,K,relu,post,taps,rowmax,group", GEMM_CASES)
def test_gemm(M, N, K, relu, post, taps, rowmax, group, precision, tol, wrows):
    _need_gpu()
    from ev2hands_amd import ops
    X = torch.from_numpy(synth.hash_normal("X", (M, K), 2)).float()
    W = torch.from_numpy(synth.hash_normal("W", (N, K * taps), 3) / np.sqrt(K * taps)).float()
    ngrp = M // group if group else 1
    b = torch.from_numpy(synth.hash_normal("b", (ngrp, N), 4)).float()
    ps = torch.from_numpy(0.5 + synth.hash_uniform("ps", (N,), 5)).float() if post else None
    pt = torch.from_numpy(synth.hash_normal("pt", (N,), 6)).float() if post else None
    Xd, Wd = X.double(), W.double()
    if taps == 3:
        seq = 256 if M >= 1024 else 128
        Xs = Xd.view(M // seq, seq, K)
        z = torch.zeros(M // seq, 1, K, dtype=torch.float64)
        Xcat = torch.cat([torch.cat([z, Xs[:, :-1]], 1), Xs, torch.cat([Xs[:, 1:], z], 1)], 2).view(M, 3 * K)
        ref = Xcat @ Wd.t()
    else:
        seq = 0
        ref = Xd @ Wd.t()
    ref = ref + (b.double().repeat_interleave(group, 0) if group else b.double())
    if relu:
        ref = ref.clamp_min(0)
    if post:
        ref = ref * ps.double() + pt.double()
    if rowmax:
        ref = ref.view(M // rowmax, rowmax, N).max(1)[0]
    got = ops.dense(X.cuda(), W.cuda(), b.cuda() if group else b[0].cuda(), relu, ps.cuda() if post else None,
                    pt.cuda() if post else None, taps, seq, rowmax, group, K, precision, w_tile_rows=wrows)
    assert got.shape == ref.shape
    err = rel(got, ref)
    print(f"gemm M={M} N={N} K={K} taps={taps} {precision}: rel err {err:.2e}")
    assert err < tol


@pytest.mark.parametrize("M,N,K,relu,post", [(256, 1024, 512, True, True), (5, 22, 1024, False, False), (1, 256, 1024, False, False),
                                              (37, 70, 48, True, False), (2048, 22, 1024, False, False)])
def test_gemm_skinny_rows_are_windows(M, N, K, relu, post):
    """The one-row-per-window layers (ev2h_gemm_desc.skinny): exact fp32 whatever the precision mode (bit-identical across modes),
    a row's result independent of how many rows the call has, range record of the output maintained."""
    _need_gpu()
    from ev2hands_amd import ops
    X = torch.from_numpy(synth.hash_normal("X", (M, K), 52)).float() * 3e5          # far outside fp16: no planes are involved
    W = torch.from_numpy(synth.hash_normal("W", (N, K), 53) / np.sqrt(K)).float()
    b = torch.from_numpy(synth.hash_normal("b", (N,), 54)).float() * 3e5
    ps = torch.from_numpy(0.5 + synth.hash_uniform("ps", (N,), 55)).float() if post else None
    pt = torch.from_numpy(synth.hash_normal("pt", (N,), 56)).float() if post else None
    ref = X.double() @ W.double().t() + b.double()
    if relu:
        ref = ref.clamp_min(0)
    if post:
        ref = ref * ps.double() + pt.double()
    cu = lambda t: None if t is None else t.cuda()            # noqa: E731
    outs = {}
    for prec in ("f32", "f16x2", "bf16"):
        ya = ops.range_record(M, "cuda")
        outs[prec] = ops.dense(X.cuda(), W.cuda(), b.cuda(), relu, cu(ps), cu(pt), precision=prec, skinny=True, y_amax=ya, y_group_rows=1)
        if prec == "f16x2":
            assert torch.equal(ops.range_values(ya).cpu(), outs[prec].abs().amax(1).cpu())
    assert rel(outs["f32"], ref) < 2e-6
    assert torch.equal(outs["f32"], outs["f16x2"]) and torch.equal(outs["f32"], outs["bf16"])
    one = ops.dense(X[M // 2:M // 2 + 1].cuda().contiguous(), W.cuda(), b.cuda(), relu, cu(ps), cu(pt), precision="f16x2", skinny=True)
    assert torch.equal(one[0], outs["f32"][M // 2])


@pytest.mark.parametrize("precision,tol", [("f16x2", 4e-6), ("f16", 2e-3)])
@pytest.mark.parametrize("wscale", [1e-2, 1e-4, 1e-6, 30.0])
@pytest.mark.parametrize("N", [256, 22])
def test_f16x2_weight_magnitude_does_not_matter(wscale, N, precision, tol):
    """fp16 has 5 exponent bits: without the exact power-of-two pre-scaling of the weight planes (pack.py: plane_unscale) the low
    plane of small weights is subnormal and the error reaches 2.6e-4 at |W| ~ 1e-4.  N = 22: the head layers, whose W is split
    in the kernel (no plane image)."""
    _need_gpu()
    from ev2hands_amd import ops
    M, K = 1024, 256
    X = torch.from_numpy(synth.hash_normal("X", (M, K), 31)).float().cuda()
    W = (torch.from_numpy(synth.hash_normal("W", (N, K), 32) / np.sqrt(K)).float() * wscale).cuda()
    b = (torch.from_numpy(synth.hash_normal("b", (N,), 33)).float() * wscale).cuda()
    ref = X.double() @ W.double().t() + b.double()
    got = ops.dense(X, W, b, False, precision=precision)
    assert rel(got, ref) < tol


@pytest.mark.parametrize("wscale", [1e-2, 1e-3])
def test_sa_f16x2_small_weights(wscale):
    """Small layer-2 / large layer-3 weights with activations kept O(1) .. O(1e3): the weight planes are pre-scaled exactly, so the
    error stays at the 1e-6 level (activations themselves must stay within fp16's range)."""
    _need_gpu()
    from ev2hands_amd import ops
    C1, C2, C3, K, B, Npts, S = 64, 96, 128, 128, 2, 512, 24
    g = lambda n, s, sc=1.0: torch.from_numpy(synth.hash_normal(n, s, 77) * sc).float()
    P1 = g("P1", (B, Npts, C1), 1.0 / wscale); xyz = cloud_xyz("U", B, Npts, 45); ctr = xyz[:, :S].contiguous()
    gidx = torch.from_numpy(synth.hash_randint("gi", 0, Npts, (B, S, K), 78)).int()
    W1x = g("W1x", (C1, 3), 0.5 / wscale)
    W2, b2 = g("W2", (C2, C1), C1 ** -0.5 * wscale), g("b2", (C2,), 0.1)                     # h2 stays O(1)
    W3, b3 = g("W3", (C3, C2), C2 ** -0.5 * wscale), g("b3", (C3,), 0.1 * wscale)
    bi = torch.arange(B).view(B, 1, 1)
    rows = P1.double()[bi, gidx.long()]
    dxyz = (xyz[bi, gidx.long()] - ctr.view(B, S, 1, 3)).double()
    h1 = (rows + dxyz @ W1x.double().t()).clamp_min(0)
    h2 = (h1 @ W2.double().t() + b2.double()).clamp_min(0)
    ref = (h2 @ W3.double().t() + b3.double()).clamp_min(0).max(2)[0]
    up = lambda x, m: (x + m - 1) // m * m
    W1x4 = torch.zeros(C1, 4); W1x4[:, :3] = W1x
    W2p = torch.zeros(up(C2, 32), C1); W2p[:C2] = W2
    b2p = torch.zeros(up(C2, 32)); b2p[:C2] = b2
    W3p = torch.zeros(C3, up(C2, 8)); W3p[:, :C2] = W3
    got = ops.sa_mlp_max(P1.cuda(), ops.pack_points(xyz.cuda()), ops.pack_points(ctr.cuda()), gidx.cuda(), W1x4.cuda(), W2p.cuda(),
                         b2p.cuda(), W3p.cuda(), b3.cuda(), C2, "f16x2")
    assert rel(got, ref) < 8e-6


@pytest.mark.parametrize("precision,tol", [("f16x2", 8e-6), ("f16", 3e-3)])
@pytest.mark.parametrize("mag", [1.0, 1e3, 1e6, 1e-4])
def test_sa_mlp_max_table_form_with_range_records(mag, precision, tol):
    """[r6] The TABLE form of the fused set abstraction (enc.sa2's) with the fp16-plane range arguments on, at layer-1 magnitudes
    1e-4 ... 1e6 (table, relative-xyz weights and the next layer's columns scaled together; the function is unchanged): the table
    arrives stored with its power of two (p1_scale), the kernel's layer 1 must produce s1 H1 without leaving the fp16 range.
    F16 runs layer 1 on the matrix pipe with launch-constant powers of two on the weight side (SaBP::a1x) -- the 1e6 case caught
    them being computed before the range arguments were set (a1x = 1: weights of 1e6 in an fp16 plane)."""
    _need_gpu()
    from ev2hands_amd import ops
    C1, C2, C3, K, B, Npts, S = 128, 128, 256, 64, 2, 512, 24
    g = lambda n, s, sc=1.0: torch.from_numpy(synth.hash_normal(n, s, 77) * sc).float()
    P1 = g("P1", (B, Npts, C1), mag)
    xyz = cloud_xyz("U", B, Npts, 45)
    ctr = xyz[:, :S].contiguous()
    gidx = torch.from_numpy(synth.hash_randint("gi", 0, Npts, (B, S, K), 78)).int()
    W1x = g("W1x", (C1, 3), 0.5 * mag)
    W2, b2 = g("W2", (C2, C1), C1 ** -0.5 / mag), g("b2", (C2,), 0.1)
    W3, b3 = g("W3", (C3, C2), C2 ** -0.5), g("b3", (C3,), 0.1)
    bi = torch.arange(B).view(B, 1, 1)
    rows = P1.double()[bi, gidx.long()]
    dxyz = (xyz[bi, gidx.long()] - ctr.view(B, S, 1, 3)).double()
    h1 = (rows + dxyz @ W1x.double().t()).clamp_min(0)
    h2 = (h1 @ W2.double().t() + b2.double()).clamp_min(0)
    ref = (h2 @ W3.double().t() + b3.double()).clamp_min(0).max(2)[0]
    W1x4 = torch.zeros(C1, 4); W1x4[:, :3] = W1x
    dmax = float(dxyz.abs().max()) * 1.0001
    bound = P1.abs().amax((1, 2)) + float(W1x.abs().sum(1).max()) * dmax            # what the table's producer bounds layer 1 by
    if precision == "f16":           # one power of two for the whole chain (ev2h_sa_desc.p1_scale, F16 contract): also (s / u2)(|W2|_1 B1 + max|b2|) < 2^15
        l1 = float(W2.abs().sum(1).max())
        u2 = 2.0 ** np.floor(np.log2(l1))
        bound = torch.maximum(bound, (l1 * 1.000001 * bound + float(b2.abs().max())) / u2)
    sc_ = torch.exp2(torch.floor(torch.log2(32768.0 / bound)))
    sc_ = torch.where(sc_ * bound >= 32768.0, sc_ / 2, sc_)
    P1s = (P1 * sc_.view(B, 1, 1)).contiguous()                                    # exact: powers of two
    p1_amax = ops.range_record(B, "cuda")
    p1_amax.view(torch.float32).copy_(P1s.abs().amax((1, 2)))
    out_amax = ops.range_record(B, "cuda")
    got = ops.sa_mlp_max(P1s.cuda(), ops.pack_points(xyz.cuda()), ops.pack_points(ctr.cuda()), gidx.cuda(), W1x4.cuda(), W2.cuda(), b2.cuda(), W3.cuda(),
                         b3.cuda(), C2, precision, p1_scale=sc_.cuda(), p1_amax=p1_amax, dmax=dmax, out_amax=out_amax)
    err = rel(got, ref)
    print(f"table form with range records, {precision}, layer-1 magnitude {mag:g}: rel err {err:.2e}")
    assert torch.isfinite(got).all() and err < tol
    assert torch.equal(ops.range_values(out_amax).cpu(), got.abs().amax(dim=(1, 2)).cpu())


def test_f16_table_scale_that_breaks_the_chain_contract_is_loud():
    """[r6] In `f16` the table's storage scale serves the whole chain (include/ev2hands_hip.h, ev2h_sa_desc.p1_scale).  A producer that
    bounds layer 1 only -- fine for f16x2 -- with a second layer whose bias pushes H2 far above H1: the kernel must not saturate
    silently; the affected windows come back NaN."""
    _need_gpu()
    from ev2hands_amd import ops
    C1, C2, C3, K, B, Npts, S = 64, 64, 128, 64, 2, 256, 8
    g = lambda n, s, sc=1.0: torch.from_numpy(synth.hash_normal(n, s, 79) * sc).float()
    P1 = g("P1", (B, Npts, C1))
    xyz = cloud_xyz("U", B, Npts, 46)
    ctr = xyz[:, :S].contiguous()
    gidx = torch.from_numpy(synth.hash_randint("gi", 0, Npts, (B, S, K), 80)).int()
    W1x4 = torch.zeros(C1, 4); W1x4[:, :3] = g("W1x", (C1, 3), 0.5)
    W2, b2 = g("W2", (C2, C1), C1 ** -0.5), g("b2", (C2,), 1e4)              # H2 ~ 1e4 next to H1 ~ 1
    W3, b3 = g("W3", (C3, C2), C2 ** -0.5), g("b3", (C3,), 0.1)
    bi = torch.arange(B).view(B, 1, 1)
    dmax = float((xyz[bi, gidx.long()] - ctr.view(B, S, 1, 3)).abs().max()) * 1.0001
    bound = P1.abs().amax((1, 2)) + float(W1x4.abs().sum(1).max()) * dmax     # layer 1 only
    sc_ = torch.exp2(torch.floor(torch.log2(32768.0 / bound)))
    sc_ = torch.where(sc_ * bound >= 32768.0, sc_ / 2, sc_)
    P1s = (P1 * sc_.view(B, 1, 1)).contiguous()
    p1_amax = ops.range_record(B, "cuda")
    p1_amax.view(torch.float32).copy_(P1s.abs().amax((1, 2)))
    run = lambda precision: ops.sa_mlp_max(P1s.cuda(), ops.pack_points(xyz.cuda()), ops.pack_points(ctr.cuda()), gidx.cuda(), W1x4.cuda(), W2.cuda(), b2.cuda(),
                                           W3.cuda(), b3.cuda(), C2, precision, p1_scale=sc_.cuda(), p1_amax=p1_amax, dmax=dmax)
    assert torch.isfinite(run("f16x2")).all()             # every layer has its own power of two there
    assert torch.isnan(run("f16")).all()


SA_CASES = [(32, 32, 64, 32), (64, 64, 128, 64), (64, 96, 128, 128), (128, 128, 256, 64), (128, 196, 256, 128)]


@pytest.mark.parametrize("precision,tol", [("f32", 2e-6), ("bf16x3", 4e-6), ("f16x2", 8e-6), ("bf16", 2e-2), ("f16", 3e-3)])
@pytest.mark.parametrize("C1,C2,C3,K", SA_CASES)
def test_sa_mlp_max(C1, C2, C3, K, precision, tol):
    _need_gpu()
    from ev2hands_amd import ops
    B, Npts, S = 2, 512, 37          # S deliberately not a multiple of the 8 groups per workgroup
    g = lambda n, s, sc=1.0: torch.from_numpy(synth.hash_normal(n, s, C1 + C2) * sc).float()
    P1 = g("P1", (B, Npts, C1))
    xyz = cloud_xyz("U", B, Npts, 41)
    ctr = xyz[:, :S].contiguous()
    gidx = torch.from_numpy(synth.hash_randint("gi", 0, Npts, (B, S, K), C3)).int()
    W1x = g("W1x", (C1, 3), 0.5)
    W2, b2 = g("W2", (C2, C1), C1 ** -0.5), g("b2", (C2,), 0.1)
    W3, b3 = g("W3", (C3, C2), C2 ** -0.5), g("b3", (C3,), 0.1)
    # fp64 restatement of pointnet2_utils.py:244-257 with layer 1 split as in the kernel
    bi = torch.arange(B).view(B, 1, 1)
    rows = P1.double()[bi, gidx.long()]                                        # [B,S,K,C1]
    dxyz = (xyz[bi, gidx.long()] - ctr.view(B, S, 1, 3)).double()               # fp32 subtraction, like the reference
    h1 = (rows + dxyz @ W1x.double().t()).clamp_min(0)
    h2 = (h1 @ W2.double().t() + b2.double()).clamp_min(0)
    h3 = (h2 @ W3.double().t() + b3.double()).clamp_min(0)
    ref = h3.max(2)[0]
    up = lambda x, m: (x + m - 1) // m * m
    W1x4 = torch.zeros(C1, 4); W1x4[:, :3] = W1x
    W2p = torch.zeros(up(C2, 32), C1); W2p[:C2] = W2
    b2p = torch.zeros(up(C2, 32)); b2p[:C2] = b2
    W3p = torch.zeros(C3, up(C2, 8)); W3p[:, :C2] = W3
    pts4 = ops.pack_points(xyz.cuda())
    ctr4 = ops.pack_points(ctr.cuda())
    got = ops.sa_mlp_max(P1.cuda(), pts4, ctr4, gidx.cuda(), W1x4.cuda(), W2p.cuda(), b2p.cuda(), W3p.cuda(), b3.cuda(), C2,
                         precision)
    err = rel(got, ref)
    print(f"sa<{C1},{C2},{C3}> {precision}: rel err {err:.2e}")
    assert err < tol


@pytest.mark.parametrize("precision,tol", [("f16x2", 8e-6), ("f16", 3e-3)])
@pytest.mark.parametrize("mag", [1.0, 1e-3, 3e4])
@pytest.mark.parametrize("C1,C2,C3,K", SA_CASES)
def test_sa_mlp_max_f16x2_layer1_from_raw_features(C1, C2, C3, K, mag, precision, tol):
    """F16X2 (and F16 [r6]: the same layer 1, one fp16 plane in layers 2-3): layer 1 on the matrix pipe straight from the raw feature rows, one power of two PER NEIGHBOUR (ev2h_sa_desc.feat), with
    the range handling on: features of magnitude `mag` next to O(1) coordinates, and one neighbour per window whose feature is 1e6 x
    larger (a hot pixel) -- fp32-class against the float64 restatement of pointnet2_utils.py:244-257."""
    _need_gpu()
    from ev2hands_amd import ops
    B, Npts, S, nfeat = 2, 512, 37, 5
    g = lambda n, s, sc=1.0: torch.from_numpy(synth.hash_normal(n, s, C1 + C2 + 3) * sc).float()
    feat = torch.zeros(B, Npts, 8)
    feat[:, :, :nfeat] = g("feat", (B, Npts, nfeat)) * mag
    feat[:, 7, 3] = mag * 1e6                                      # the hot pixel
    xyz = cloud_xyz("U", B, Npts, 47)
    ctr = xyz[:, :S].contiguous()
    gidx = torch.from_numpy(synth.hash_randint("gi", 0, Npts, (B, S, K), C3 + 2)).int()
    gidx[:, :, 1] = 7                                               # every group sees it
    W1f, b1, W1x = g("W1f", (C1, nfeat), 0.4 / mag), g("b1", (C1,), 0.1), g("W1x", (C1, 3), 0.5)
    W2, b2 = g("W2", (C2, C1), C1 ** -0.5), g("b2", (C2,), 0.1)
    W3, b3 = g("W3", (C3, C2), C2 ** -0.5), g("b3", (C3,), 0.1)
    bi = torch.arange(B).view(B, 1, 1)
    f_rows = feat[:, :, :nfeat].double()[bi, gidx.long()]
    dxyz = (xyz[bi, gidx.long()] - ctr.view(B, S, 1, 3)).double()
    h1 = (f_rows @ W1f.double().t() + b1.double() + dxyz @ W1x.double().t()).clamp_min(0)
    h2 = (h1 @ W2.double().t() + b2.double()).clamp_min(0)
    ref = (h2 @ W3.double().t() + b3.double()).clamp_min(0).max(2)[0]
    up = lambda x, m: (x + m - 1) // m * m
    W1x4 = torch.zeros(C1, 4); W1x4[:, :3] = W1x
    W2p = torch.zeros(up(C2, 32), C1); W2p[:C2] = W2
    b2p = torch.zeros(up(C2, 32)); b2p[:C2] = b2
    W3p = torch.zeros(C3, up(C2, 8)); W3p[:, :C2] = W3
    amax = ops.range_record(B, "cuda")
    amax.view(torch.float32).copy_(feat.abs().amax((1, 2)))
    dmax = float(dxyz.abs().max()) * 1.0001
    got = ops.sa_mlp_max(None, ops.pack_points(xyz.cuda()), ops.pack_points(ctr.cuda()), gidx.cuda(), W1x4.cuda(), W2p.cuda(), b2p.cuda(), W3p.cuda(),
                         b3.cuda(), C2, precision, feat=feat.cuda(), W1f=W1f.cuda(), b1=b1.cuda(), feat_amax=amax, dmax=dmax)
    err = rel(got, ref)
    print(f"sa<{C1},{C2},{C3}> {precision} layer 1 from raw features (|f| ~ {mag:g}, hot pixel x1e6): rel err {err:.2e}")
    assert torch.isfinite(got).all() and err < tol


@pytest.mark.parametrize("C1,C2,C3,K", SA_CASES)
@pytest.mark.parametrize("nfeat", [4, 5])
def test_sa_mlp_max_bf16_layer1_from_raw_features(C1, C2, C3, K, nfeat):
    """BF16: layer 1 on the matrix pipe straight from the raw feature rows (ev2h_sa_desc.feat; no layer-1 table), against the float64
    restatement of pointnet2_utils.py:244-257 on [features | relative xyz], and against the table form of the same kernel (the
    two differ only by the bf16 rounding of the layer-1 weights: inputs enter as two bf16 planes)."""
    _need_gpu()
    from ev2hands_amd import ops
    B, Npts, S = 2, 512, 37
    g = lambda n, s, sc=1.0: torch.from_numpy(synth.hash_normal(n, s, C1 + C2 + nfeat) * sc).float()
    feat = torch.zeros(B, Npts, 8)
    feat[:, :, :nfeat] = g("feat", (B, Npts, nfeat)) * torch.tensor([1.0, 1.0, 1.0, 3.0, 3.0][:nfeat])
    xyz = cloud_xyz("U", B, Npts, 47)
    ctr = xyz[:, :S].contiguous()
    gidx = torch.from_numpy(synth.hash_randint("gi", 0, Npts, (B, S, K), C3 + 2)).int()
    W1f, b1, W1x = g("W1f", (C1, nfeat), 0.4), g("b1", (C1,), 0.1), g("W1x", (C1, 3), 0.5)
    W2, b2 = g("W2", (C2, C1), C1 ** -0.5), g("b2", (C2,), 0.1)
    W3, b3 = g("W3", (C3, C2), C2 ** -0.5), g("b3", (C3,), 0.1)
    bi = torch.arange(B).view(B, 1, 1)
    f_rows = feat[:, :, :nfeat].double()[bi, gidx.long()]
    dxyz = (xyz[bi, gidx.long()] - ctr.view(B, S, 1, 3)).double()
    h1 = (f_rows @ W1f.double().t() + b1.double() + dxyz @ W1x.double().t()).clamp_min(0)
    h2 = (h1 @ W2.double().t() + b2.double()).clamp_min(0)
    ref = (h2 @ W3.double().t() + b3.double()).clamp_min(0).max(2)[0]
    up = lambda x, m: (x + m - 1) // m * m
    W1x4 = torch.zeros(C1, 4); W1x4[:, :3] = W1x
    W2p = torch.zeros(up(C2, 32), C1); W2p[:C2] = W2
    b2p = torch.zeros(up(C2, 32)); b2p[:C2] = b2
    W3p = torch.zeros(C3, up(C2, 8)); W3p[:, :C2] = W3
    pts4, ctr4 = ops.pack_points(xyz.cuda()), ops.pack_points(ctr.cuda())
    args = (pts4, ctr4, gidx.cuda(), W1x4.cuda(), W2p.cuda(), b2p.cuda(), W3p.cuda(), b3.cuda(), C2, "bf16")
    got = ops.sa_mlp_max(None, *args, feat=feat.cuda(), W1f=W1f.cuda(), b1=b1.cuda())
    P1 = (feat[:, :, :nfeat] @ W1f.t() + b1).cuda()
    table = ops.sa_mlp_max(P1, *args)
    e_ref, e_tab = rel(got, ref), rel(got, table)
    print(f"sa<{C1},{C2},{C3}> bf16 layer 1 from raw features (nfeat {nfeat}): vs float64 {e_ref:.2e}, vs the table form {e_tab:.2e}")
    assert e_ref < 2e-2 and e_tab < 2e-2


@pytest.mark.parametrize("mag", [1.0, 1e-3, 3e4])
@pytest.mark.parametrize("C1,C2,C3,K", SA_CASES)
def test_sa_mlp_max_bf16x3_layer1_from_raw_features(C1, C2, C3, K, mag):
    """[r5] BF16X3: layer 1 on the matrix pipe straight from the raw feature rows (three MFMAs per chunk = the six plane products, the
    bias as the accumulator's start, no factors), same construction as the F16X2 test -- features of magnitude `mag` next to O(1)
    coordinates and a 1e6 x hot pixel in every group -- fp32-class against the float64 restatement of pointnet2_utils.py:244-257,
    and against the table form of the same mode (exact fp32 layer 1)."""
    _need_gpu()
    from ev2hands_amd import ops
    B, Npts, S, nfeat = 2, 512, 37, 5
    g = lambda n, s, sc=1.0: torch.from_numpy(synth.hash_normal(n, s, C1 + C2 + 3) * sc).float()
    feat = torch.zeros(B, Npts, 8)
    feat[:, :, :nfeat] = g("feat", (B, Npts, nfeat)) * mag
    feat[:, 7, 3] = mag * 1e6                                      # the hot pixel
    xyz = cloud_xyz("U", B, Npts, 47)
    ctr = xyz[:, :S].contiguous()
    gidx = torch.from_numpy(synth.hash_randint("gi", 0, Npts, (B, S, K), C3 + 2)).int()
    gidx[:, :, 1] = 7                                               # every group sees it
    W1f, b1, W1x = g("W1f", (C1, nfeat), 0.4 / mag), g("b1", (C1,), 0.1), g("W1x", (C1, 3), 0.5)
    W2, b2 = g("W2", (C2, C1), C1 ** -0.5), g("b2", (C2,), 0.1)
    W3, b3 = g("W3", (C3, C2), C2 ** -0.5), g("b3", (C3,), 0.1)
    bi = torch.arange(B).view(B, 1, 1)
    f_rows = feat[:, :, :nfeat].double()[bi, gidx.long()]
    dxyz = (xyz[bi, gidx.long()] - ctr.view(B, S, 1, 3)).double()
    h1 = (f_rows @ W1f.double().t() + b1.double() + dxyz @ W1x.double().t()).clamp_min(0)
    h2 = (h1 @ W2.double().t() + b2.double()).clamp_min(0)
    ref = (h2 @ W3.double().t() + b3.double()).clamp_min(0).max(2)[0]
    up = lambda x, m: (x + m - 1) // m * m
    W1x4 = torch.zeros(C1, 4); W1x4[:, :3] = W1x
    W2p = torch.zeros(up(C2, 32), C1); W2p[:C2] = W2
    b2p = torch.zeros(up(C2, 32)); b2p[:C2] = b2
    W3p = torch.zeros(C3, up(C2, 8)); W3p[:, :C2] = W3
    args = (ops.pack_points(xyz.cuda()), ops.pack_points(ctr.cuda()), gidx.cuda(), W1x4.cuda(), W2p.cuda(), b2p.cuda(), W3p.cuda(), b3.cuda(), C2, "bf16x3")
    got = ops.sa_mlp_max(None, *args, feat=feat.cuda(), W1f=W1f.cuda(), b1=b1.cuda())
    P1 = (feat[:, :, :nfeat].double() @ W1f.double().t() + b1.double()).float().cuda()
    table = ops.sa_mlp_max(P1, *args)
    e_ref, e_tab = rel(got, ref), rel(got, table)
    print(f"sa<{C1},{C2},{C3}> bf16x3 layer 1 from raw features (|f| ~ {mag:g}, hot pixel x1e6): vs float64 {e_ref:.2e}, vs the table form {e_tab:.2e}")
    assert torch.isfinite(got).all() and e_ref < 8e-6 and e_tab < 8e-6


@pytest.mark.parametrize("precision", ["f16x2", "bf16x3", "bf16", "f16"])
@pytest.mark.parametrize("C1,C2,C3,K", [(64, 96, 128, 128), (128, 196, 256, 128), (128, 128, 256, 64)])
def test_sa_mlp_max_skips_padding_strips(C1, C2, C3, K, precision):
    """query_ball_point pads the slots past a group's neighbour count with slot 0 (pointnet2_utils.py:104-106); given the
    counts, the kernel skips 32-slot strips that hold only padding -- the output must not change by a single bit."""
    _need_gpu()
    from ev2hands_amd import ops
    B, Npts, S = 2, 512, 43
    g = lambda n, s, sc=1.0: torch.from_numpy(synth.hash_normal(n, s, C1 + C2 + 1) * sc).float()
    P1 = g("P1", (B, Npts, C1))
    xyz = cloud_xyz("U", B, Npts, 43)
    ctr = xyz[:, :S].contiguous()
    gidx = torch.from_numpy(synth.hash_randint("gi", 0, Npts, (B, S, K), C3 + 1)).int()
    cnt = torch.from_numpy(synth.hash_randint("cnt", 1, K + 1, (B, S), 5)).int()
    cnt[0, :8] = K                                   # one workgroup of full groups, one group with a single neighbour
    cnt[1, 3] = 1
    slot = torch.arange(K).view(1, 1, K)
    gidx = torch.where(slot < cnt.unsqueeze(-1), gidx, gidx[:, :, :1]).contiguous()
    up = lambda x, m: (x + m - 1) // m * m
    W1x4 = torch.zeros(C1, 4); W1x4[:, :3] = g("W1x", (C1, 3), 0.5)
    W2p = torch.zeros(up(C2, 32), C1); W2p[:C2] = g("W2", (C2, C1), C1 ** -0.5)
    b2p = torch.zeros(up(C2, 32)); b2p[:C2] = g("b2", (C2,), 0.1)
    W3p = torch.zeros(C3, up(C2, 8)); W3p[:, :C2] = g("W3", (C3, C2), C2 ** -0.5)
    b3 = g("b3", (C3,), 0.1)
    args = (P1.cuda(), ops.pack_points(xyz.cuda()), ops.pack_points(ctr.cuda()), gidx.cuda(), W1x4.cuda(), W2p.cuda(), b2p.cuda(),
            W3p.cuda(), b3.cuda(), C2, precision)
    full = ops.sa_mlp_max(*args)
    skipped = ops.sa_mlp_max(*args, cnt=cnt.cuda())
    assert torch.equal(full, skipped)


def test_attention():
    _need_gpu()
    from ev2hands_amd import ops
    from oracle import tehnet_oracle as O
    B, N = 2, 2048
    g = lambda n, s: torch.from_numpy(synth.hash_normal(n, s, 9)).float()
    key, value = g("k", (B, 4, N)), g("v", (B, 256, N))
    q = [g("qL", (B, 256, N)) * 0.3, g("qR", (B, 256, N)) * 0.3]
    ref = [O.attention(key, value, q[h]) for h in range(2)]
    logits_pm = key.permute(0, 2, 1).contiguous().cuda()
    query_pm = torch.stack([q[h].permute(0, 2, 1).contiguous() for h in range(2)]).cuda()
    value_pm = value.permute(0, 2, 1).contiguous().cuda()
    sim, hf8 = ops.attention(logits_pm, query_pm, value_pm)
    for h in range(2):
        got = hf8[h, :, :, :4].permute(0, 2, 1)
        assert rel(got, ref[h]) < 1e-5
        assert float(hf8[h, :, :, 4:].abs().max()) == 0.0


# the softmax arguments here are sums of N products of O(1) numbers (|s| up to ~10): an fp32 sum of 2048 terms carries ~1e-6 of
# absolute error, which the exponential turns into the same RELATIVE error of the map -- for either order of summation
TOL_SIM = 5e-6


@pytest.mark.parametrize("B,N", [(2, 2048), (3, 1000), (1, 130), (2, 257)])
def test_attention_sim_folded(B, N):
    """ev2h_attn_sim_folded (the last query Conv1d(k=3) -> BN folded behind the attention's sum over the points) against the
    reference's order of operations in float64: zero-padded F.conv1d over the points, bmm with the key, scale, softmax over the
    classes (TEHNet.py:13-22,155-156).  N not a multiple of the 256-point partial sums; a bias, so the sum-of-keys term matters."""
    _need_gpu()
    import torch.nn.functional as F
    from ev2hands_amd import ops
    g = lambda n, s, sc=1.0: torch.from_numpy(synth.hash_normal(n, s, 19) * sc).float()                   # noqa: E731
    key = g("k", (B, 4, N))
    q1 = g("q1", (B, 512, N))                                  # both hands' first-block outputs (signed: post-ReLU BN shifts them)
    W = [g(f"W{h}", (256, 256, 3), 0.05) for h in range(2)]    # [O, I, tap]
    bias = [g(f"b{h}", (256,), 0.2) for h in range(2)]
    ref = []
    for h in range(2):
        q2 = F.conv1d(q1[:, h * 256:(h + 1) * 256].double(), W[h].double(), bias[h].double(), padding=1)          # [B,256,N]
        ref.append(torch.softmax(torch.bmm(key.double(), q2.permute(0, 2, 1)) * 256 ** -0.5, dim=1))             # [B,4,256]
    Wt = [W[h].permute(0, 2, 1).reshape(256, 768).contiguous().cuda() for h in range(2)]                         # tap-major
    sim = ops.attention_sim_folded(key.permute(0, 2, 1).contiguous().cuda(), q1.permute(0, 2, 1).contiguous().cuda(), Wt,
                                   [b.cuda() for b in bias])
    errs = [rel(sim[:, h], ref[h].float()) for h in range(2)]
    print(f"folded sim B={B} N={N}: rel err {errs[0]:.2e} {errs[1]:.2e}")
    for h in range(2):
        assert errs[h] < TOL_SIM, errs[h]
    # and next to the unfolded kernels of this library on the same inputs (k=3 GEMM in exact fp32, then ev2h_attn_sim)
    q2g = torch.stack([ops.dense(q1.permute(0, 2, 1).reshape(B * N, 512)[:, h * 256:(h + 1) * 256].contiguous().cuda(), Wt[h], bias[h].cuda(),
                                 taps=3, rows_per_seq=N, K=256).view(B, N, 256) for h in range(2)])
    sim2, _ = ops.attention(key.permute(0, 2, 1).contiguous().cuda(), q2g.contiguous(), torch.zeros(B, N, 256, device="cuda"))
    print(f"   unfolded kernels vs float64: {rel(sim2[:, 0], ref[0].float()):.2e}; folded vs unfolded: {rel(sim, sim2):.2e}")
    assert rel(sim2[:, 0], ref[0].float()) < TOL_SIM and rel(sim, sim2) < 2 * TOL_SIM


@pytest.mark.parametrize("side", ["left", "right"])
def test_mano_layer(side):
    _need_gpu()
    from ev2hands_amd.mano import ManoHand
    from oracle import mano_oracle
    a = synth.synth_mano_assets(side, 7)
    B = 5
    prm = torch.from_numpy(synth.hash_normal("prm", (B, 22), 8) * 0.5).float()
    prm[:, 19:] *= 0.2
    args = (prm[:, :3], prm[:, 3:9], prm[:, 9:19], prm[:, 19:])
    ref = mano_oracle.ManoOracle(a)(*args)
    ref64 = mano_oracle.ManoOracle(a, dtype=torch.float64)(*[x.double() for x in args])
    hand = ManoHand(a, "cuda:0")
    got = hand(*[x.cuda() for x in args])
    assert rel(got.vertices, ref.vertices) < 1e-5 and rel(got.joints, ref.joints) < 1e-5
    assert float((got.vertices.cpu().double() - ref64.vertices).abs().max()) < 1e-5     # metres
    assert float((got.joints.cpu().double() - ref64.joints).abs().max()) < 1e-5


def test_mano_rotation_matrices_match_the_reference_formula():
    """The rotation matrices ev2h_mano builds (exposed by ev2h_mano_rotations) against tests/golden/rodrigues_0.npz, which holds
    outputs of the reference's own batch_rodrigues (/root/reference/src/Ev2Hands/losses.py:14-51, compiled from the reference
    file by oracle/make_golden_rodrigues.py).  fp32 sin / cos / sqrt / divide of the device differ from the host's libm in the last
    bits: 2e-6 absolute on matrix entries (entries are <= 1)."""
    _need_gpu()
    import os
    from ev2hands_amd import ops
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "rodrigues_0.npz"))
    th, R = torch.from_numpy(g["theta"]), torch.from_numpy(g["R"])
    M = th.shape[0]
    pad = (-M) % 16
    thp = torch.cat([th, torch.zeros(pad, 3)]).view(-1, 16, 3)
    got = ops.mano_rotations(thp.cuda()).cpu().view(-1, 3, 3)[:M]
    fin = torch.isfinite(R).all(-1).all(-1)
    assert torch.equal(torch.isfinite(got).all(-1).all(-1), fin)            # theta + 1e-8 == 0: NaN in the reference, NaN here
    err = float((got[fin] - R[fin]).abs().max())
    print("rotation max abs err", err)
    assert err < 2e-6
    # tiny angles: where the 1e-8 offset decides the axis the matrices must still be rotations
    eye_err = float((got[fin] @ got[fin].transpose(1, 2) - torch.eye(3)).abs().max())
    assert eye_err < 2e-6
