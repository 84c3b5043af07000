"""Operator-level host API over the C ABI, named after the reference functions they replace
(/root/reference/src/Ev2Hands/model/pointnet2_utils.py).  Tensors in, tensors out, all on the GPU;
every function calls straight into libev2hands_hip.so -- nothing here computes with torch ops.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _st():
    return _lib.stream_handle()


def pack_points(xyz: torch.Tensor) -> torch.Tensor:
    """[B,N,3] -> [B,N,4] with (x*x+y*y)+z*z in the 4th slot, via ev2h_prep_points."""
    B, N, _ = xyz.shape
    cm = torch.zeros(B, 4, N, device=xyz.device, dtype=torch.float32)
    cm[:, :3] = xyz.permute(0, 2, 1)
    pts4 = torch.empty(B, N, 4, device=xyz.device, dtype=torch.float32)
    feat8 = torch.empty(B, N, 8, device=xyz.device, dtype=torch.float32)
    _lib.check(_lib.lib().ev2h_prep_points(cm.data_ptr(), B, 4, N, 0, pts4.data_ptr(), feat8.data_ptr(), None, _st()), "prep")
    return pts4


def range_record(groups: int, device) -> torch.Tensor:
    """A zeroed F16X2 range record (uint32 bit patterns of max |value| per group, include/ev2hands_hip.h "Range records")."""
    return torch.zeros(groups, dtype=torch.int32, device=device)


def range_values(rec: torch.Tensor) -> torch.Tensor:
    """The maxima a range record holds, as float32."""
    return rec.view(torch.float32)


def farthest_point_sample(xyz: torch.Tensor, npoint: int, init: torch.Tensor | None = None) -> torch.Tensor:
    """pointnet2_utils.py:63-84.  xyz [B,N,3] (cuda) -> int64 [B,npoint]."""
    B, N, _ = xyz.shape
    if init is None:
        init = torch.randint(0, N, (B,), dtype=torch.long)
    init = init.to(xyz.device, torch.long).contiguous()
    pts4 = pack_points(xyz)
    idx = torch.empty(B, npoint, device=xyz.device, dtype=torch.int32)
    ctr = torch.empty(B, npoint, 4, device=xyz.device, dtype=torch.float32)
    _lib.check(_lib.lib().ev2h_fps(pts4.data_ptr(), B, N, npoint, init.data_ptr(), idx.data_ptr(), ctr.data_ptr(), _st()),
               "ev2h_fps")
    return idx.long()


def query_ball_point(radius, nsample, xyz: torch.Tensor, new_xyz: torch.Tensor, return_counts: bool = False):
    """pointnet2_utils.py:87-107.  radius / nsample may be scalars or equal-length lists (one pass)."""
    radii = list(radius) if isinstance(radius, (list, tuple)) else [radius]
    ks = list(nsample) if isinstance(nsample, (list, tuple)) else [nsample]
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    pts4, ctr4 = pack_points(xyz), pack_points(new_xyz)
    outs = [torch.empty(B, S, k, device=xyz.device, dtype=torch.int32) for k in ks]
    cnt = torch.empty(B, S, len(ks), device=xyz.device, dtype=torch.int32)
    n = len(ks)
    r_arr = (C.c_double * n)(*[float(r) for r in radii])
    k_arr = (C.c_int * n)(*ks)
    g_arr = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    _lib.check(_lib.lib().ev2h_ball_query(pts4.data_ptr(), ctr4.data_ptr(), B, N, S, n, r_arr, k_arr, g_arr, cnt.data_ptr(),
                                          _st()), "ev2h_ball_query")
    res = [o.long() for o in outs]
    if not isinstance(radius, (list, tuple)):
        res = res[0]
    return (res, cnt) if return_counts else res


def three_nn_interpolate(xyz1: torch.Tensor, xyz2: torch.Tensor, feat2: torch.Tensor, out_amax: torch.Tensor | None = None):
    """pointnet2_utils.py:296-303.  xyz1 [B,N,3], xyz2 [B,S,3], feat2 [B,S,D] -> (interp [B,N,D], idx, weight).
    out_amax: optional range record [B] of the interpolated rows."""
    B, N, _ = xyz1.shape
    S, D = xyz2.shape[1], feat2.shape[2]
    p1, p2 = pack_points(xyz1), pack_points(xyz2)
    f2 = feat2.contiguous()
    out = torch.empty(B, N, D, device=xyz1.device, dtype=torch.float32)
    idx = torch.empty(B, N, 3, device=xyz1.device, dtype=torch.int32)
    w = torch.empty(B, N, 3, device=xyz1.device, dtype=torch.float32)
    _lib.check(_lib.lib().ev2h_three_nn_interp(p1.data_ptr(), p2.data_ptr(), B, N, S, f2.data_ptr(), D, D, out.data_ptr(), D,
                                               idx.data_ptr(), w.data_ptr(), _lib.ptr(out_amax), _st()), "ev2h_three_nn_interp")
    return out, idx.long(), w


def dense(X: torch.Tensor, W: torch.Tensor, bias=None, relu=False, post_scale=None, post_shift=None, taps=1,
          rows_per_seq=0, rowmax_rows=0, bias_group_rows=0, K=None, precision: str = "f32", presplit: bool = True, w_image=None, w_tile_rows: int = 128,
          x_amax=None, x_amax2=None, x_group_rows=0, y_amax=None, y_group_rows=0, y_scale=None, y_bound_w=0.0, y_bound_b=0.0,
          skinny: bool = False) -> torch.Tensor:
    """Y = post(relu(X W^T + b)); X [M,ldx], W [N,ldw] (ev2h_gemm).  K defaults to X.shape[1].
    x_amax .. y_bound_b: the F16X2 range arguments of ev2h_gemm_desc (range_record tensors)."""
    M, ldx = X.shape
    N, ldw = W.shape
    K = (ldx if K is None else K)
    rows_out = M // rowmax_rows if rowmax_rows else M
    Y = torch.empty(rows_out, N, device=X.device, dtype=torch.float32)
    d = _lib.GemmDesc()
    d.X, d.ldx, d.W, d.ldw, d.Y, d.ldy = X.data_ptr(), ldx, W.data_ptr(), ldw, Y.data_ptr(), N
    d.M, d.N, d.K = M, N, K
    d.bias = _lib.ptr(bias)
    d.bias_group_rows = bias_group_rows
    d.ldbias = bias.shape[-1] if (bias is not None and bias_group_rows) else 0
    d.relu = int(relu)
    d.post_scale, d.post_shift = _lib.ptr(post_scale), _lib.ptr(post_shift)
    d.taps, d.rows_per_seq, d.rowmax_rows = taps, rows_per_seq, rowmax_rows
    d.precision = _lib.PREC[precision]
    d.x_amax, d.x_amax2, d.x_group_rows = _lib.ptr(x_amax), _lib.ptr(x_amax2), x_group_rows
    d.y_amax, d.y_group_rows = _lib.ptr(y_amax), y_group_rows
    d.y_scale, d.y_bound_w, d.y_bound_b = _lib.ptr(y_scale), y_bound_w, y_bound_b
    d.skinny = int(skinny)
    keep = None
    d.ws_tile_rows = w_tile_rows
    if w_image is not None:
        keep, d.w_unscale = w_image
        d.Ws = keep.data_ptr()
    elif precision != "f32" and N >= 96 and presplit:
        keep, d.w_unscale = make_w_image(W, precision, w_tile_rows)
        d.Ws = keep.data_ptr()
    elif precision != "f32":
        from .pack import NS_OF, plane_unscale
        d.w_unscale = plane_unscale(W.detach().cpu().double().numpy(), NS_OF[precision])
    L = _lib.lib()
    _lib.check(L.ev2h_init(), "ev2h_init")
    _lib.check(L.ev2h_gemm(C.byref(d), _st()), "ev2h_gemm")
    return Y


def make_w_image(W: torch.Tensor, precision: str, rows: int = 128):
    """bf16 plane images of a dense weight for the bf16 GEMM kernels (host-side packing, do it once per weight)."""
    from .pack import NS_OF, gemm_bf16_w_image
    img, u = gemm_bf16_w_image(W.detach().cpu().double().numpy(), NS_OF[precision], rows)
    return torch.from_numpy(img).to(W.device), u          # (device image, power-of-two unscale)


def sa_mlp_max(P1, pts4, ctr4, gidx, W1x, W2, b2, W3, b3, C2: int, precision: str = "f32", cnt=None,
               p1_scale=None, p1_amax=None, dmax: float = 0.0, out_amax=None, feat=None, W1f=None, b1=None, feat_amax=None) -> torch.Tensor:
    """Fused grouped MLP + max (ev2h_sa_mlp_max).  P1 [B,Npts,C1], gidx [B,S,K] int32 -> [B,S,C3].
    W2 [roundup(C2,32), C1], W3 [C3, roundup(C2,8)] fp32 (padded); 16-bit tile images are built here when needed.
    cnt [B,S] int32 (optional): distinct neighbours per group (query_ball_point's count); padding strips are skipped.
    p1_scale [B] float32 / p1_amax [B] range record / dmax / out_amax: the F16X2 range arguments of ev2h_sa_desc (P1 then
    holds p1_scale[b] * table)."""
    if feat is not None and P1 is None and precision not in ("bf16", "f16x2", "bf16x3", "f16"):
        raise ValueError(f"sa_mlp_max: feature rows without a layer-1 table need a plane precision ('bf16', 'f16x2', 'bf16x3', 'f16'), not {precision!r}")
    if feat is not None:
        # "bf16" / "f16x2" / "bf16x3": layer 1 from the raw feature rows feat [B,Npts,8] (first W1f.shape[1] <= 5 columns used) with W1f [C1,nfeat],
        # b1 [C1] -- no table (P1 may be None); feat_amax [B] range record of the rows + dmax: F16X2 range handling
        B, Npts, C1 = feat.shape[0], feat.shape[1], W1x.shape[0]
        dev = feat.device
    else:
        B, Npts, C1 = P1.shape
        dev = P1.device
    S, K = gidx.shape[1], gidx.shape[2]
    C3 = W3.shape[0]
    out = torch.empty(B, S, C3, device=dev, dtype=torch.float32)
    d = _lib.SaDesc()
    d.P1, d.ldp, d.pts4, d.ctr4, d.gidx = _lib.ptr(P1), C1, pts4.data_ptr(), ctr4.data_ptr(), gidx.data_ptr()
    if feat is not None:
        W1fc, b1c = W1f.contiguous(), b1.contiguous()
        d.feat, d.ldf, d.W1f, d.ldw1f, d.b1, d.nfeat = feat.data_ptr(), feat.shape[2], W1fc.data_ptr(), W1fc.shape[1], b1c.data_ptr(), W1fc.shape[1]
        if precision in ("f16x2", "f16"):
            from .pack import NS_OF, plane_unscale
            d.w1f_unscale = plane_unscale(W1fc.detach().cpu().double().numpy(), NS_OF[precision])
            d.w1x_unscale = plane_unscale(W1x[:, :3].detach().cpu().double().numpy(), NS_OF[precision])
            if feat_amax is not None:            # F16X2 range handling of the feature mode (ev2h_sa_desc.feat_amax)
                d.feat_amax, d.dmax = feat_amax.data_ptr(), dmax
                d.w1f_norm, d.b1_max = float(W1fc.abs().sum(1).max()) * (1 + 1e-6), float(b1c.abs().max()) * (1 + 1e-6)
                d.w1x_norm = float(W1x[:, :3].abs().sum(1).max()) * (1 + 1e-6)
                d.w2_norm = float(W2[:C2].abs().sum(1).max()) * (1 + 1e-6)
                d.b2_max = float(b2[:C2].abs().max()) * (1 + 1e-6)
    d.W1x, d.W2, d.b2, d.W3, d.b3 = W1x.data_ptr(), W2.data_ptr(), b2.data_ptr(), W3.data_ptr(), b3.data_ptr()
    d.out, d.ldo = out.data_ptr(), C3
    d.B, d.Npts, d.S, d.K, d.C1, d.C2, d.C3 = B, Npts, S, K, C1, C2, C3
    d.precision = _lib.PREC[precision]
    if cnt is not None:
        assert cnt.dtype == torch.int32 and cnt.is_contiguous() and cnt.shape == (B, S)
        d.cnt, d.cnt_ld = cnt.data_ptr(), 1
    d.out_amax = _lib.ptr(out_amax)
    if p1_scale is not None:
        d.p1_scale, d.p1_amax, d.dmax = p1_scale.data_ptr(), p1_amax.data_ptr(), dmax
        d.w1x_norm = float(W1x[:, :3].abs().sum(1).max())
        d.w2_norm = float(W2[:C2].abs().sum(1).max())
        d.b2_max = float(b2[:C2].abs().max())
    keep = []
    if precision != "f32":
        from .pack import NS_OF, sa_bf16_images
        i2, i3, d.w2_unscale, d.w3_unscale = sa_bf16_images(W2[:C2].detach().cpu().double().numpy(),
                                                            W3[:, :C2].detach().cpu().double().numpy(), NS_OF[precision])
        keep = [torch.from_numpy(i2).to(dev), torch.from_numpy(i3).to(dev)]
        d.W2s, d.W3s = keep[0].data_ptr(), keep[1].data_ptr()
    _lib.check(_lib.lib().ev2h_sa_mlp_max(C.byref(d), _st()), "ev2h_sa_mlp_max")
    return out


def feature_propagation(xyz1: torch.Tensor, xyz2: torch.Tensor, points2: torch.Tensor, W1, b1, W2, b2, W3, b3, precision: str = "f16x2",
                        ranges: bool = False, x_amax=None, out_amax=None):
    """PointNetFeaturePropagation.forward with points1 = None and a three-layer MLP (pointnet2_utils.py:280-316, fp1 of
    TEHNet.py:129,186) in the fused form of the 16-bit path: 3-NN search (ev2h_three_nn_interp without output), layer-1 table of the
    S coarse points (ev2h_gemm), blend + layers 2-3 (ev2h_fp_mlp).  xyz1 [B,N,3], xyz2 [B,S,3], points2 [B,S,D]; W*, b* the folded
    fp32 weights (128-128-256).  ranges=True: F16X2 range handling (x_amax = record [B] of points2, out_amax = record of the output).
    Returns (out [B,N,C3], idx [B,N,3], weight [B,N,3])."""
    from .pack import NS_OF, sa_bf16_images
    B, N, _ = xyz1.shape
    S, D = xyz2.shape[1], points2.shape[2]
    dev = xyz1.device
    C1, C2, C3 = W1.shape[0], W2.shape[0], W3.shape[0]
    p1, p2 = pack_points(xyz1), pack_points(xyz2)
    idx = torch.empty(B, N, 3, device=dev, dtype=torch.int32)
    w = torch.empty(B, N, 3, device=dev, dtype=torch.float32)
    L = _lib.lib()
    _lib.check(L.ev2h_three_nn_interp(p1.data_ptr(), p2.data_ptr(), B, N, S, None, 0, 0, None, 0, idx.data_ptr(), w.data_ptr(), None, _st()),
               "ev2h_three_nn_interp")
    t_scale = t_amax = None
    if ranges:
        t_scale = torch.empty(B, device=dev, dtype=torch.float32)
        t_amax = range_record(B, dev)
    T = dense(points2.reshape(B * S, D).contiguous(), W1.contiguous(), b1, precision=precision,
              x_amax=x_amax if ranges else None, x_group_rows=S if ranges else 0, y_amax=t_amax, y_group_rows=S if ranges else 0,
              y_scale=t_scale, y_bound_w=float(W1.abs().sum(1).max()) * (1 + 1e-6), y_bound_b=float(b1.abs().max()) * (1 + 1e-6))
    out = torch.empty(B, N, C3, device=dev, dtype=torch.float32)
    i2, i3, u2, u3 = sa_bf16_images(W2.detach().cpu().double().numpy(), W3.detach().cpu().double().numpy(), NS_OF[precision])
    keep = [torch.from_numpy(i2).to(dev), torch.from_numpy(i3).to(dev), b2.contiguous(), b3.contiguous()]
    d = _lib.FpDesc()
    d.T, d.ldt, d.nn_idx, d.nn_w = T.data_ptr(), C1, idx.data_ptr(), w.data_ptr()
    d.b2, d.b3, d.W2s, d.W3s, d.w2_unscale, d.w3_unscale = keep[2].data_ptr(), keep[3].data_ptr(), keep[0].data_ptr(), keep[1].data_ptr(), u2, u3
    d.out, d.ldo, d.B, d.N, d.S, d.C1, d.C2, d.C3 = out.data_ptr(), C3, B, N, S, C1, C2, C3
    d.precision = _lib.PREC[precision]
    if ranges:
        d.t_scale, d.t_amax = t_scale.data_ptr(), t_amax.data_ptr()
        d.w2_norm, d.b2_max = float(W2.abs().sum(1).max()) * (1 + 1e-6), float(b2.abs().max()) * (1 + 1e-6)
        d.out_amax = _lib.ptr(out_amax)
    _lib.check(L.ev2h_fp_mlp(C.byref(d), _st()), "ev2h_fp_mlp")
    return out, idx.long(), w


def row_chain(X: torch.Tensor, W2, b2, W3, b3, precision: str = "f16x2", relu_out: bool = False, x_amax=None, out_amax=None):
    """Two dense layers on plain rows in one kernel (ev2h_fp_mlp form (b); the segmentation head TEHNet.py:135-141 with its BN
    folded into W3 / b3 by the caller): out = W3 relu(W2 X + b2) + b3.  X [B,N,256], W2 [256,256], W3 [n<=32, 256].
    x_amax: F16X2 range record [B] of X (None = no range handling).  Returns (out [B,N,n], the same channel-major [B,n,N])."""
    from .pack import NS_OF, _pad, sa_bf16_images
    B, N, C1 = X.shape
    C2, n = W2.shape[0], W3.shape[0]
    dev = X.device
    W3p = _pad(W3.detach().cpu().double().numpy(), 32, C2)
    i2, i3, u2, u3 = sa_bf16_images(W2.detach().cpu().double().numpy(), W3p, NS_OF[precision])
    b3p = torch.zeros(32, device=dev, dtype=torch.float32)
    b3p[:n] = b3
    keep = [torch.from_numpy(i2).to(dev), torch.from_numpy(i3).to(dev), b2.contiguous(), b3p]
    out = torch.empty(B, N, n, device=dev, dtype=torch.float32)
    out_cm = torch.empty(B, n, N, device=dev, dtype=torch.float32)
    Xc = X.contiguous()
    d = _lib.FpDesc()
    d.T, d.ldt = Xc.data_ptr(), C1
    d.b2, d.b3, d.W2s, d.W3s, d.w2_unscale, d.w3_unscale = keep[2].data_ptr(), keep[3].data_ptr(), keep[0].data_ptr(), keep[1].data_ptr(), u2, u3
    d.out, d.ldo, d.out_cols, d.no_relu_out, d.out_cm = out.data_ptr(), n, n, int(not relu_out), out_cm.data_ptr()
    d.B, d.N, d.C1, d.C2, d.C3 = B, N, C1, C2, 32
    d.precision = _lib.PREC[precision]
    if x_amax is not None:
        d.t_amax = x_amax.data_ptr()
        d.w2_norm, d.b2_max = float(W2.abs().sum(1).max()) * (1 + 1e-6), float(b2.abs().max()) * (1 + 1e-6)
    d.out_amax = _lib.ptr(out_amax)
    _lib.check(_lib.lib().ev2h_fp_mlp(C.byref(d), _st()), "ev2h_fp_mlp")
    return out, out_cm


def attention(logits_pm: torch.Tensor, query_pm: torch.Tensor, value_pm: torch.Tensor, hf_amax: torch.Tensor | None = None,
              value_unscale: torch.Tensor | None = None):
    """TEHNet.py:13-27 for both hands.  logits_pm [B,N,4], query_pm [2,B,N,256], value_pm [B,N,256]
    -> (sim [B,2,4,256], hf8 [2,B,N,8]).  hf_amax: optional range records [2,B] of the context features;
    value_unscale: optional [256] per-channel factors of value (ev2h_attn_context)."""
    B, N, _ = logits_pm.shape
    sim = torch.empty(B, 2, 4, 256, device=logits_pm.device, dtype=torch.float32)
    hf8 = torch.empty(2, B, N, 8, device=logits_pm.device, dtype=torch.float32)
    L = _lib.lib()
    _lib.check(L.ev2h_attn_sim(logits_pm.data_ptr(), query_pm.data_ptr(), 256, B * N * 256, B, N, sim.data_ptr(), _st()), "sim")
    _lib.check(L.ev2h_attn_context(sim.data_ptr(), value_pm.data_ptr(), 256, B, N, hf8.data_ptr(), _lib.ptr(hf_amax), B, _lib.ptr(value_unscale), _st()), "ctx")
    return sim, hf8


def attention_sim_folded(logits_pm: torch.Tensor, q1_pm: torch.Tensor, W4, b4) -> torch.Tensor:
    """sim [B,2,4,256] of TEHNet.py:20-22 from the FIRST query block's output: the last Conv1d(k=3) -> BN of each hand is folded
    behind the sum over the points (ev2h_attn_sim_folded).  logits_pm [B,N,4], q1_pm [B,N,512] (hand h at columns h*256..),
    W4 = (left, right) folded weights [256, 768] tap-major, b4 = (left, right) [256]."""
    B, N, _ = logits_pm.shape
    dev = logits_pm.device
    L = _lib.lib()
    wt = [w.t().contiguous() for w in W4]
    bb = [b.contiguous() for b in b4]
    scratch = torch.empty(L.ev2h_attn_sim_folded_scratch(B, N), device=dev, dtype=torch.float32)
    sim = torch.empty(B, 2, 4, 256, device=dev, dtype=torch.float32)
    lg, q1 = logits_pm.contiguous(), q1_pm.contiguous()
    _lib.check(L.ev2h_attn_sim_folded(lg.data_ptr(), q1.data_ptr(), 512, B, N, wt[0].data_ptr(), wt[1].data_ptr(), bb[0].data_ptr(),
                                      bb[1].data_ptr(), scratch.data_ptr(), sim.data_ptr(), _st()), "ev2h_attn_sim_folded")
    return sim


def mano_rotations(theta: torch.Tensor) -> torch.Tensor:
    """Axis-angle vectors [M,16,3] -> the rotation matrices [M,16,3,3] the MANO kernel computes for them (ev2h_mano_rotations with
    an identity pose basis and a zero mean pose, so that the full pose IS theta).  Parity hook for losses.py:14-51."""
    M = theta.shape[0]
    dev = theta.device
    prm = theta.reshape(M, 48).to(torch.float32).contiguous()
    mean = torch.zeros(45, device=dev, dtype=torch.float32)
    comps = torch.eye(45, device=dev, dtype=torch.float32).contiguous()
    c = _lib.ManoConsts()
    c.hands_mean, c.comps, c.ncomps = mean.data_ptr(), comps.data_ptr(), 45
    rot = torch.empty(M, 16, 3, 3, device=dev, dtype=torch.float32)
    _lib.check(_lib.lib().ev2h_mano_rotations(C.byref(c), prm.data_ptr(), 48, M, rot.data_ptr(), _st()), "ev2h_mano_rotations")
    return rot
