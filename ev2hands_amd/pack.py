"""Checkpoint -> packed device weights for libev2hands_hip.so (host-side weight loading).

Eval-mode BatchNorm is folded in float64 and rounded once to fp32:
  * Conv -> BN -> ReLU blocks (set abstraction / feature propagation,
    /root/reference/src/Ev2Hands/model/pointnet2_utils.py:198,256,314) fold into W, b;
  * Conv/Linear -> ReLU -> BN blocks (classifier TEHNet.py:135-141, FC head :49-55, first query conv
    :150-153) keep the BN as an explicit post-ReLU scale/shift -- folding it forward is not exact
    for the zero-padded k=3 convolution;
  * the second query conv (Conv -> BN, TEHNet.py:155-156) folds exactly.
Layouts follow include/ev2hands_hip.h: layer-1 feature weights of all radius branches stacked
(one table GEMM per module), W2 rows padded to 32, W3 columns padded to 8, group-all inputs
re-ordered to [features | xyz | pad], k=3 conv weights tap-major.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, synth

BN_EPS = 1e-5


def _np(t) -> np.ndarray:
    return t.detach().cpu().double().numpy()


def _bn_affine(sd, p):
    alpha = _np(sd[p + ".weight"]) / np.sqrt(_np(sd[p + ".running_var"]) + BN_EPS)
    beta = _np(sd[p + ".bias"]) - _np(sd[p + ".running_mean"]) * alpha
    return alpha, beta


def _fold(sd, pc, pb):
    """Conv -> BN: returns (W' [O, I...], b' [O]) in float64."""
    W, b = _np(sd[pc + ".weight"]), _np(sd[pc + ".bias"])
    alpha, beta = _bn_affine(sd, pb)
    return W * alpha.reshape((-1,) + (1,) * (W.ndim - 1)), alpha * b + beta


def _pad(a: np.ndarray, rows: int | None = None, cols: int | None = None) -> np.ndarray:
    r = a.shape[0] if rows is None else rows
    if a.ndim == 1:
        out = np.zeros((r,), dtype=a.dtype)
        out[:a.shape[0]] = a
        return out
    c = a.shape[1] if cols is None else cols
    out = np.zeros((r, c), dtype=a.dtype)
    out[:a.shape[0], :a.shape[1]] = a
    return out


def _up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


NS_OF = {"f32": 0, "bf16": 1, "f16x2": 2, "bf16x3": 3}     # operand planes per precision (csrc/planes.hpp)


def split_bf16_planes(a: np.ndarray, ns: int):
    """fp32 array -> list of `ns` uint16 arrays of 16-bit plane patterns, the same splits the kernels apply to
    activations (csrc/planes.hpp: split_planes).
    ns == 1: bf16, round to nearest even.  ns == 2: two fp16 planes, a = h + l with h = rne(a), l = rne(a - h).
    ns == 3: exact truncation split into three bf16 planes a = h + m + l (8 + 8 + 8 mantissa bits)."""
    x = np.ascontiguousarray(a, dtype=np.float32)
    if ns == 2:
        if x.size and float(np.abs(x).max()) >= 65504.0:
            raise ValueError("f16x2 needs |weight| < 65504 (fp16 range); use precision='bf16x3' or 'f32' for this checkpoint")
        h = x.astype(np.float16)
        l = (x - h.astype(np.float32)).astype(np.float16)
        return [h.view(np.uint16), l.view(np.uint16)]
    if ns == 1:
        u = x.view(np.uint32).astype(np.uint64)
        return [((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)]
    planes, r = [], x.copy()
    for _ in range(ns):
        u = r.view(np.uint32) & np.uint32(0xFFFF0000)
        planes.append((u >> np.uint32(16)).astype(np.uint16))
        r = (r - u.view(np.float32)).astype(np.float32)        # exact: the difference has fewer significant bits
    return planes


def plane_unscale(W: np.ndarray, ns: int) -> float:
    """Power of two u such that the planes are taken of W / u.  f16x2 only: fp16 has 5 exponent bits, so the low plane of a
    weight below 2^-3 is subnormal and small-magnitude layers lose accuracy (measured 2.6e-4 at |W| ~ 1e-4).  Dividing by
    u = 2^-k with max|W / u| in [2^13, 2^14) is exact and the kernels multiply the accumulated product by u (also exact)."""
    if ns != 2:
        return 1.0
    m = float(np.abs(np.asarray(W, dtype=np.float64)).max()) if np.size(W) else 0.0
    if m == 0.0 or not np.isfinite(m):
        return 1.0
    k = int(np.floor(np.log2(16384.0 / m)))
    k = max(-24, min(k, 60))
    return float(2.0 ** -k)


def sa_bf16_geometry(C2: int):
    T2 = _up(C2, 32) // 32
    rem = C2 % 32
    m_last = 2 if rem == 0 else (1 if rem <= 16 else 2)
    return T2, 32 * (T2 - 1) + 16 * m_last


def sa_bf16_images(W2: np.ndarray, W3: np.ndarray, ns: int):
    """Byte images of the LDS weight tiles of sa_mlp_max_bf16_kernel (see SaBCfg in csrc/sa_mlp_bf16.hip).
    W2 [C2, C1], W3 [C3, C2] folded fp32 weights.  Returns (W2s, W3s, u2, u3): uint8 images of W2 / u2 and W3 / u3 and the
    power-of-two factors (plane_unscale) the kernel multiplies back."""
    u2, u3 = plane_unscale(W2, ns), plane_unscale(W3, ns)
    W2 = np.asarray(W2, dtype=np.float64) / u2
    W3 = np.asarray(W3, dtype=np.float64) / u3
    C2, C1 = W2.shape
    C3 = W3.shape[0]
    T2, C2P = sa_bf16_geometry(C2)
    W2p = _pad(W2, T2 * 32, C1)
    p2 = split_bf16_planes(W2p, ns)
    rs2 = ns * 64 + 16
    img2 = np.zeros((C1 // 32, T2 * 32, rs2), dtype=np.uint8)
    for c in range(C1 // 32):
        for s_ in range(ns):
            blk = np.ascontiguousarray(p2[s_][:, 32 * c:32 * c + 32])               # [rows, 32] uint16
            img2[c, :, s_ * 64:(s_ + 1) * 64] = blk.view(np.uint8).reshape(T2 * 32, 64)
    # layer-3 contraction order follows the MFMA D layout of layer 2: position 32t+16m+8h+e <-> channel 32t+16m+4h+(e&3)+8(e>>2)
    pos = np.arange(C2P)
    t, w_ = pos // 32, pos % 32
    m, h, e = w_ // 16, (w_ % 16) // 8, w_ % 8
    ch = 32 * t + 16 * m + 4 * h + (e & 3) + 8 * (e >> 2)
    W3p = np.zeros((C3, C2P), dtype=np.float64)
    ok = ch < C2
    W3p[:, ok] = W3[:, ch[ok]]
    p3 = split_bf16_planes(W3p, ns)
    rs3 = ns * C2P * 2 + 16
    img3 = np.zeros((C3 // 32, 32, rs3), dtype=np.uint8)
    for s_ in range(ns):
        blk = p3[s_].view(np.uint8).reshape(C3 // 32, 32, C2P * 2)
        img3[:, :, s_ * C2P * 2:(s_ + 1) * C2P * 2] = blk
    return img2.reshape(-1), img3.reshape(-1), u2, u3


GEMM_W_TILE_ROWS = 128     # rows per W image tile (128: occupancy kernel, 256: wide kernel)


def gemm_bf16_w_image(W: np.ndarray, ns: int, rows: int = GEMM_W_TILE_ROWS):
    """Plane images of a dense weight W [N, Ktot] for the 16-bit GEMM kernels: for every `rows`-row N tile and every 32-wide
    K tile one LDS tile image [rows][ns*64 + 16 bytes] (planes side by side, 16 B row pad).  Returns (image, u): the planes
    are those of W / u (plane_unscale)."""
    u = plane_unscale(W, ns)
    W = np.asarray(W, dtype=np.float64) / u
    N, K = W.shape
    tn, nk = _up(N, rows) // rows, _up(K, 32) // 32
    Wp = _pad(W, tn * rows, nk * 32)
    planes = split_bf16_planes(Wp, ns)
    rs = ns * 64 + 16
    img = np.zeros((tn, nk, rows, rs), dtype=np.uint8)
    for s_ in range(ns):
        blk = planes[s_].reshape(tn, rows, nk, 32).transpose(0, 2, 1, 3)         # [tn, nk, rows, 32] uint16
        img[:, :, :, s_ * 64:(s_ + 1) * 64] = np.ascontiguousarray(blk).view(np.uint8).reshape(tn, nk, rows, 64)
    return img.reshape(-1), u


class PackedWeights:
    """Owns the device tensors and the ev2h_weights struct that points at them."""

    def __init__(self, sd: dict, device, in_channels: int, precision: str = "f32"):
        self.device = torch.device(device)
        self.in_channels = in_channels
        self.precision = precision
        self.ns = NS_OF[precision]
        self._keep = []
        self.tensors = {}
        self.struct = _lib.Weights()
        w = self.struct
        w.precision = _lib.PREC[precision]
        self._sa_module(w.sa1, sd, "sa1", in_channels, 8, synth.SA1_NPOINT, synth.SA1_RADII, synth.SA1_NSAMPLE)
        self._sa_module(w.sa2, sd, "sa2", 320, 320, synth.SA2_NPOINT, synth.SA2_RADII, synth.SA2_NSAMPLE)
        for h, side in enumerate(("left", "right")):
            p = f"{side}_mano_regressor"
            self._sa_module(w.mano_sa1[h], sd, p + ".sa1", 4, 8, synth.MANO_SA1_NPOINT, synth.MANO_SA1_RADII,
                            synth.MANO_SA1_NSAMPLE)
            self._group_all(w.mano_sa2[h], sd, p + ".sa2", 2)
            W, b = _np(sd[p + ".mano_regressor.0.weight"]), _np(sd[p + ".mano_regressor.0.bias"])
            a, be = _bn_affine(sd, p + ".mano_regressor.2")
            self._dense(w.head0[h], p + ".head0", W, b, a, be)
            self._dense(w.head4[h], p + ".head4", _np(sd[p + ".mano_regressor.4.weight"]),
                        _np(sd[p + ".mano_regressor.4.bias"]))
        self._group_all(w.sa3, sd, "sa3", 3)
        # fp3: 1536 = 512 skip (l2_points) + 1024 broadcast (l3_points), pointnet2_utils.py:293-294,307
        W, b = _fold(sd, "fp3.mlp_convs.0", "fp3.mlp_bns.0")
        W = W[:, :, 0]
        self._dense(w.fp3_skip, "fp3.skip", W[:, :512], None)
        self._dense(w.fp3_bcast, "fp3.bcast", W[:, 512:], b)
        W, b = _fold(sd, "fp3.mlp_convs.1", "fp3.mlp_bns.1")
        self._dense(w.fp3_1, "fp3.1", W[:, :, 0], b)
        for k in range(2):
            W, b = _fold(sd, f"fp2.mlp_convs.{k}", f"fp2.mlp_bns.{k}")
            self._dense(w.fp2[k], f"fp2.{k}", W[:, :, 0], b)
        for k in range(3):
            W, b = _fold(sd, f"fp1.mlp_convs.{k}", f"fp1.mlp_bns.{k}")
            self._dense(w.fp1[k], f"fp1.{k}", W[:, :, 0], b)
        if self.ns:
            self._fp_module(w.fp1m, sd, "fp1")
        a, be = _bn_affine(sd, "classifier.2")
        self._dense(w.cls0, "cls0", _np(sd["classifier.0.weight"])[:, :, 0], _np(sd["classifier.0.bias"]), a, be)
        self._dense(w.cls4, "cls4", _np(sd["classifier.4.weight"])[:, :, 0], _np(sd["classifier.4.bias"]))
        if self.ns:
            self._cls_module(w.clsm, _np(sd["classifier.0.weight"])[:, :, 0], _np(sd["classifier.0.bias"]), a, be,
                             _np(sd["classifier.4.weight"])[:, :, 0], _np(sd["classifier.4.bias"]))
        # query convs: tap-major [O][3*I]; both hands' first conv stacked along O
        W0, b0, a0, be0 = [], [], [], []
        for h, side in enumerate(("left", "right")):
            p = f"{side}_query_conv"
            W = _np(sd[p + ".0.weight"])                                  # [O, I, 3]
            W0.append(np.ascontiguousarray(W.transpose(0, 2, 1)).reshape(W.shape[0], -1))
            b0.append(_np(sd[p + ".0.bias"]))
            a, be = _bn_affine(sd, p + ".2")
            a0.append(a)
            be0.append(be)
            W4, b4 = _fold(sd, p + ".4", p + ".5")
            W4 = np.ascontiguousarray(W4.transpose(0, 2, 1)).reshape(W4.shape[0], -1)
            self._dense(w.qconv4[h], p + ".4", W4, b4, K=256)
            w.qconv4T[h] = self._dev(p + ".4.WT", np.ascontiguousarray(W4.T))        # [768, 256]: ev2h_attn_sim_folded
        self._dense(w.qconv0, "qconv0", np.concatenate(W0, 0), np.concatenate(b0), np.concatenate(a0),
                    np.concatenate(be0), K=256)

    # ------------------------------------------------------------------ helpers
    def _dev(self, name: str, a: np.ndarray) -> int:
        t = torch.from_numpy(np.ascontiguousarray(a.astype(np.float32))).to(self.device)
        self._keep.append(t)
        self.tensors[name] = t
        return t.data_ptr()

    def _dev_bytes(self, name: str, a: np.ndarray) -> int:
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint8)).to(self.device)
        self._keep.append(t)
        self.tensors[name] = t
        return t.data_ptr()

    def _dense(self, d, name, W, b, post_scale=None, post_shift=None, K=None):
        O, Kfull = W.shape
        ldw = _up(Kfull, 4)
        d.W = self._dev(name + ".W", _pad(W, O, ldw))
        d.b = self._dev(name + ".b", b) if b is not None else None
        d.post_scale = self._dev(name + ".ps", post_scale) if post_scale is not None else None
        d.post_shift = self._dev(name + ".pt", post_shift) if post_shift is not None else None
        d.O, d.K, d.ldw = O, (ldw if K is None else K), ldw
        if self.ns and O >= 96:       # all but the tiny heads: pre-split W images, streamed by LDS-DMA
            img, d.w_unscale = gemm_bf16_w_image(_pad(W, O, ldw), self.ns)
            d.Ws = self._dev_bytes(name + ".Ws", img)
            d.ws_tile_rows = GEMM_W_TILE_ROWS
        else:
            d.w_unscale = plane_unscale(W, self.ns)        # the kernel splits W / w_unscale on the fly

    def _group_all(self, arr, sd, prefix, nlayers):
        """sample_and_group_all concatenates [xyz(3), features(512)] (pointnet2_utils.py:155); our
        buffers hold [features(512) | xyz(3) | 0 x 5] so that K = 520 is a multiple of 8."""
        for k in range(nlayers):
            W, b = _fold(sd, f"{prefix}.mlp_convs.{k}", f"{prefix}.mlp_bns.{k}")
            W = W[:, :, 0, 0]
            if k == 0:
                assert W.shape[1] == 515
                W = np.concatenate([W[:, 3:], W[:, :3], np.zeros((W.shape[0], 5))], 1)
            self._dense(arr[k], f"{prefix}.{k}", W, b)

    def _cls_module(self, br, W0, b0, alpha, beta, W4, b4):
        """The segmentation head Conv1d -> ReLU -> BN -> (Dropout) -> Conv1d (TEHNet.py:135-141) as a two-layer ev2h_fp_mlp chain:
        the BN affine y = alpha relu(z) + beta sits between a ReLU and a k=1 convolution, so it folds forward exactly (float64,
        rounded once): W4' = W4 diag(alpha), b4' = W4 beta + b4.  The 4 output rows are zero-padded to one 32-row tile."""
        W4f, b4f = W4 * alpha[None, :], W4 @ beta + b4
        C2, C1 = W0.shape
        assert (C1, C2) == (256, 256) and W4f.shape[0] <= 32
        W4p, b4p = _pad(W4f, 32, C2), _pad(b4f, 32)
        br.b2 = self._dev("clsm.b2", b0)
        br.b3 = self._dev("clsm.b3", b4p)
        br.C1, br.C2, br.C3, br.K, br.radius = C1, C2, 32, 32, 0.0
        br.w1x_norm = 0.0
        br.w2_norm = float(np.abs(W0).sum(1).max()) * (1 + 1e-6)
        br.b2_max = float(np.abs(b0).max()) * (1 + 1e-6)
        i2, i3, br.w2_unscale, br.w3_unscale = sa_bf16_images(W0, W4p, self.ns)
        br.W2s = self._dev_bytes("clsm.W2s", i2)
        br.W3s = self._dev_bytes("clsm.W3s", i3)

    def _fp_module(self, m, sd, prefix):
        """A three-layer feature-propagation MLP without skip input (fp1, TEHNet.py:129) in the form ev2h_fp_mlp takes: the first
        layer as a table over the coarse points (W1f, b1 -- it commutes with the interpolation), layers 2-3 as the tile images of
        the fused set-abstraction kernel."""
        Ws, bs = [], []
        for j in range(3):
            W, b = _fold(sd, f"{prefix}.mlp_convs.{j}", f"{prefix}.mlp_bns.{j}")
            Ws.append(W[:, :, 0])
            bs.append(b)
        C1, C2, C3 = (x.shape[0] for x in Ws)
        m.kf, m.npoint, m.nbranch = Ws[0].shape[1], 0, 1
        assert (C1, C2, C3) == (128, 128, 256) and m.kf % 32 == 0
        br = m.br[0]
        n = prefix + "m"
        br.b2 = self._dev(n + ".b2", bs[1])
        br.b3 = self._dev(n + ".b3", bs[2])
        br.C1, br.C2, br.C3, br.K, br.radius = C1, C2, C3, 32, 0.0
        br.w1x_norm = 0.0
        br.w2_norm = float(np.abs(Ws[1]).sum(1).max()) * (1 + 1e-6)
        br.b2_max = float(np.abs(bs[1]).max()) * (1 + 1e-6)
        i2, i3, br.w2_unscale, br.w3_unscale = sa_bf16_images(Ws[1], Ws[2], self.ns)
        br.W2s = self._dev_bytes(n + ".W2s", i2)
        br.W3s = self._dev_bytes(n + ".W3s", i3)
        m.W1f = self._dev(n + ".W1f", Ws[0])
        m.b1 = self._dev(n + ".b1", bs[0])
        img, m.w1f_unscale = gemm_bf16_w_image(Ws[0], self.ns)
        m.W1fs = self._dev_bytes(n + ".W1fs", img)
        m.w1f_norm = float(np.abs(Ws[0]).sum(1).max()) * (1 + 1e-6)
        m.b1_max = float(np.abs(bs[0]).max()) * (1 + 1e-6)

    def _sa_module(self, m, sd, prefix, nfeat, kf, npoint, radii, nsamples):
        W1f, b1 = [], []
        m.kf, m.npoint, m.nbranch = kf, npoint, len(radii)
        for i, (r, K) in enumerate(zip(radii, nsamples)):
            Ws, bs = [], []
            for j in range(3):
                W, b = _fold(sd, f"{prefix}.conv_blocks.{i}.{j}", f"{prefix}.bn_blocks.{i}.{j}")
                Ws.append(W[:, :, 0, 0])
                bs.append(b)
            C1, C2, C3 = (x.shape[0] for x in Ws)
            assert Ws[0].shape[1] == nfeat + 3                 # [features..., dx, dy, dz] (pointnet2_utils.py:248)
            W1f.append(_pad(Ws[0][:, :nfeat], C1, kf))
            b1.append(bs[0])
            br = m.br[i]
            n = f"{prefix}.{i}"
            br.W1x = self._dev(n + ".W1x", _pad(Ws[0][:, nfeat:], C1, 4))
            br.W2 = self._dev(n + ".W2", _pad(Ws[1], _up(C2, 32), C1))
            br.b2 = self._dev(n + ".b2", _pad(bs[1], _up(C2, 32)))
            br.W3 = self._dev(n + ".W3", _pad(Ws[2], C3, _up(C2, 8)))
            br.b3 = self._dev(n + ".b3", bs[2])
            br.C1, br.C2, br.C3, br.K, br.radius = C1, C2, C3, K, float(r)
            # F16X2 range bounds (ev2h_sa_desc): rounded up a little so that fp32 rounding can never make a bound too small
            br.w1x_norm = float(np.abs(Ws[0][:, nfeat:]).sum(1).max()) * (1 + 1e-6)
            br.w2_norm = float(np.abs(Ws[1]).sum(1).max()) * (1 + 1e-6)
            br.b2_max = float(np.abs(bs[1]).max()) * (1 + 1e-6)
            if self.ns:
                i2, i3, br.w2_unscale, br.w3_unscale = sa_bf16_images(Ws[1], Ws[2], self.ns)
                br.W2s = self._dev_bytes(n + ".W2s", i2)
                br.W3s = self._dev_bytes(n + ".W3s", i3)
        W1f_all, b1_all = np.concatenate(W1f, 0), np.concatenate(b1, 0)
        m.W1f = self._dev(prefix + ".W1f", W1f_all)
        m.b1 = self._dev(prefix + ".b1", b1_all)
        if self.ns and kf >= 32:                               # enc.sa2 (K = 320): plane images, the fast GEMM kernel
            img, m.w1f_unscale = gemm_bf16_w_image(W1f_all, self.ns)
            m.W1fs = self._dev_bytes(prefix + ".W1fs", img)
        else:
            m.w1f_unscale = plane_unscale(W1f_all, self.ns)    # K = 8 tables run as fp32 fma chains (table_k8_kernel)
        m.w1f_norm = float(np.abs(W1f_all).sum(1).max()) * (1 + 1e-6)
        m.b1_max = float(np.abs(b1_all).max()) * (1 + 1e-6)

    def nbytes(self) -> int:
        return sum(t.numel() * 4 for t in self._keep)
