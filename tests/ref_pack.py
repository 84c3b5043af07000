"""TEST INFRASTRUCTURE: an independent numpy restatement of the library's weight packer (csrc/pack.hip, ev2h_pack_weights) --
the Python packer the product used up to round 3.  tests/test_pack_abi.py requires the C packer to reproduce it byte for byte
(every packed array, every plane image, every scalar of ev2h_weights); the product never imports this file.
Sums that decide a rounding (norms of the equalisation, the bound norms, the folded head bias) are taken SEQUENTIALLY (_seqsum)
so that they are defined independently of numpy's pairwise summation.

Checkpoint -> packed device weights for libev2hands_hip.so (host-side weight loading).

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

from ev2hands_amd import _lib, synth

BN_EPS = 1e-5


def _seqsum(a: np.ndarray, axis: int) -> np.ndarray:
    """sum along `axis` in index order (np.cumsum is strictly sequential; np.sum is pairwise)"""
    return np.take(np.cumsum(a, axis=axis), -1, axis=axis)


def _l1(W: np.ndarray) -> float:
    """max row L1 norm, rows summed in order"""
    return float(_seqsum(np.abs(W), 1).max())


def _colnorm(cols: np.ndarray) -> np.ndarray:
    """sqrt of the column sums of squares of [O, n] or [O, n, taps]: taps first ((t0 + t1) + t2), then the rows in order"""
    sq = cols ** 2
    if sq.ndim == 3:
        acc = sq[:, :, 0]
        for t in range(1, sq.shape[2]):
            acc = acc + sq[:, :, t]
        sq = acc
    return np.sqrt(_seqsum(sq, 0))


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


NS_OF = {"f32": 0, "bf16": 1, "f16x2": 2, "bf16x3": 3, "f16": 4}     # plane-mode code per precision (csrc/planes.hpp; 4 = ONE fp16 plane)


def npl(ns: int) -> int:
    """planes stored per operand (the code 4 = "f16" stores one)"""
    return 1 if ns == 4 else ns


def split_bf16_planes(a: np.ndarray, ns: int):
    """fp32 array -> list of `ns` uint16 arrays of 16-bit plane patterns, the same splits the kernels apply to
    activations (csrc/planes.hpp: split_planes).
    ns == 1: bf16, round to nearest even.  ns == 2: two fp16 planes, a = h + l with h = rne(a), l = rne(a - h).
    ns == 3: exact truncation split into three bf16 planes a = h + m + l (8 + 8 + 8 mantissa bits)."""
    x = np.ascontiguousarray(a, dtype=np.float32)
    if ns in (2, 4):
        if x.size and float(np.abs(x).max()) >= 65504.0:
            raise ValueError("f16x2 needs |weight| < 65504 (fp16 range); use precision='bf16x3' or 'f32' for this checkpoint")
        h = x.astype(np.float16)
        if ns == 4:                                    # "f16": the high plane alone
            return [h.view(np.uint16)]
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
    if ns not in (2, 4):
        return 1.0
    m = float(np.abs(np.asarray(W, dtype=np.float64)).max()) if np.size(W) else 0.0
    if m == 0.0 or not np.isfinite(m):
        return 1.0
    import math
    k = math.frexp(16384.0 / m)[1] - 1          # floor(log2(16384 / m)), exactly
    k = max(-24, min(k, 60))
    return float(2.0 ** -k)


def chain_unscale(W2: np.ndarray) -> float:
    """f16 (one fp16 plane): the factor of a chain's second matrix, 2^floor(log2 of the largest row L1 norm) -- the fused
    set-abstraction kernels keep one power of two per window for the whole chain in that mode (csrc/pack.hip: chain_unscale)."""
    import math
    m = float(np.abs(np.asarray(W2, dtype=np.float64)).sum(axis=1).max()) if np.size(W2) else 0.0
    if m == 0.0 or not np.isfinite(m):
        return 1.0
    return float(2.0 ** max(-60, min(math.frexp(m)[1] - 1, 24)))


def sa_bf16_geometry(C2: int):
    T2 = _up(C2, 32) // 32
    rem = C2 % 32
    m_last = 2 if rem == 0 else (1 if rem <= 16 else 2)
    return T2, 32 * (T2 - 1) + 16 * m_last


def kernel_geometry(C1: int, C2: int, C3: int, ns: int) -> dict:
    """Tile-image geometry of a chain as the KERNELS define it (ev2h_tile_geometry: csrc/sa_mlp_bf16.hip SaBCfg, csrc/gemm_bf16.hip
    GBCfg) -- the one source of truth the image builders below are asserted against."""
    out = (C.c_int * 10)()
    _lib.check(_lib.lib().ev2h_tile_geometry(C1, C2, C3, ns, out), "ev2h_tile_geometry")
    return dict(zip(("T2", "C2P", "RS2", "RS3", "TB2", "TB3", "GEMM_RS", "GEMM_BK", "LEFTOVER", "W2PERM"), out))


def sa_bf16_images(W2: np.ndarray, W3: np.ndarray, ns: int):
    """Byte images of the LDS weight tiles of sa_mlp_max_bf16_kernel (see SaBCfg in csrc/sa_mlp_bf16.hip).
    W2 [C2, C1], W3 [C3, C2] folded fp32 weights.  Returns (W2s, W3s, u2, u3): uint8 images of W2 / u2 and W3 / u3 and the
    power-of-two factors (plane_unscale) the kernel multiplies back."""
    u2, u3 = (chain_unscale(W2) if ns == 4 else plane_unscale(W2, ns)), plane_unscale(W3, ns)
    W2 = np.asarray(W2, dtype=np.float64) / u2
    W3 = np.asarray(W3, dtype=np.float64) / u3
    C2, C1 = W2.shape
    C3 = W3.shape[0]
    T2, C2P = sa_bf16_geometry(C2)
    g = kernel_geometry(C1, C2, C3, ns)
    left = g["LEFTOVER"]               # 0, or the 1..4 channels past the last full tile whose plane products share MFMAs (SaBCfg::PACK4)
    base = 32 * (T2 - 1)
    W2p = _pad(W2, T2 * 32, C1)
    if g.get("W2PERM"):                # BF16: k slots of every 32-column chunk in the D-register order of the layer-1 MFMA
        pos = np.arange(32)
        h, m, e = pos // 16, (pos % 16) // 8, pos % 8
        src = 16 * m + 4 * h + (e & 3) + 8 * (e >> 2)
        W2p = np.concatenate([W2p[:, 32 * c:32 * c + 32][:, src] for c in range(C1 // 32)], 1)
    p2 = split_bf16_planes(W2p, ns)
    if left:
        assert ns == 2 and C2 - base == left and left <= 4
        p2[0][base + 8:base + 8 + left] = p2[1][base:base + left]        # rows 8.. of the high-plane image: the leftover rows' LOW plane
    rs2 = npl(ns) * 64 + 16
    img2 = np.zeros((C1 // 32, T2 * 32, rs2), dtype=np.uint8)
    for c in range(C1 // 32):
        for s_ in range(npl(ns)):
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
    if left:                                                              # last 16 k-slots of every row: [wh | wh | wl | 0]
        wh, wl = p3[0][:, base:base + 4].copy(), p3[1][:, base:base + 4].copy()      # (positions base..base+3 <-> channels base..base+3)
        p3[0][:, base + 4:base + 8] = wh
        p3[0][:, base + 8:base + 12] = wl
        p3[0][:, base + 12:base + 16] = 0
    rs3 = npl(ns) * C2P * 2 + 16
    img3 = np.zeros((C3 // 32, 32, rs3), dtype=np.uint8)
    for s_ in range(npl(ns)):
        blk = p3[s_].view(np.uint8).reshape(C3 // 32, 32, C2P * 2)
        img3[:, :, s_ * C2P * 2:(s_ + 1) * C2P * 2] = blk
    mine = {"T2": T2, "C2P": C2P, "RS2": rs2, "RS3": rs3, "TB2": img2[0].size, "TB3": img3[0].size}
    if any(g[k] != v for k, v in mine.items()):
        raise _lib.Ev2hError(f"tile geometry of chain {C1}-{C2}-{C3} (planes {ns}): pack.py builds {mine}, the kernels expect {g}")
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
    rs = npl(ns) * 64 + 16
    g = kernel_geometry(128, 128, 256, ns)
    if (g["GEMM_RS"], g["GEMM_BK"]) != (rs, 32):
        raise _lib.Ev2hError(f"dense W image geometry: pack.py builds rows of {rs} B x 32 k, the kernels expect {g['GEMM_RS']} B x {g['GEMM_BK']} k")
    img = np.zeros((tn, nk, rows, rs), dtype=np.uint8)
    for s_ in range(npl(ns)):
        blk = planes[s_].reshape(tn, rows, nk, 32).transpose(0, 2, 1, 3)         # [tn, nk, rows, 32] uint16
        img[:, :, :, s_ * 64:(s_ + 1) * 64] = np.ascontiguousarray(blk).view(np.uint8).reshape(tn, nk, rows, 64)
    return img.reshape(-1), u



# ------------------------------------------------------------------------------------ folded checkpoint + channel equalisation
def fold_checkpoint(sd: dict) -> dict:
    """Every layer of the path with eval-BN folded (float64): name -> {"W": [O, I] (k=3 convolutions: [O, I, 3]), "b": [O],
    "ps"/"pt": post-ReLU BatchNorm scale / shift [O] or None}, input columns in the CHECKPOINT's order."""
    F = {}

    def msg(prefix, nbranch):
        for i in range(nbranch):
            for j in range(3):
                W, b = _fold(sd, f"{prefix}.conv_blocks.{i}.{j}", f"{prefix}.bn_blocks.{i}.{j}")
                F[f"{prefix}.{i}.{j}"] = {"W": W[:, :, 0, 0], "b": b, "ps": None, "pt": None}

    def stack(prefix, n):
        for k in range(n):
            W, b = _fold(sd, f"{prefix}.mlp_convs.{k}", f"{prefix}.mlp_bns.{k}")
            F[f"{prefix}.{k}"] = {"W": W.reshape(W.shape[0], W.shape[1]), "b": b, "ps": None, "pt": None}

    def post(name, pc, pb):
        W = _np(sd[pc + ".weight"])
        a, be = _bn_affine(sd, pb)
        F[name] = {"W": W[:, :, 0] if (W.ndim == 3 and W.shape[2] == 1) else W, "b": _np(sd[pc + ".bias"]), "ps": a, "pt": be}

    def plain(name, pc):
        W = _np(sd[pc + ".weight"])
        F[name] = {"W": W[:, :, 0] if W.ndim == 3 else W, "b": _np(sd[pc + ".bias"]), "ps": None, "pt": None}

    msg("sa1", 3)
    msg("sa2", 2)
    stack("sa3", 3)
    stack("fp3", 2)
    stack("fp2", 2)
    stack("fp1", 3)
    post("cls0", "classifier.0", "classifier.2")
    plain("cls4", "classifier.4")
    for side in ("left", "right"):
        q = f"{side}_query_conv"
        post(q + ".0", q + ".0", q + ".2")                               # W [256, 256, 3]
        W4, b4 = _fold(sd, q + ".4", q + ".5")
        F[q + ".4"] = {"W": W4, "b": b4, "ps": None, "pt": None}
        p = f"{side}_mano_regressor"
        msg(p + ".sa1", 2)
        stack(p + ".sa2", 2)
        post(p + ".head0", p + ".mano_regressor.0", p + ".mano_regressor.2")
        plain(p + ".head4", p + ".mano_regressor.4")
    return F


def hidden_tensors():
    """The hidden tensors of the path as (name, producers, consumers): producers = [(layer, "rows" | "post")] in channel order
    (a concatenation has several), consumers = [(layer, first input column)].  Every producer ends in a ReLU (positively
    homogeneous) or a post-ReLU affine, and everything between producer and consumer (gather, max-pool, 3-NN interpolation,
    concatenation, broadcast) acts per channel -- so channel c may be multiplied by any e_c > 0 at the producer and divided at the
    consumers without changing the network function.  Column offsets: pointnet2_utils.py:155,248,261,307, TEHNet.py:179-195."""
    T = []

    def msg(p, nb, consumers):
        outs = []
        for i in range(nb):
            T.append((f"{p}.{i}.h1", [(f"{p}.{i}.0", "rows")], [(f"{p}.{i}.1", 0)]))
            T.append((f"{p}.{i}.h2", [(f"{p}.{i}.1", "rows")], [(f"{p}.{i}.2", 0)]))
            outs.append((f"{p}.{i}.2", "rows"))
        T.append((p + ".out", outs, consumers))

    msg("sa1", 3, [("sa2.0.0", 0), ("sa2.1.0", 0), ("fp2.0", 0)])
    msg("sa2", 2, [("sa3.0", 3), ("fp3.0", 0)])
    T.append(("sa3.h1", [("sa3.0", "rows")], [("sa3.1", 0)]))
    T.append(("sa3.h2", [("sa3.1", "rows")], [("sa3.2", 0)]))
    T.append(("l3", [("sa3.2", "rows")], [("fp3.0", 512)]))
    T.append(("fp3.h", [("fp3.0", "rows")], [("fp3.1", 0)]))
    T.append(("fp3.out", [("fp3.1", "rows")], [("fp2.0", 320)]))
    T.append(("fp2.h", [("fp2.0", "rows")], [("fp2.1", 0)]))
    T.append(("fp2.out", [("fp2.1", "rows")], [("fp1.0", 0)]))
    T.append(("fp1.h1", [("fp1.0", "rows")], [("fp1.1", 0)]))
    T.append(("fp1.h2", [("fp1.1", "rows")], [("fp1.2", 0)]))
    # l0 also is the attention's `value` (TEHNet.py:20-26: context = sim @ value, linear in value): that consumer divides by e_c
    # through ev2h_weights.l0_unscale
    T.append(("l0", [("fp1.2", "rows")], [("cls0", 0), ("left_query_conv.0", 0), ("right_query_conv.0", 0)]))
    T.append(("cls.h", [("cls0", "post")], [("cls4", 0)]))
    for side in ("left", "right"):
        q, p = f"{side}_query_conv", f"{side}_mano_regressor"
        T.append((q + ".h", [(q + ".0", "post")], [(q + ".4", 0)]))
        msg(p + ".sa1", 2, [(p + ".sa2.0", 3)])
        T.append((p + ".sa2.h", [(p + ".sa2.0", "rows")], [(p + ".sa2.1", 0)]))
        T.append((p + ".sa2.out", [(p + ".sa2.1", "rows")], [(p + ".head0", 0)]))
        T.append((p + ".fc1", [(p + ".head0", "post")], [(p + ".head4", 0)]))
    return T


# Layers whose contraction mixes hidden channels with RAW inputs (the group-all set abstractions read [xyz | features],
# pointnet2_utils.py:155): layer -> the raw input columns.  The hidden tensor that feeds such a layer is anchored as a whole (see
# equalize_channels): coordinates are O(1), and a checkpoint whose features are 1e6 x larger (with weights 1e-6 x smaller) would
# otherwise push the coordinates 2^20 below the window's maximum and the feature weights 2^20 below the matrix maximum.
RAW_COLUMNS = {"sa3.0": slice(0, 3), "left_mano_regressor.sa2.0": slice(0, 3), "right_mano_regressor.sa2.0": slice(0, 3)}


def equalize_channels(F: dict, sweeps: int = 3) -> dict:
    """Cross-layer channel equalisation by exact powers of two, in place; returns name -> e [channels] (the accumulated factor
    of every hidden tensor).

    Why: the f16x2 arithmetic scales every tensor by ONE power of two per window and every weight matrix by one per matrix
    (csrc/planes.hpp, plane_unscale); a value then keeps its 22 bits only down to 2^-17 of the tensor's maximum.  A checkpoint
    may distribute magnitude between a hidden channel and the weights that read it in any way (BatchNorm scales gamma_c of a
    trained model differ by orders of magnitude; gamma_c -> a gamma_c with the consumers' columns / a is the same network):
    with channels 2^16 apart the small channels and, symmetrically, the small weight columns lost most of their low planes
    (measured 2.7e-3 instead of 3e-7).  This pass picks the representative of that equivalence class that the split arithmetic
    likes: e_c = 2^round(log2(sqrt(column norm of the consumers / row norm of the producer))) balances what the channel carries
    against what multiplies it (Nagel et al., data-free quantisation, 2019, restricted to powers of two so that every product
    and every partial sum changes by an exact power of two: the fp32 result of every layer is bit-identical, only the 16-bit
    planes see better-conditioned operands).  No calibration data is needed and nothing runs per forward."""
    acc = {}
    for _ in range(sweeps):
        for name, producers, consumers in hidden_tensors():
            r = []
            for layer, kind in producers:
                L = F[layer]
                W2 = L["W"].reshape(L["W"].shape[0], -1)
                rn = np.sqrt(_seqsum(W2 ** 2, 1) + L["b"] ** 2)
                r.append(np.abs(L["ps"]) * rn + np.abs(L["pt"]) if kind == "post" else rn)
            r = np.concatenate(r)
            n = r.shape[0]
            om = np.zeros(n)
            for layer, off in consumers:
                W = F[layer]["W"]
                cols = W[:, off:off + n]
                cn = _colnorm(cols)
                rms = np.sqrt(_seqsum(cn ** 2, 0) / n)
                if rms > 0:
                    om += (cn / rms) ** 2
            om = np.sqrt(om)
            ok = (r > 0) & (om > 0) & np.isfinite(r) & np.isfinite(om)
            lg = np.zeros(n)
            lg[ok] = 0.5 * (np.log2(om[ok]) - np.log2(r[ok]))
            if ok.any():
                lg[ok] -= np.median(lg[ok])                  # keep the tensor's overall magnitude where the checkpoint put it ...
            for layer, off in consumers:                     # ... unless a consumer mixes it with raw inputs in one contraction:
                if layer in RAW_COLUMNS and ok.any():        # then the hidden columns are brought level with the raw ones
                    W = F[layer]["W"]
                    raw = _colnorm(W[:, RAW_COLUMNS[layer]])
                    hid = _colnorm(W[:, off:off + n])[ok] / np.exp2(np.round(lg[ok]))     # column norms after this step
                    if raw.size and np.median(raw) > 0 and np.median(hid) > 0:
                        lg[ok] -= np.round(np.log2(np.median(raw) / np.median(hid)))
            e = np.exp2(np.clip(np.round(lg), -40, 40))
            o = 0
            for layer, kind in producers:
                L = F[layer]
                k = L["W"].shape[0]
                ek = e[o:o + k]
                if kind == "post":
                    L["ps"] = L["ps"] * ek
                    L["pt"] = L["pt"] * ek
                else:
                    L["W"] = L["W"] * ek.reshape((-1,) + (1,) * (L["W"].ndim - 1))
                    L["b"] = L["b"] * ek
                o += k
            for layer, off in consumers:
                W = F[layer]["W"].copy()
                W[:, off:off + n] = W[:, off:off + n] / e.reshape((1, n) + (1,) * (W.ndim - 2))
                F[layer]["W"] = W
            acc[name] = acc.get(name, np.ones(n)) * e
    return acc


class PackedWeights:
    """Owns the device tensors and the ev2h_weights struct that points at them."""

    def __init__(self, sd: dict, device, in_channels: int, precision: str = "f32", equalize: bool = True):
        self.device = torch.device(device)
        self.in_channels = in_channels
        self.precision = precision
        self.ns = NS_OF[precision]
        self._keep = []
        self.tensors = {}
        self.struct = _lib.Weights()
        w = self.struct
        w.precision = _lib.PREC[precision]
        if precision == "f16":
            w.f16_families = _lib.FAM_ALL
        F = fold_checkpoint(sd)
        # exact power-of-two channel equalisation (equalize_channels): applied in EVERY precision mode, so that all modes run the
        # same network representation (the exact-fp32 results do not change by a bit)
        self.equalization = equalize_channels(F) if equalize else {}
        self._sa_module(w.sa1, F, "sa1", in_channels, 8, synth.SA1_NPOINT, synth.SA1_RADII, synth.SA1_NSAMPLE)
        self._sa_module(w.sa2, F, "sa2", 320, 320, synth.SA2_NPOINT, synth.SA2_RADII, synth.SA2_NSAMPLE)
        for h, side in enumerate(("left", "right")):
            p = f"{side}_mano_regressor"
            self._sa_module(w.mano_sa1[h], F, p + ".sa1", 4, 8, synth.MANO_SA1_NPOINT, synth.MANO_SA1_RADII,
                            synth.MANO_SA1_NSAMPLE)
            self._group_all(w.mano_sa2[h], F, p + ".sa2", 2)
            L0, L4 = F[p + ".head0"], F[p + ".head4"]
            self._dense(w.head0[h], p + ".head0", L0["W"], L0["b"], L0["ps"], L0["pt"])
            self._dense(w.head4[h], p + ".head4", L4["W"], L4["b"])
        self._group_all(w.sa3, F, "sa3", 3)
        # fp3: 1536 = 512 skip (l2_points) + 1024 broadcast (l3_points), pointnet2_utils.py:293-294,307
        W, b = F["fp3.0"]["W"], F["fp3.0"]["b"]
        self._dense(w.fp3_skip, "fp3.skip", W[:, :512], None)
        self._dense(w.fp3_bcast, "fp3.bcast", W[:, 512:], b)
        self._dense(w.fp3_1, "fp3.1", F["fp3.1"]["W"], F["fp3.1"]["b"])
        for k in range(2):
            self._dense(w.fp2[k], f"fp2.{k}", F[f"fp2.{k}"]["W"], F[f"fp2.{k}"]["b"])
        for k in range(3):
            self._dense(w.fp1[k], f"fp1.{k}", F[f"fp1.{k}"]["W"], F[f"fp1.{k}"]["b"])
        if self.ns:
            self._fp_module(w.fp1m, F, "fp1")
        c0, c4 = F["cls0"], F["cls4"]
        self._dense(w.cls0, "cls0", c0["W"], c0["b"], c0["ps"], c0["pt"])
        self._dense(w.cls4, "cls4", c4["W"], c4["b"])
        if self.ns:
            self._cls_module(w.clsm, c0["W"], c0["b"], c0["ps"], c0["pt"], c4["W"], c4["b"])
        # query convs: tap-major [O][3*I]; both hands' first conv stacked along O
        W0, b0, a0, be0 = [], [], [], []
        for h, side in enumerate(("left", "right")):
            p = f"{side}_query_conv"
            L = F[p + ".0"]                                                # W [O, I, 3]
            W0.append(np.ascontiguousarray(L["W"].transpose(0, 2, 1)).reshape(L["W"].shape[0], -1))
            b0.append(L["b"])
            a0.append(L["ps"])
            be0.append(L["pt"])
            W4, b4 = F[p + ".4"]["W"], F[p + ".4"]["b"]
            W4 = np.ascontiguousarray(W4.transpose(0, 2, 1)).reshape(W4.shape[0], -1)
            self._dense(w.qconv4[h], p + ".4", W4, b4, K=256)
            w.qconv4T[h] = self._dev(p + ".4.WT", np.ascontiguousarray(W4.T))        # [768, 256]: ev2h_attn_sim_folded
        self._dense(w.qconv0, "qconv0", np.concatenate(W0, 0), np.concatenate(b0), np.concatenate(a0),
                    np.concatenate(be0), K=256)
        # the attention's `value` is l0 itself (TEHNet.py:20-26): its channels are divided by their equalisation factor there
        e_l0 = self.equalization.get("l0")
        w.l0_unscale = self._dev("l0.unscale", 1.0 / e_l0) if e_l0 is not None and np.any(e_l0 != 1.0) else None

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

    def _group_all(self, arr, F, prefix, nlayers):
        """sample_and_group_all concatenates [xyz(3), features(512)] (pointnet2_utils.py:155); our
        buffers hold [features(512) | xyz(3) | 0 x 5] so that K = 520 is a multiple of 8."""
        for k in range(nlayers):
            W, b = F[f"{prefix}.{k}"]["W"], F[f"{prefix}.{k}"]["b"]
            if k == 0:
                assert W.shape[1] == 515
                W = np.concatenate([W[:, 3:], W[:, :3], np.zeros((W.shape[0], 5))], 1)
            self._dense(arr[k], f"{prefix}.{k}", W, b)

    def _cls_module(self, br, W0, b0, alpha, beta, W4, b4):
        """The segmentation head Conv1d -> ReLU -> BN -> (Dropout) -> Conv1d (TEHNet.py:135-141) as a two-layer ev2h_fp_mlp chain:
        the BN affine y = alpha relu(z) + beta sits between a ReLU and a k=1 convolution, so it folds forward exactly (float64,
        rounded once): W4' = W4 diag(alpha), b4' = W4 beta + b4.  The 4 output rows are zero-padded to one 32-row tile."""
        W4f, b4f = W4 * alpha[None, :], _seqsum(W4 * beta[None, :], 1) + b4
        C2, C1 = W0.shape
        assert (C1, C2) == (256, 256) and W4f.shape[0] <= 32
        W4p, b4p = _pad(W4f, 32, C2), _pad(b4f, 32)
        br.b2 = self._dev("clsm.b2", b0)
        br.b3 = self._dev("clsm.b3", b4p)
        br.C1, br.C2, br.C3, br.K, br.radius = C1, C2, 32, 32, 0.0
        br.w1x_norm = 0.0
        br.w2_norm = _l1(W0) * (1 + 1e-6)
        br.b2_max = float(np.abs(b0).max()) * (1 + 1e-6)
        br.w3_norm = _l1(W4p) * (1 + 1e-6)
        br.b3_max = float(np.abs(b4p).max()) * (1 + 1e-6)
        i2, i3, br.w2_unscale, br.w3_unscale = sa_bf16_images(W0, W4p, self.ns)
        br.W2s = self._dev_bytes("clsm.W2s", i2)
        br.W3s = self._dev_bytes("clsm.W3s", i3)

    def _fp_module(self, m, F, prefix):
        """A three-layer feature-propagation MLP without skip input (fp1, TEHNet.py:129) in the form ev2h_fp_mlp takes: the first
        layer as a table over the coarse points (W1f, b1 -- it commutes with the interpolation), layers 2-3 as the tile images of
        the fused set-abstraction kernel."""
        Ws = [F[f"{prefix}.{j}"]["W"] for j in range(3)]
        bs = [F[f"{prefix}.{j}"]["b"] for j in range(3)]
        C1, C2, C3 = (x.shape[0] for x in Ws)
        m.kf, m.npoint, m.nbranch = Ws[0].shape[1], 0, 1
        assert (C1, C2, C3) == (128, 128, 256) and m.kf % 32 == 0
        br = m.br[0]
        n = prefix + "m"
        br.b2 = self._dev(n + ".b2", bs[1])
        br.b3 = self._dev(n + ".b3", bs[2])
        br.C1, br.C2, br.C3, br.K, br.radius = C1, C2, C3, 32, 0.0
        br.w1x_norm = 0.0
        br.w2_norm = _l1(Ws[1]) * (1 + 1e-6)
        br.b2_max = float(np.abs(bs[1]).max()) * (1 + 1e-6)
        br.w3_norm = _l1(Ws[2]) * (1 + 1e-6)
        br.b3_max = float(np.abs(bs[2]).max()) * (1 + 1e-6)
        i2, i3, br.w2_unscale, br.w3_unscale = sa_bf16_images(Ws[1], Ws[2], self.ns)
        br.W2s = self._dev_bytes(n + ".W2s", i2)
        br.W3s = self._dev_bytes(n + ".W3s", i3)
        m.W1f = self._dev(n + ".W1f", Ws[0])
        m.b1 = self._dev(n + ".b1", bs[0])
        img, m.w1f_unscale = gemm_bf16_w_image(Ws[0], self.ns)
        m.W1fs = self._dev_bytes(n + ".W1fs", img)
        m.w1f_norm = _l1(Ws[0]) * (1 + 1e-6)
        m.b1_max = float(np.abs(bs[0]).max()) * (1 + 1e-6)

    def _sa_module(self, m, F, prefix, nfeat, kf, npoint, radii, nsamples):
        W1f, b1 = [], []
        m.kf, m.npoint, m.nbranch = kf, npoint, len(radii)
        for i, (r, K) in enumerate(zip(radii, nsamples)):
            Ws = [F[f"{prefix}.{i}.{j}"]["W"] for j in range(3)]
            bs = [F[f"{prefix}.{i}.{j}"]["b"] for j in range(3)]
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
            br.w1x_norm = _l1(Ws[0][:, nfeat:]) * (1 + 1e-6)
            br.w1f_unscale = plane_unscale(Ws[0][:, :nfeat], self.ns)
            br.w1x_unscale = plane_unscale(Ws[0][:, nfeat:], self.ns)
            br.w2_norm = _l1(Ws[1]) * (1 + 1e-6)
            br.b2_max = float(np.abs(bs[1]).max()) * (1 + 1e-6)
            br.w3_norm = _l1(Ws[2]) * (1 + 1e-6)
            br.b3_max = float(np.abs(bs[2]).max()) * (1 + 1e-6)
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
        m.w1f_norm = _l1(W1f_all) * (1 + 1e-6)
        m.b1_max = float(np.abs(b1_all).max()) * (1 + 1e-6)

    def nbytes(self) -> int:
        return sum(t.numel() * 4 for t in self._keep)
