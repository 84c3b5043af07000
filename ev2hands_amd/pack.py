"""Checkpoint -> packed device weights: marshalling for `ev2h_pack_weights` (include/ev2hands_hip.h, csrc/pack.hip).

The work itself -- eval-BatchNorm folding in float64, the power-of-two channel equalisation the f16x2 arithmetic relies on,
layouts, operand-plane images, the range bounds and the upload -- happens inside libev2hands_hip.so, so that a host in any
language gets the same packed weights (and the same accuracy contract) as this wrapper.  What the reference does with
`load_state_dict` (/root/reference/src/Ev2Hands/model/model.py:14-23, demo.py:83-84) ends here: the state dict's tensors are
handed over as host arrays under their own names.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

NS_OF = {"f32": 0, "bf16": 1, "f16x2": 2, "bf16x3": 3, "f16": 4}     # plane-mode code per precision (csrc/planes.hpp; 4 = "f16": ONE fp16 plane)
GEMM_W_TILE_ROWS = 128     # rows per W image tile (128: occupancy kernel, 256: wide kernel)

_DT = {np.dtype(np.float32): _lib.DT_F32, np.dtype(np.float64): _lib.DT_F64, np.dtype(np.int64): _lib.DT_I64}


def _host_array(t) -> np.ndarray:
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    if a.dtype not in _DT:
        a = a.astype(np.float32)
    return np.ascontiguousarray(a)


def tensor_descs(sd: dict):
    """state dict -> (ctypes array of ev2h_tensor_desc, the objects that keep its pointers alive)"""
    keep, descs = [], (_lib.TensorDesc * len(sd))()
    for d, (name, t) in zip(descs, sd.items()):
        a = _host_array(t)
        if a.ndim > 4:
            raise _lib.Ev2hError(f"checkpoint entry {name} has {a.ndim} dimensions")
        nm = name.encode()
        keep += [a, nm]
        d.name, d.data, d.dtype, d.ndim = nm, a.ctypes.data, _DT[a.dtype], a.ndim
        for k, n in enumerate(a.shape):
            d.shape[k] = n
    return descs, keep


class PackedWeights:
    """Owns an `ev2h_packed` handle (device allocation + the ev2h_weights view into it).  device "cpu": host-only pack (layout
    tests without a GPU)."""

    def __init__(self, sd: dict, device, in_channels: int, precision: str = "f32", equalize: bool = True, f16_families: int = 0):
        self.device = torch.device(device)
        self.in_channels = in_channels
        self.precision = precision
        self.ns = NS_OF[precision]
        L = _lib.lib()
        descs, keep = tensor_descs(sd)
        flags = (_lib.PACK_EQUALIZE if equalize else _lib.PACK_UNEQUALIZED_OK)     # equalize=False is the explicit opt-out
        if precision == "f16":
            flags |= _lib.pack_f16_families(f16_families)      # 0 = all families (the mode's definition); a partial mask = mixed with f16x2
        self._handle = C.c_void_p()
        self._free = L.ev2h_packed_free
        if self.device.type == "cuda":
            with torch.cuda.device(self.device):
                rc = L.ev2h_pack_weights(descs, len(descs), in_channels, _lib.PREC[precision], flags, C.byref(self._handle))
        else:
            rc = L.ev2h_pack_weights(descs, len(descs), in_channels, _lib.PREC[precision], flags | _lib.PACK_HOST_ONLY, C.byref(self._handle))
        del keep
        _lib.check(rc, "ev2h_pack_weights")
        self.struct = _lib.Weights.from_address(L.ev2h_packed_weights(self._handle))
        self._tensors = None
        self._eq = None

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h:
            try:
                self._free(h)
            except Exception:       # interpreter shutdown
                pass

    @property
    def tensors(self) -> dict:
        """name -> CPU tensor (a copy of the handle's host image): fp32 arrays in the kernels' layouts, uint8 plane images"""
        if self._tensors is None:
            L, out = _lib.lib(), {}
            name, rows, cols, dt, host = C.c_char_p(), C.c_int(), C.c_int(), C.c_int(), C.c_void_p()
            for i in range(L.ev2h_packed_tensor_count(self._handle)):
                _lib.check(L.ev2h_packed_tensor(self._handle, i, C.byref(name), C.byref(rows), C.byref(cols), C.byref(dt), C.byref(host), None))
                n = max(rows.value, 1) * cols.value
                if dt.value == _lib.DT_F32:
                    a = np.ctypeslib.as_array(C.cast(host, C.POINTER(C.c_float)), (n,)).copy() if n else np.zeros(0, np.float32)
                    a = a.reshape(rows.value, cols.value) if rows.value else a
                else:
                    a = np.ctypeslib.as_array(C.cast(host, C.POINTER(C.c_uint8)), (n,)).copy() if n else np.zeros(0, np.uint8)
                out[name.value.decode()] = torch.from_numpy(a)
            self._tensors = out
        return self._tensors

    @property
    def equalization(self) -> dict:
        """hidden tensor -> float64 [channels]: the power of two each channel was multiplied by ({} without equalisation)"""
        if self._eq is None:
            L, out = _lib.lib(), {}
            name, e, n = C.c_char_p(), C.POINTER(C.c_double)(), C.c_int()
            for i in range(L.ev2h_packed_equalization_count(self._handle)):
                _lib.check(L.ev2h_packed_equalization(self._handle, i, C.byref(name), C.byref(e), C.byref(n)))
                out[name.value.decode()] = np.ctypeslib.as_array(e, (n.value,)).copy()
            self._eq = out
        return self._eq

    def weight_spread(self) -> dict:
        """f16x2 only: matrix -> (non-zero weights, weights more than ~2^17 below the matrix maximum, more than ~2^28 below)"""
        L, out = _lib.lib(), {}
        name, c = C.c_char_p(), (C.c_uint64 * 3)()
        for i in range(L.ev2h_packed_weight_spread_count(self._handle)):
            _lib.check(L.ev2h_packed_weight_spread(self._handle, i, C.byref(name), c))
            out[name.value.decode()] = tuple(int(v) for v in c)
        return out

    def nbytes(self) -> int:
        return int(_lib.lib().ev2h_packed_bytes(self._handle))


# ------------------------------------------------------------------------------------ single images (operator-level callers)
def _pad(a: np.ndarray, rows: int, cols: int) -> np.ndarray:
    out = np.zeros((rows, cols), dtype=a.dtype)
    out[:a.shape[0], :a.shape[1]] = a
    return out


def _f64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def plane_unscale(W, ns: int) -> float:
    W = _f64(W)
    return float(_lib.lib().ev2h_plane_unscale(W.ctypes.data, W.size, ns))


def sa_bf16_images(W2, W3, ns: int):
    """(W2s, W3s, u2, u3): uint8 tile images of W2 [C2, C1] / u2 and W3 [C3, C2] / u3 for ev2h_sa_mlp_max / ev2h_fp_mlp"""
    W2, W3 = _f64(W2), _f64(W3)
    (C2, C1), C3 = W2.shape, W3.shape[0]
    L = _lib.lib()
    nb = (C.c_size_t * 2)()
    _lib.check(L.ev2h_pack_sa_image_bytes(C1, C2, C3, ns, nb), "ev2h_pack_sa_image_bytes")
    i2, i3 = np.empty(nb[0], np.uint8), np.empty(nb[1], np.uint8)
    u2, u3 = C.c_float(), C.c_float()
    _lib.check(L.ev2h_pack_sa_images(W2.ctypes.data, W3.ctypes.data, C1, C2, C3, ns, i2.ctypes.data, i3.ctypes.data, C.byref(u2), C.byref(u3)),
               "ev2h_pack_sa_images")
    return i2, i3, u2.value, u3.value


def gemm_bf16_w_image(W, ns: int, rows: int = GEMM_W_TILE_ROWS):
    """(image, u): plane images of a dense weight W [N, K] / u for the 16-bit GEMM kernels"""
    W = _f64(W)
    N, K = W.shape
    L = _lib.lib()
    img = np.empty(L.ev2h_pack_gemm_image_bytes(N, K, ns, rows), np.uint8)
    u = C.c_float()
    _lib.check(L.ev2h_pack_gemm_image(W.ctypes.data, N, K, ns, rows, img.ctypes.data, C.byref(u)), "ev2h_pack_gemm_image")
    return img, u.value


def kernel_geometry(C1: int, C2: int, C3: int, ns: int) -> dict:
    """Tile-image geometry of a chain as the KERNELS define it (ev2h_tile_geometry)"""
    out = (C.c_int * 10)()
    _lib.check(_lib.lib().ev2h_tile_geometry(C1, C2, C3, ns, out), "ev2h_tile_geometry")
    return dict(zip(("T2", "C2P", "RS2", "RS3", "TB2", "TB3", "GEMM_RS", "GEMM_BK", "LEFTOVER", "W2PERM"), out))
