"""Host side of the MANO layer: asset handling and the adapter object the reference's callers use.

Mirrors the interface of the reference's `create_mano_layers` / `SmplxAdapter`
(/root/reference/src/Ev2Hands/model/utils.py:13-42): a dict {'left','right'} of objects with
`.faces` (ndarray [1538,3]), `.shapedirs` (tensor), `.m` (the layer, with manopth's `th_*`
buffer names) and `__call__(global_orient, hand_pose, betas, transl) -> obj(.vertices, .joints)`
in metres.  The arithmetic runs in the HIP kernel `ev2h_mano` (csrc/mano.hip).
"""
from __future__ import annotations

import ctypes as C
import os
import pickle

import numpy as np
import torch

from . import _lib, synth


class ManoOutput:
    def __init__(self, vertices, joints):
        self.vertices = vertices
        self.joints = joints


class _LayerBuffers:
    """Stand-in for manopth.ManoLayer's registered buffers (names as in manopth)."""
    pass


class ManoHand:
    def __init__(self, assets: dict, device, ncomps: int = synth.MANO_CMPS):
        self.side = assets["side"]
        self.device = torch.device(device)
        self.ncomps = ncomps
        self._assets = {k: (np.array(v, dtype=np.float64) if isinstance(v, np.ndarray) and v.dtype.kind == "f" else v)
                        for k, v in assets.items()}
        self.faces = np.asarray(assets["faces"]).astype(np.int64)
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device=self.device)
        m = _LayerBuffers()
        m.th_shapedirs = t(self._assets["shapedirs"])
        m.th_posedirs = t(self._assets["posedirs"])
        m.th_v_template = t(self._assets["v_template"]).unsqueeze(0)
        m.th_J_regressor = t(self._assets["J_regressor"])
        m.th_weights = t(self._assets["weights"])
        m.th_faces = torch.as_tensor(self.faces, device=self.device)
        m.th_hands_mean = t(self._assets["hands_mean"]).unsqueeze(0)
        m.th_selected_comps = t(self._assets["hands_components"][:ncomps])
        self.m = m
        self.shapedirs = m.th_shapedirs
        self._consts = None
        self._keep = []
        self._packed_version = None

    # -- packing -------------------------------------------------------------------------------
    def consts(self) -> "_lib.ManoConsts":
        """Packed constants for ev2h_mano; rebuilt when `shapedirs` was modified in place
        (the reference flips the left hand's first shape component, utils.py:38-40)."""
        ver = self.shapedirs._version
        if self._consts is not None and self._packed_version == ver:
            return self._consts
        a = self._assets
        sd = self.shapedirs.detach().cpu().double().numpy()
        pdirs = a["posedirs"]
        blend = np.zeros((145, 2336), dtype=np.float64)
        blend[:10, :2334] = sd.reshape(2334, 10).T
        blend[10:, :2334] = pdirs.reshape(2334, 135).T
        jr = a["J_regressor"]
        J_t = jr @ a["v_template"]                                  # [16,3]
        J_s = np.einsum("jv,vck->kjc", jr, sd).reshape(10, 48)     # [10][j*3+c]
        keep = []

        def dev(x):
            tt = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float32))).to(self.device)
            keep.append(tt)
            return tt.data_ptr()

        c = _lib.ManoConsts()
        c.hands_mean = dev(a["hands_mean"])
        c.comps = dev(a["hands_components"][:self.ncomps])
        c.blend_T = dev(blend)
        c.v_template = dev(a["v_template"].reshape(-1))
        c.J_template = dev(J_t.reshape(-1))
        c.J_shape = dev(J_s)
        c.weights = dev(a["weights"])
        for i, v in enumerate(synth.MANO_TIPS[self.side]):
            c.tips[i] = v
        c.ncomps = self.ncomps
        self._keep = keep
        self._consts = c
        self._packed_version = ver
        return c

    # -- reference adapter interface -------------------------------------------------------------
    def __call__(self, global_orient, hand_pose, betas, transl):
        prm = torch.cat([global_orient, hand_pose, betas, transl], 1).to(self.device, torch.float32).contiguous()
        B = prm.shape[0]
        verts = torch.empty(B, 778, 3, device=self.device, dtype=torch.float32)
        joints = torch.empty(B, 21, 3, device=self.device, dtype=torch.float32)
        c = self.consts()
        L = _lib.lib()
        _lib.check(L.ev2h_mano(C.byref(c), prm.data_ptr(), prm.shape[1], B, verts.data_ptr(), 0, joints.data_ptr(), 0,
                               _lib.stream_handle()), "ev2h_mano")
        return ManoOutput(verts, joints)


# ------------------------------------------------------------------------------------------- assets
class _Stub:
    """Placeholder for unpicklable chumpy/scipy objects inside MANO_*.pkl: keeps constructor args
    and state so plain arrays can be dug out without importing chumpy."""

    def __init__(self, *a, **k):
        self._args = a

    def __setstate__(self, state):
        self._state = state

    def __reduce_ex__(self, protocol):  # pragma: no cover
        raise TypeError("stub objects are read-only")


class _ManoUnpickler(pickle.Unpickler):
    """MANO_*.pkl are Python-2 protocol-2 pickles that reference `chumpy.ch.Ch` (and other chumpy classes) and
    `scipy.sparse.csc.csc_matrix` as scipy laid it out in 2017.  Neither package is needed to get the arrays out: both families
    are replaced by stubs that keep the pickled state (so the reader does not depend on chumpy being installed or on how the
    installed scipy restores a matrix pickled by an old one)."""

    def find_class(self, module, name):
        if module.startswith("chumpy") or module.startswith("scipy.sparse"):
            return type(name, (_Stub,), {"_pickled_module": module})
        return super().find_class(module, name)


def _as_array(x) -> np.ndarray:
    """ndarray out of a plain array, a scipy sparse matrix (live, or a stub holding its pickled data / indices / indptr / shape)
    or a chumpy stub (its state holds 'x')."""
    if isinstance(x, np.ndarray):
        return x
    if hasattr(x, "toarray"):
        return np.asarray(x.toarray())
    st = getattr(x, "_state", None)
    if isinstance(st, dict):
        if "indptr" in st and "indices" in st and "data" in st:                  # compressed sparse column / row matrix
            shape = st.get("_shape", st.get("shape"))
            if shape is None:
                raise TypeError("sparse matrix state without a shape")
            rows, cols = int(shape[0]), int(shape[1])
            data, indices, indptr = (np.asarray(st[k]) for k in ("data", "indices", "indptr"))
            by_column = type(x).__name__.startswith("csc")
            if not by_column and not type(x).__name__.startswith("csr"):
                raise TypeError(f"unsupported sparse format {type(x).__name__}")
            if indptr.shape[0] != (cols if by_column else rows) + 1:
                raise TypeError("sparse matrix state is inconsistent (indptr length)")
            out = np.zeros((rows, cols), dtype=np.float64)
            major = np.repeat(np.arange(indptr.shape[0] - 1), np.diff(indptr))
            if by_column:
                np.add.at(out, (indices, major), data)                           # duplicates sum, as scipy's toarray() does
            else:
                np.add.at(out, (major, indices), data)
            return out
        for key in ("x", "a"):
            if key in st:
                return _as_array(st[key])
    raise TypeError(f"cannot extract an array from {type(x).__name__}")


def load_mano_pkl(path: str, side: str) -> dict:
    """Chumpy-free reader for the licensed MANO_{LEFT,RIGHT}.pkl (the files manopth's
    `ready_arguments` loads; reference call site model/utils.py:21).  Untested against the real
    files in this repo's CI because they cannot be redistributed."""
    with open(path, "rb") as f:
        d = _ManoUnpickler(f, encoding="latin1").load()
    kin = np.asarray(d["kintree_table"])
    parents = [-1] + [int(v) for v in kin[0, 1:]]
    return {
        "side": side,
        "v_template": _as_array(d["v_template"]).astype(np.float64),
        "shapedirs": _as_array(d["shapedirs"]).astype(np.float64),
        "posedirs": _as_array(d["posedirs"]).astype(np.float64),
        "J_regressor": _as_array(d["J_regressor"]).astype(np.float64),
        "weights": _as_array(d["weights"]).astype(np.float64),
        "hands_components": _as_array(d["hands_components"]).astype(np.float64),
        "hands_mean": _as_array(d["hands_mean"]).astype(np.float64),
        "faces": _as_array(d["f"]).astype(np.int64),
        "parents": parents,
    }


def create_mano_layers(mano_path, device, n_cmps: int = synth.MANO_CMPS, assets: dict | None = None) -> dict:
    """model/utils.py:13-42.  `assets` = {'left': dict, 'right': dict} overrides the pkl files
    under f'{mano_path}/mano/MANO_{LEFT,RIGHT}.pkl'."""
    if assets is None:
        assets = {}
        for side in ("left", "right"):
            p = os.path.join(str(mano_path), "mano", f"MANO_{side.upper()}.pkl")
            if not os.path.exists(p):
                raise FileNotFoundError(f"{p} not found: MANO assets are licensed and not shipped; pass `assets=` "
                                        f"(e.g. ev2hands_amd.synth.synth_mano_assets) or place the files")
            assets[side] = load_mano_pkl(p, side)
    layers = {side: ManoHand(assets[side], device, n_cmps) for side in ("left", "right")}
    if torch.sum(torch.abs(layers["left"].m.th_shapedirs[:, 0, :] - layers["right"].m.th_shapedirs[:, 0, :])) < 1:
        print("Fix th_shapedirs bug of MANO")
        layers["left"].m.th_shapedirs[:, 0, :] *= -1
    return layers
